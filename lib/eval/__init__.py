"""Import-path shim: ``lib.eval`` resolves to the MI355X-path KITTI evaluator (same names as the reference's lib/eval)."""
