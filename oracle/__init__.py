"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path (mumianyuxin/M3DSSD: DLA-34 + DCNv2
alignment + ANAB + decode + NMS).  It exists to CHECK the HIP product path and to
serve as bench.py's ``cpu_baseline`` leg.  Nothing under ``m3dssd_amd/``, ``model/``
or ``lib/`` may import it; only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg do, and only as the checker.

Pinning (what makes this oracle trustworthy):
  * tests/golden/*.npz were produced by tools/gen_golden.py, which imports the
    reference's own Python modules from /root/reference in the build container
    (model/M3d_inference_align.py, pose_dla_dcn.py, module/*.py, lib/rpn_util.py
    helpers, lib/nms/py_cpu_nms.py) and dumps their outputs; tests/test_oracle_golden.py
    checks this package against them.
  * The DCNv2 op itself has NO runnable reference implementation here (CUDA-only,
    TH/THC ffi build): it is pinned by the reference's one known-answer test
    (model/DCNv2/test.py:32-65, zero-offset identity) and by closed-form properties
    (zero offset + unit mask == F.conv2d; integer offsets == shifted conv; mask
    linearity).  See DESIGN.md "Oracle".
"""
