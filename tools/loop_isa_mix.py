#!/usr/bin/env python
"""Instruction mix of the loop blocks of every kernel of a .hip file (gfx950 ISA from `hipcc -S`): MFMA / VALU / v_mov / LDS /
vector-memory counts per basic block that sits inside a loop and holds matrix instructions (or, with --all, any loop block).

Why: VALU instructions next to an MFMA stream cost ~11 matrix-pipe cycles each (tools/ubench/mfma_side_cost), and the compiler
adds them silently -- round 6 found 64 `v_mov_b32` per K step in `conv_wave_kernel<true, 4>` (a runtime flag that defined the
sampling registers on two paths; the join shuttled them through a second register set).  A block whose v_mov count is a large
share of its VALU count, next to MFMAs, is the signature.

    python tools/loop_isa_mix.py m3dssd_amd/csrc/dcn_wave.hip [--all] [-DFLAG ...]
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "m3dssd_amd", "csrc")


def device_asm(src, defs=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "dev.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
               "--cuda-device-only", "-S", src, "-o", out] + list(defs)
        res = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stderr)
            raise SystemExit(1)
        return open(out).read()


def mix(block):
    c = Counter()
    for ln in block.split("\n"):
        ln = ln.strip()
        if not ln or ln[0] in ";." or ln.endswith(":"):
            continue
        op = ln.split()[0]
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            c["mfma"] += 1
        elif op.startswith("v_mov") or op.startswith("v_accvgpr"):
            c["v_mov"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("buffer_", "global_", "scratch_", "flat_")):
            c["vmem"] += 1
            if op.startswith("scratch_"):
                c["scratch"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
    return c


def kernels(asm):
    for m in re.finditer(r"^(_Z\w+|\w+):\s*; @\1\s*$", asm, re.M):
        start = m.end()
        end = asm.find(".Lfunc_end", start)
        yield m.group(1), asm[start:end]


def main(argv):
    show_all = "--all" in argv
    defs = [a for a in argv if a.startswith("-D")]
    files = [a for a in argv if not a.startswith("-")]
    for f in files:
        asm = device_asm(f, defs)
        for name, body in kernels(asm):
            try:
                dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
            except OSError:
                dem = name
            parts = re.split(r"\n(\.LBB\d+_\d+):", body)
            rows = []
            for k in range(1, len(parts), 2):
                b = parts[k + 1]
                if "in Loop" not in b[:400] and "Loop Header" not in b[:400]:
                    continue
                c = mix(b)
                if c["mfma"] or show_all:
                    rows.append((parts[k], c))
            if rows:
                print(dem)
                for lbl, c in rows:
                    print("   %-10s mfma %3d  valu %3d  v_mov %3d  lds %3d  vmem %3d%s" % (lbl, c["mfma"], c["valu"], c["v_mov"], c["lds"], c["vmem"],
                                                                                          "  SCRATCH %d" % c["scratch"] if c["scratch"] else ""))


if __name__ == "__main__":
    main(sys.argv[1:])
