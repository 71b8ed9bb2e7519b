#!/usr/bin/env python
"""Static check of the gfx950 device code of every product kernel (run by __graft_entry__.build() and `make -C m3dssd_amd/csrc check`):

  1. no kernel that counts its outstanding vector-memory instructions by hand (HAND_COUNTED below) uses scratch memory: a register
     spill is a vector-memory instruction; the launchers refuse such a build at run time -- this makes the BUILD fail.  Spills of
     the other kernels (fp64 evaluator / refinement natives, ...) are listed as notes;
  2. the ">64-bit store data" write-after-read hazard: a VALU instruction that writes a data register of a 12/16-byte
     buffer / global / flat / scratch store issued less than 1 (buffer store with an SGPR soffset) or 2 (all other forms) wait
     states earlier.  gfx950 HAS this hazard (tools/ubench/vmem_war_hazards.hip: 5 % wrong dwords, element 1 of lanes 12-15 /
     28-31 / 44-47 / 60-63 -- the store reads its data after it has issued) and hipcc (ROCm 7.2) does not insert the wait states:
     round 4 found `buffer_store_dwordx4 v[4:7]` directly followed by `v_pk_fma_f32 v[4:5]` in the K-pair F(4x4) epilogue.

usage: python tools/check_isa_hazards.py [file.hip ...]      (default: every .hip of m3dssd_amd/csrc); exit code 1 on a finding.
The device assembly comes from `hipcc --cuda-device-only -S` with the flags of the Makefile.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "m3dssd_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]

# kernels whose main loops carry hand-counted `s_waitcnt vmcnt(N)` (asm loads): a spill inside them corrupts results
HAND_COUNTED = ("wino44_kernel", "bf16_conv3x3_wide_kernel", "head_mlp_kernel", "wino_wave_kernelILb0E", "bf16_frontend2_kernel")

STORE = re.compile(r"^\s*(buffer|global|flat|scratch)_store_(dwordx3|dwordx4|b96|b128)\s+(.*)$")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(tok):
    """'v[4:7]' / 'v5' / 'a[0:3]' -> set of (file, index)"""
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def store_info(line):
    m = STORE.match(line)
    if not m:
        return None
    kind, ops = m.group(1), [o.strip() for o in m.group(3).split(",")]
    if kind == "buffer":              # buffer_store vdata, vaddr, srsrc, soffset ...
        data = regs_of(ops[0])
        soff = ops[3].split()[0] if len(ops) > 3 else "0"
        need = 1 if re.match(r"^(s\d+|m0|ttmp\d+)$", soff) else 2
    elif kind == "scratch":           # scratch_store vaddr/off, vdata, ...
        data, need = regs_of(ops[1]), 2
    else:                             # global / flat: addr, vdata, ...
        data, need = regs_of(ops[1]), 2
    return data, need


DUAL_DST = ("v_swap_b32", "v_swap_b16", "v_permlane16_swap", "v_permlane32_swap")     # write BOTH of their first two operands


def valu_dst(line):
    """registers written by a VALU instruction (first operand; both operands of the swap forms), else empty."""
    s = line.strip()
    if not s.startswith("v_") or s.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        return set()
    ops = s.split(None, 1)
    if len(ops) < 2:
        return set()
    parts = ops[1].split(",")
    out = regs_of(parts[0])
    if ops[0].startswith(DUAL_DST) and len(parts) > 1:
        out |= regs_of(parts[1])
    return out


BRANCH = re.compile(r"^(s_branch|s_cbranch_\w+)\s+(\S+)")


def scan(asm_text):
    findings, kernel = [], None
    lines = asm_text.splitlines()
    labels = {}
    for i, raw in enumerate(lines):
        m = re.match(r"^([A-Za-z_.$][\w$.]*):", raw.split(";")[0].rstrip())
        if m:
            labels[m.group(1)] = i

    def walk(start, waited, need, data, seen):
        """First VALU write of `data` reachable from line `start` within the hazard window, following BOTH arms of a branch (a
        16-byte store at the end of a loop body followed by the back edge to a write of its data registers is the case the
        fall-through-only scan of round 4 missed) -> (instruction text, wait states seen) or None."""
        j = start
        while waited < need and j < len(lines):
            if (j, waited) in seen:
                return None
            seen.add((j, waited))
            nxt = lines[j].split(";")[0].strip()
            j += 1
            if not nxt or nxt.startswith((".", ";")) or nxt.endswith(":"):
                continue
            if valu_dst(nxt) & data:
                return nxt, waited
            if nxt.startswith("s_endpgm"):
                return None
            b = BRANCH.match(nxt)
            if b:
                waited += 1                                   # the branch itself is one wait state
                tgt = labels.get(b.group(2))
                if tgt is not None and waited < need:
                    hit = walk(tgt, waited, need, data, seen)
                    if hit:
                        return hit
                if b.group(1) == "s_branch":
                    return None                               # unconditional: no fall-through
                continue
            m = re.match(r"^s_nop\s+(\d+)", nxt)
            waited += int(m.group(1)) + 1 if m else 1
        return None

    for i, raw in enumerate(lines):
        line = raw.split(";")[0].rstrip()
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L"):
            kernel = m.group(1)
        si = store_info(line)
        if not si:
            continue
        data, need = si
        hit = walk(i + 1, 0, need, data, set())
        if hit:
            findings.append((kernel, i + 1, line.strip(), hit[0], hit[1], need))
    # scratch per kernel from the metadata block (keys of an entry come in alphabetical order: .name ... .symbol)
    spills = []
    cur = {}
    for raw in lines:
        m = re.match(r"\s*-?\s*\.(name|private_segment_fixed_size|symbol):\s*(\S+)", raw)
        if m:
            cur[m.group(1)] = m.group(2)
            if m.group(1) == "symbol":                   # last key of a kernel's metadata entry (alphabetical order)
                if int(cur.get("private_segment_fixed_size", "0")):
                    spills.append((cur.get("name", "?"), int(cur["private_segment_fixed_size"])))
                cur = {}
    return findings, spills


def device_asm(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "dev.s")
        res = subprocess.run([HIPCC] + FLAGS + ["--cuda-device-only", "-S", src, "-o", out], stdout=subprocess.DEVNULL,
                             stderr=subprocess.PIPE, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stderr)
            raise RuntimeError("check_isa_hazards: hipcc failed on %s (exit code %d)" % (src, res.returncode))
        return open(out).read()


def main(files):
    if not files:
        files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    for f in files:
        text = open(f).read() if f.endswith(".s") else device_asm(f)
        findings, spills = scan(text)
        for kernel, ln, st, wr, waited, need in findings:
            bad += 1
            print("%s: %s: store-data hazard (asm line %d): `%s` then `%s` after %d wait state(s), needs %d"
                  % (os.path.basename(f), kernel, ln, st, wr, waited, need))
        for name, nbytes in spills:
            fatal = any(h in name for h in HAND_COUNTED)
            bad += 1 if fatal else 0
            print("%s: %s: %s uses %d bytes of scratch per lane (register spills)"
                  % (os.path.basename(f), "ERROR" if fatal else "note", name, nbytes))
    print("check_isa_hazards: %d file(s), %d finding(s)" % (len(files), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
