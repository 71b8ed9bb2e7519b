#!/usr/bin/env python
"""bench.py -- images/sec of the M3DSSD hot path (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic frames already resident in HBM:
RPN.forward (DLA-34 + DCNv2 alignment + ANAB + heads, fp32) -> output bundling -> decode -> top-3000 ->
NMS -> fixed-size detection blocks, and for N > 1 the single RCCL all-gather of those blocks.
Workload = BASELINE.json configs[1]: full M3d_inference_align, bs = 8 per GPU, 1280x384, fp32; weak scaling
(per-GPU batch fixed, images shard naturally, weights replicated).

Reported alongside `value`:
  roofline     -- dominant kernel (the igemm instantiation with the largest share of GPU time): algorithmic
                  FLOPs of its launches / their HIP-event duration measured on the launch stream inside the
                  timed region, vs the 157.3 TFLOP/s dense fp32 MFMA peak of MI355X (MI355X_MICROARCH.md).
  cpu_baseline -- the CPU restatement of the reference (oracle/, kind "port": the reference has no CPU DCNv2)
                  timed on this host on bounded samples (~10 s each at bs = 8 and bs = 1), rank 0 at N = 1 only; the
                  CPU model string, the host's core count and the threads used are reported.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD, 256 CUs, 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), measured 2495
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured streaming copy)
CROP = (384, 1280)
PER_GPU_BATCH = 8
MFMA_FAMILIES = ("igemm", "wino", "head_mlp", "conv_wave", "anab_attend", "bf16_conv", "bf16_halo", "bf16_wide", "bf16_anab", "bf16_dcn_patch", "bf16_dcn1x1", "bf16_c64",
                 "bf16_head_mlp", "bf16_head2", "bf16_tail2", "bf16_qkvs", "bf16_tree_entry", "bf16_frontend")
# SURVEY 8d, per image: 105.8 GFLOP; activations 1003.6 MB (fp32) + outputs 21 MB + input 5.9 MB; weights 82.6 MB (fp32) per batch
ALG_GFLOP_PER_IMAGE = 105.8
ALG_MB_PER_IMAGE_F32 = 1003.6 + 21.0 + 5.9
ALG_MB_WEIGHTS_F32 = 82.6


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_worker(spec):
    """`python bench.py --cpu-worker i,K,threads,budget_s`: one of K oracle processes of the whole-host CPU baseline, pinned to its
    own core set; prints {"frames": n, "seconds": dt} (frames at bs = 1: K frames are in flight on the host at any time)."""
    i, K, threads, budget = spec.split(",")
    i, K, threads, budget = int(i), int(K), int(threads), float(budget)
    cpus = sorted(os.sched_getaffinity(0))
    per = max(1, len(cpus) // K)
    mine = cpus[i * per:(i + 1) * per] or cpus
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        pass
    torch.set_num_threads(min(threads, len(mine)))
    from m3dssd_amd import synth
    from oracle import detect as odet
    from oracle import model_cpu
    sd = synth.synth_state_dict(0)
    conf = synth.synth_conf(CROP, 0, batch_size=1, device="cpu")
    x = synth.synth_frames(1, CROP, 99 + i)

    def one():
        with torch.no_grad():
            cls, prob, b2, b3, fs, rois = model_cpu.rpn_forward(sd, conf, x)
            odet.detect_image(prob[0], b2[0], b3[0], rois, conf)
    one()
    print(json.dumps({"ready": i}), flush=True)
    sys.stdin.readline()                                     # all K workers start their timed loop together
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget or n >= 400:
            break
    print(json.dumps({"frames": n, "seconds": dt, "cpus": len(mine)}), flush=True)


def cpu_baseline(sd, budget_s=12.0):
    """Oracle forward + decode + NMS on ALL host cores: K = host_threads / 32 oracle processes pinned to disjoint
    core sets, each with 32 torch threads at bs = 1 (the torch-CPU convolutions on these maps stop scaling past ~32 threads, and
    one process at bs = 8 is slower per image than at bs = 1: the port's DCN im2col loops per image), all timed over the same
    window -> `value` = aggregate images/s with K frames in flight; `value_bs1` = one such process alone on the host."""
    import subprocess
    host = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # the container's CPU-time quota (cgroup v2 cpu.max / v1 cfs_quota): the GPU boxes show 256 logical CPUs and grant 16 of them
    # ("1600000 100000") -- more runnable threads than the quota only adds throttling (8 x 32 threads: 0.26 images/s against
    # 2.9 for one process, round 4), so the baseline uses what the container may actually burn and says so
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) if q > 0 else None
        except (OSError, ValueError):
            pass
    usable = max(1, min(host, int(quota))) if quota else host
    threads = min(32, usable)
    K = max(1, usable // threads)

    def run(k_total, budget):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "%d,%d,%d,%g" % (i, k_total, threads, budget)],
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env,
                                  cwd=ROOT) for i in range(k_total)]
        try:
            for p in procs:                                  # wait until every worker has imported, built and warmed up
                line = p.stdout.readline()
                if "ready" not in line:
                    raise RuntimeError("cpu_baseline worker failed to start")
            for p in procs:
                p.stdin.write("go\n")
                p.stdin.flush()
            res = [json.loads(p.stdout.readline()) for p in procs]
        finally:
            for p in procs:
                try:
                    p.wait(timeout=60)
                except Exception:
                    p.kill()
        return res
    alone = run(1, min(budget_s, 6.0))[0] if K > 1 else None      # (K == 1: `value` IS the one-process figure)
    res = run(K, budget_s)
    frames, secs = sum(r["frames"] for r in res), max(r["seconds"] for r in res)
    out = {"value": round(frames / secs, 3), "unit": "images/sec", "cores": K * threads, "host_cores": host,
           "cgroup_cpu_quota": quota,
           "cpu_model": _cpu_model(), "kind": "port", "processes": K, "threads_per_process": threads,
           "sample": "oracle forward+decode+NMS (torch-CPU + oracle/*.c), 1280x384: %d processes x %d threads pinned to disjoint core "
                     "sets, bs=1 each (%d frames in flight), %d frames in %.1f s -> value" % (K, threads, K, frames, secs)}
    if alone is not None:
        out["value_bs1"] = round(alone["frames"] / alone["seconds"], 3)
        out["sample"] += "; one such process alone: %d frames in %.1f s -> value_bs1" % (alone["frames"], alone["seconds"])
    return out


def kernel_symbol(label):
    """engine label -> demangled kernel name as rocprofv3 prints it."""
    import re
    if label.startswith("bf16_anab"):
        return "void bf16_anab_attend_kernel<true>(AnabArgs)"
    if label.startswith("anab_attend"):
        return "void anab_attend_f32_kernel<168>(AnabF32Args)"
    if label.startswith("bf16_head2"):
        return "bf16_head2_kernel(Head2Args)"
    if label.startswith("bf16_tail2"):
        return "bf16_tail2_kernel(Tail2Args)"
    if label.startswith("bf16_qkvs"):
        return "bf16_qkvs_kernel(QkvsArgs)"
    if label.startswith("bf16_tree_entry"):
        return "void bf16_tree_entry_kernel<4, 2>(TreeEntryArgs)"
    if label.startswith("bf16_head_mlp"):
        return "bf16_head_mlp_kernel(HeadArgs)"
    if label.startswith("bf16_frontend2"):
        return "void bf16_frontend2_kernel<2, %s>(Front2Args)" % ("true" if "u8" in label else "false")
    if label.startswith("bf16_frontend"):
        return "bf16_frontend_kernel(FrontArgs)"
    if label.startswith("bf16_c64"):               # (two instantiations, with / without a residual)
        return "void bf16_conv3x3_c64_kernel<true>(Bf16Args)"
    if label.startswith("bf16_dcn1x1"):
        return "bf16_dcn1x1_kernel(Bf16Args)"
    if label.startswith("bf16_dcn_patch"):
        th = re.findall(r"\d+", label.split("<", 1)[1])[0]
        return "void bf16_dcn_patch_kernel<%s, %s>(Bf16Args, void const*, unsigned int const*)" % (th, "9, 3" if th == "16" else "6, 1")
    if label.startswith("bf16_wide"):              # (two instantiations, with / without a residual: the one without stands for the family)
        return "void bf16_conv3x3_wide_kernel<false>(WideArgs)"
    if label.startswith("bf16_halo"):
        bn, tw = re.findall(r"\d+", label.split("<", 1)[1])[:2]
        return "void bf16_conv3x3_halo_kernel<%s, %s, %d, %d, %d>(Bf16Args)" % (bn, tw, 8 * int(tw), 4 if tw == "16" else 8,
                                                                                32 if (bn == "128" and tw == "16") else 64)
    if label.startswith("bf16_conv"):
        return "void bf16_conv_kernel<%s, %s>(Bf16Args)" % (re.findall(r"\d+", label.split("<", 1)[1])[0],
                                                            "true" if "deform" in label else "false")
    if label.startswith("wino44_c16"):
        return "wino44_c16_kernel(Wino44C16Args)"
    if label.startswith("wino44"):                 # wino44<16,32[,splitkN]>: 16 tiles x 32 (NB = 2) or 16 (NB = 1) channels per wave
        nb = int(re.findall(r"\d+", label)[2]) // 16   # (the 64-channel form runs two workgroups per CU unless M3D_W44_OCC2=0)
        occ = 2 if (nb == 1 and os.environ.get("M3D_W44_OCC2", "1") != "0") else 1
        return "void wino44_kernel<%d, %d, %d>(Wino44Args)" % (nb, occ, 2 if "kpair" in label else 1)
    if label.startswith("wino_wave"):
        return "void wino_wave_kernel<%s>(WinoArgs)" % ("true" if "splitk" in label else "false")
    if label.startswith("wino"):
        return "wino_kernel(WinoArgs)"
    if label.startswith("conv_wave"):
        # (+ split-K reduce; template tail = corners gathered per tap, register bound in waves per SIMD: csrc/dcn_wave.hip)
        return "void conv_wave_kernel<%s, 4, 4, %d>(ConvWaveArgs)" % (("true", 2) if "deform" in label else ("false", 3))
    if label.startswith("head_mlp"):
        layers, n3 = re.findall(r"\d+", label)[:2]
        return "void head_mlp_kernel<%s, %s>(MlpBatch)" % ("true" if layers == "3" else "false", n3)
    nums = re.findall(r"\d+", label)[:3]
    wm, wn = (4, 1) if nums[1] == "32" else (2, 2)
    return "void igemm_kernel<%s, %s, %s, %d, %d, %s, %s>(IgemmArgs)" % (
        nums[0], nums[1], nums[2], wm, wn, "true" if "deform" in label else "false",
        "true" if "planar" in label else "false")


# engine family label prefix -> the kernel sources whose edit invalidates a PMC pass of that family (plus the shared headers)
FAMILY_SOURCES = (("bf16_anab", ("bf16_anab.hip",)), ("anab_attend", ("anab_attend.hip",)), ("bf16_head2", ("bf16_head_mlp2.hip",)), ("bf16_tail2", ("bf16_head_mlp2.hip",)), ("bf16_qkvs", ("bf16_head_mlp2.hip",)),
                  ("bf16_tree_entry", ("bf16_tree_entry.hip",)), ("bf16_head_mlp", ("bf16_head_mlp.hip",)),
                  ("bf16_frontend2", ("bf16_frontend2.hip",)), ("bf16_frontend", ("bf16_frontend.hip",)),
                  ("bf16_dcn1x1", ("bf16_dcn1x1.hip",)), ("bf16_c64", ("bf16_conv_c64.hip",)), ("bf16_dcn_patch", ("bf16_dcn_patch.hip", "bf16_conv.hip")), ("bf16_wide", ("bf16_conv_wide.hip",)),
                  ("bf16_halo", ("bf16_conv.hip",)), ("bf16_conv", ("bf16_conv.hip",)), ("wino44", ("wino44_conv.hip",)),
                  ("wino", ("wino_conv.hip",)), ("conv_wave", ("dcn_wave.hip", "igemm_conv.hip")), ("head_mlp", ("head_mlp.hip",)),
                  ("igemm", ("igemm_conv.hip",)))
SHARED_SOURCES = ("common.h", "bf16_tile.h", "m3dssd_hip.h")
PROFILES_DIR = os.path.join(ROOT, "profiles")


def family_sources(kernel_label):
    for prefix, files in FAMILY_SOURCES:
        if kernel_label.startswith(prefix):
            return files + SHARED_SOURCES
    return SHARED_SOURCES


def traffic_is_stale(doc, kernel_label, lib_hashes):
    """A committed PMC pass is stale for a family when the library that is LOADED NOW was built from other sources of that
    family's kernels than the library the pass ran: `csrc_files` of the pass (name -> sha256[:16], written by
    tools/pmc_traffic.py from m3d_source_hashes() of the profiled library) against the loaded library's own record.  A pass that
    carries no record (rounds 1-4) cannot be checked: stale."""
    rec = doc.get("csrc_files")
    if not rec or not lib_hashes:
        return True
    return any(rec.get(f) != lib_hashes.get(f) for f in family_sources(kernel_label))


def pmc_traffic(kernel_label, lib_hashes=None, profiles_dir=None):
    """(HBM bytes per launch, source file, stale) of a kernel family from the committed PMC passes (profiles/*_hbm_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 wide-read correction applied); (None, None, None) if the
    kernel was not profiled.  PMC counters cannot be collected inside the timed run itself: the figure is a constant from the
    builder's profiling lease, labelled as such (`traffic_source`) and checked against the loaded library's source record
    (`traffic_stale`)."""
    import glob
    if lib_hashes is None:
        from m3dssd_amd import _hip
        lib_hashes = _hip.lib_source_hashes()
    sym = kernel_symbol(kernel_label)
    for f in sorted(glob.glob(os.path.join(profiles_dir or PROFILES_DIR, "*_hbm_traffic.json")), reverse=True):
        try:
            doc = json.load(open(f))
        except Exception:
            continue
        fam = doc.get("families") or {}
        rel = os.path.relpath(f, ROOT)
        if kernel_label in fam:          # per-FAMILY average (tools/pmc_traffic.py aligned with --dump-launches)
            return fam[kernel_label]["hbm_bytes_per_launch"], rel + "#families", traffic_is_stale(doc, kernel_label, lib_hashes)
        if fam:
            continue                     # a file with family rows that lacks this label is from another dtype / plan
        if sym in doc.get("kernels", {}):   # older passes: per-SYMBOL average (one symbol can serve several families)
            return doc["kernels"][sym]["hbm_bytes_per_launch"], rel + "#symbol-average", traffic_is_stale(doc, kernel_label, lib_hashes)
    return None, None, None


def algorithmic_bytes(op):
    """HBM bytes one launch has to move: input + output (+ residual, + offsets/masks) + weights, once each, fp32."""
    d = op[4]
    if d is None:
        return None
    if hasattr(d, "hbm_bytes"):                           # engine.OpCost: a helper launch that states its own byte count
        return d.hbm_bytes
    if hasattr(d, "waf"):                                 # m3d_tail2_bf16_desc: 256-channel input, two weight sets once, planar fp32 output
        return d.M * 256 * 2 + d.M * d.Cout * 4 + (256 * 256 + 256 * d.Cout) * 2
    if hasattr(d, "w1f"):                                 # m3d_head2_bf16_desc: G heads, one 128-channel input, weights once
        return (d.M * 128 * 2 + d.groups * (d.M * d.Cout * 4 + (128 * 256 + 256 * 256 + 256 * 64) * 2))
    if hasattr(d, "out_group_off") and hasattr(d, "groups") and not hasattr(d, "Kpad"):   # m3d_head_bf16_desc: G heads, one input
        return (d.M * d.Cin * 2 + d.groups * (d.M * d.Cout * 4 + (d.Cin * 256 + 256 * 256 + 256 * d.Cout_pad) * 2))
    if hasattr(d, "Kpad"):                                # m3d_conv_bf16_desc: bf16 in / weights, bf16 or fp32 out
        g = max(d.groups, 1)
        o = d.N * d.Ho * d.Wo * d.Cout * (2 if d.out_mode == 0 else 4) * g
        b = d.N * d.H * d.W * d.Cin * 2 * (g if d.in_group_off else 1) + o + d.Cout * d.kh * d.kw * d.Cin * 2 * g
        if d.wgt_img_stride:
            b += (d.N - 1) * d.Cout * d.Cin * 2
        if d.res:
            b += d.N * d.Ho * d.Wo * d.Cout * 2
        if d.dcn_offmask:
            b += d.N * d.Ho * d.Wo * 3 * d.kh * d.kw * 4
        return b
    if hasattr(d, "Ho"):                                  # m3d_conv_desc
        o = d.N * d.Ho * d.Wo * d.Cout * 4
        b = d.N * d.H * d.W * d.Cin * 4 + o + (o if d.res else 0) + d.Cout * d.kh * d.kw * d.Cin * 4
        if d.dcn_offmask:
            b += d.N * d.Ho * d.Wo * 3 * d.kh * d.kw * 4
        return b
    b = 0                                                 # m3d_mlp_desc or an array of them (batched heads)
    seen = set()
    for h in (d if hasattr(d, "__len__") else [d]):
        if h.inp not in seen:                             # heads of one launch that read the same map read it once
            seen.add(h.inp)
            b += h.M * h.Cin * 4
        b += h.M * h.Cout * 4 + ((h.Cin * 256 if h.w1 else 0) + 256 * 256 + 256 * h.Cout) * 4
    return b


def step_counter_bytes(families, helpers):
    """HBM bytes of one step from the committed PMC passes: launches x traffic per launch, per family (None without any pass)."""
    tot, have = 0.0, False
    for v in families.values():
        if v.get("traffic"):
            tot += v["launches_per_step"] * v["traffic"]
            have = True
        elif v.get("algorithmic_bytes_per_launch"):
            tot += v["launches_per_step"] * v["algorithmic_bytes_per_launch"]
    for v in helpers.values():
        if v.get("algorithmic_bytes_per_launch"):
            tot += v["launches_per_step"] * v["algorithmic_bytes_per_launch"]
    return int(tot) if have else None


def rccl_proof(args, mdist, net, conf, B, rank, world, dev, last_gather):
    """N > 1 only: evidence that the collective saw `world` distinct devices, what it costs in isolation, and that the gathered
    block holds every rank's shard where the contract says (rank r's images at rows [r*B, (r+1)*B): rank 0 recomputes each
    rank's batch -- seed 1234 + r -- on its own GPU and compares)."""
    import torch.distributed as dist
    from m3dssd_amd import synth
    from m3dssd_amd.host.detect import detect_device, select_block
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device_index": dev.index, "name": props.name, "uuid": str(getattr(props, "uuid", "")),
          "pci_bus_id": getattr(props, "pci_bus_id", None), "pid": os.getpid()}
    seen = [None] * world
    dist.all_gather_object(seen, me)
    dets, counts = last_gather[0].clone(), last_gather[1].clone()     # (views of the cached receive buffer: the timing loop reuses it)
    blk = torch.zeros(B, dets.shape[1] + 1, dets.shape[2], device=dev, dtype=torch.float32)
    for _ in range(10):
        mdist.gather_block(blk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        mdist.gather_block(blk)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10.0
    ok, checked = True, 0
    if rank == 0:
        for r in range(world):
            xr = synth.synth_frames(B, CROP, 1234 + r).to(dev)
            blk_r, cnt_r = select_block(*detect_device(net, xr, conf), conf)
            same = torch.equal(blk_r[:, :-1], dets[r * B:(r + 1) * B]) and torch.equal(cnt_r, counts[r * B:(r + 1) * B])
            ok, checked = ok and bool(same), checked + 1
    dist.barrier()
    distinct = len({(d["uuid"], d["pci_bus_id"], d["device_index"]) for d in seen})
    return {"world_size": world, "backend": mdist.backend_name(), "ranks_seen": seen, "distinct_devices": distinct,
            "gathered_rows": [int(v) for v in dets.shape], "allgather_us": round(us, 2),
            "allgather_bytes_per_rank": int(blk.numel() * 4),
            "shards_recomputed_on_rank0": checked, "shards_match": ok}


LINE_LIMIT = 4096          # the driver keeps the tail of stdout: the one JSON line must fit (BENCH_r05.json: parsed = null at 24 KB)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _compact_roofline(r):
    """The dominant symbol's roofline object, numbers only (notes / sources / family lists stay in the detail file)."""
    if not isinstance(r, dict):
        return None
    c = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches_timed", "traffic",
                  "algorithmic_bytes_per_launch", "traffic_stale", "share_of_gpu_time"))
    c.setdefault("traffic", None)                        # the contract's key: null when no PMC pass covers the kernel
    if r.get("traffic") and r.get("algorithmic_bytes_per_launch"):
        c["traffic_ratio"] = round(r["traffic"] / r["algorithmic_bytes_per_launch"], 2)
    return c


def _compact_config(o):
    """value / ms_per_step / whole-step fractions / dominant-kernel fraction of one measured configuration."""
    if not isinstance(o, dict):
        return None
    if "error" in o and "value" not in o:
        return {"value": None, "error": str(o["error"])[:160]}
    c = _pick(o, ("value", "ms_per_step", "steps", "dtype", "mfma_time_weighted_frac", "sclk_under_step_ghz"))
    sr = o.get("step_roofline") or {}
    if sr:
        c["step"] = _pick(sr, ("algorithmic_tflops", "mfma_frac", "hbm_frac"))
    if isinstance(o.get("config"), dict) and "per_gpu_batch" in o["config"]:
        c["per_gpu_batch"] = o["config"]["per_gpu_batch"]
    rf = _compact_roofline(o.get("roofline"))
    if rf:
        c["roofline"] = rf
    if "dropin" in o:
        c["dropin"] = o["dropin"]
    if "work_in_timed_region" in o:
        c["work_in_timed_region"] = o["work_in_timed_region"]
    return c


def _clamp_strings(v, n=200):
    if isinstance(v, dict):
        return {k: (x if k == "workload" else _clamp_strings(x, n)) for k, x in v.items()}
    if isinstance(v, list):
        return [_clamp_strings(x, n) for x in v]
    if isinstance(v, str) and len(v) > n:
        return v[:n - 3] + "..."
    return v


def compact_line(out, detail_path=None):
    """The ONE stdout line: headline keys of the bench contract, the dominant kernel's roofline, cpu_baseline and
    compact summaries of the side legs.  Everything else lives in the detail file.  Guaranteed < LINE_LIMIT bytes: optional
    blocks are dropped (least important first) until it fits."""
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    line = {k: out.get(k) for k in head}
    cfg = out.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "per_gpu_batch", "global_batch", "resolution", "parallelism"))
    if len(line["config"].get("workload", "")) > 260:
        line["config"]["workload"] = line["config"]["workload"][:257] + "..."
    line["launch"] = str(out.get("launch", ""))[:120]
    if "work_in_timed_region" in out:
        line["work_in_timed_region"] = out["work_in_timed_region"]
    line["roofline"] = _compact_roofline(out.get("roofline"))
    sr = out.get("step_roofline") or {}
    if sr:
        line["step_roofline"] = _pick(sr, ("algorithmic_tflops", "mfma_frac", "hbm_frac", "mfma_time_weighted_frac"))
    for k in ("mfma_time_weighted_frac", "sclk_under_step_ghz", "roofline_frac_at_held_clock", "dist_backend"):
        if out.get(k) is not None:
            line[k] = out[k]
    if "dropin" in out:
        line["dropin"] = out["dropin"]
    if "configs2_bf16" in out:
        line["configs2_bf16"] = _compact_config(out["configs2_bf16"])
    c3 = out.get("configs3_shard32")
    if isinstance(c3, dict):
        line["configs3_shard32"] = {k: (_pick(v, ("value", "ms_per_step", "steps", "mfma_frac", "hbm_frac", "error")) if isinstance(v, dict) else v)
                                    for k, v in c3.items() if k in ("f32", "bf16", "allgather_us_one_rank_rccl", "allgather_bytes_per_rank")}
    if isinstance(out.get("feed_u8"), dict):
        line["feed_u8"] = _pick(out["feed_u8"], ("value", "ms_per_step", "vs_resident", "h2d_bytes_per_step", "error"))
    if isinstance(out.get("rccl"), dict):
        line["rccl"] = _pick(out["rccl"], ("world_size", "backend", "distinct_devices", "allgather_us", "shards_match", "gathered_rows"))
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "cpu_model", "host_cores", "error"))
        c.setdefault("value", None)
        if cb.get("sample"):
            c["sample"] = str(cb["sample"])[:200]
        line["cpu_baseline"] = c
    if detail_path:
        line["detail"] = detail_path
    line = _clamp_strings(line)
    # never exceed the limit: drop optional blocks, least important first
    for drop in (None, "feed_u8", "rccl", "configs3_shard32", "step_roofline", "launch", "dropin", "configs2_bf16"):
        if drop is not None:
            line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
        if len(text) < LINE_LIMIT:
            return text
    raise RuntimeError("bench.py: the headline line alone exceeds %d bytes" % LINE_LIMIT)


def emit(out, detail_path, fd):
    """Full record -> the detail file (and stderr); ONE compact line -> the real stdout."""
    if detail_path is None:
        detail_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_detail.json")
    shown = None
    try:
        with open(detail_path, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
        shown = os.path.relpath(detail_path, os.path.dirname(os.path.abspath(__file__)))
        if shown.startswith(".."):
            shown = detail_path
    except OSError as e:                               # read-only checkout: the line still goes out
        sys.stderr.write("bench.py: cannot write %s: %s\n" % (detail_path, e))
    sys.stderr.write("bench.py detail record:\n" + json.dumps(out) + "\n")
    sys.stderr.flush()
    sys.stdout.flush()
    os.write(fd, (compact_line(out, shown) + "\n").encode())


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 200 for f32 bs=8, 60 for bf16 bs=64: ~1.2 s timed)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 8 for f32, 64 for bf16)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = BASELINE.json configs[1] (the headline metric); bf16 = configs[2] (bs=64, bf16 storage / MFMA)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs2", action="store_true",
                    help="skip the extra configs[2] (bs=64 bf16) measurement that the default N=1 f32 run appends as 'configs2_bf16'")
    ap.add_argument("--configs2-steps", type=int, default=60)
    ap.add_argument("--no-configs3", action="store_true",
                    help="skip the configs[3]-shaped leg (32 images per GPU, f32 and bf16) that the default N=1 f32 run appends as 'configs3_shard32'")
    ap.add_argument("--configs3-steps", type=int, default=30)
    ap.add_argument("--no-feed", action="store_true",
                    help="skip the fed-input leg (uint8 frames from pinned host memory every step) that the default N=1 f32 run appends as 'feed_u8'")
    ap.add_argument("--dump-layers", default=None, help="write the per-launch table of one instrumented step here")
    ap.add_argument("--dump-launches", default=None,
                    help="write the ordered [family label, kernel base name] list of one step's MFMA launches (for tools/pmc_traffic.py)")
    ap.add_argument("--detail", default=None,
                    help="where the full record (kernel family tables, helper tables, per-kernel breakdowns, notes) is written; "
                         "default: bench_detail.json next to this script.  stdout carries ONE compact line (< 4 KB)")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the drop-in leg (the reference's own call sequence net(x) -> six tensors -> detect, eager and as one graph)")
    ap.add_argument("--dropin-steps", type=int, default=None)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not overlap decode/top-k/NMS of batch k with the forward of batch k+1")
    return ap.parse_args()


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(sys.argv[2])
    args = parse_args()
    # ONE JSON line on stdout, whatever the libraries print: RCCL writes its version banner to the C stdout when a communicator is
    # created (seen behind the JSON line of a run with the one-rank collective leg).  File descriptor 1 points at stderr for the
    # whole run (and is NOT restored: RCCL's banner sits in the C library's stdout buffer until the process exits); the line goes
    # to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    return _main(args, real_stdout)


def _main(args, real_stdout):
    from m3dssd_amd import dist as mdist

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one process per GPU under torch.distributed.run
        # (RCCL rendezvous on 127.0.0.1; rank 0 prints the JSON line on the inherited stdout)
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("OMP_NUM_THREADS", "8")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.dup2(real_stdout, 1)                    # the launcher and its ranks inherit the real stdout (rank 0 prints the line)
        os.execv(sys.executable, cmd)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    backend = os.environ.get("M3D_DIST_BACKEND")          # None: init_from_env's own default (nccl = RCCL when a GPU is visible)
    if args.gpus > 1 and backend in (None, "nccl") and torch.cuda.device_count() < args.gpus:
        raise SystemExit("--gpus %d with the nccl (RCCL) backend needs %d visible devices, found %d "
                         "(M3D_DIST_BACKEND=gloo lets test ranks share a device)" % (args.gpus, args.gpus, torch.cuda.device_count()))
    rank, world, local = mdist.init_from_env(backend)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    out, sd = run_config(args, args.dtype, args.steps, args.batch, rank, world, dev, mdist, dump_layers=args.dump_layers,
                         feed=(world == 1 and args.dtype == "f32" and args.batch is None and not args.no_feed))
    if rank == 0:
        if world > 1:
            out["dist_backend"] = mdist.backend_name()
        if world == 1 and args.dtype == "f32" and args.batch is None and not args.no_configs2:
            # BASELINE.json configs[2] (bs = 64, bf16 storage + MFMA) on the same clock as the headline line: same step
            # definition, same timed-region bracket, its own roofline objects.
            torch.cuda.empty_cache()
            try:
                c2, _ = run_config(args, "bf16", args.configs2_steps, None, rank, world, dev, mdist)
            except Exception as e:                 # noqa: BLE001  (side leg: the headline line survives)
                c2 = {"error": "%s: %s" % (type(e).__name__, e)}
            out["configs2_bf16"] = {k: c2[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config",
                                                        "launch", "roofline", "step_roofline", "gpu_ms_by_kernel_one_step", "dropin", "work_in_timed_region",
                                                        "mfma_kernel_families", "helper_kernels", "mfma_time_weighted_frac",
                                                        "sclk_under_step_ghz", "roofline_frac_at_held_clock", "error") if k in c2}
        if world == 1 and args.dtype == "f32" and args.batch is None and not args.no_configs3:
            # BASELINE.json configs[3]: bs = 256 sharded 32 per GPU over 8 MI355X.  One GPU measures its shard (32 images per
            # step, both precisions, same bracket) and the one-rank RCCL cost of the [32, 41, 14] block every rank contributes;
            # `python bench.py --gpus 8 --batch 32` is the full shape when a node is available.
            c3 = {"workload": "BASELINE.json configs[3] per-GPU shard: 32 images/GPU per step, 1280x384 (bs 256 = 8 x 32)"}
            for dt_ in ("f32", "bf16"):
                torch.cuda.empty_cache()
                try:
                    c3[dt_] = shard_leg(dt_, 32, args.configs3_steps, dev)
                except Exception as e:             # noqa: BLE001  (side leg)
                    c3[dt_] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
            c3["allgather_us_one_rank_rccl"] = rccl_one_rank_allgather_us(32, 40, dev)
            c3["allgather_bytes_per_rank"] = 32 * 41 * 14 * 4
            out["configs3_shard32"] = c3
        if world == 1 and not args.no_cpu_baseline:
            try:                                   # a reported side figure: its failure must not take the headline line with it
                out["cpu_baseline"] = cpu_baseline(sd)
            except Exception as e:                 # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "images/sec", "kind": "port", "error": "%s: %s" % (type(e).__name__, e)}
        emit(out, args.detail, real_stdout)
    if world > 1:
        torch.distributed.destroy_process_group()


def dropin_leg(net, x, conf, steps):
    """The reference's OWN call sequence on the same frames: ``net(x)`` returning the six tensors of
    M3d_inference_align.py:303-313 -- bundle_outputs runs, the four [B, N, C] outputs are fresh tensors -- then decode + top-3000 + NMS
    + row selection on those tensors (lib/rpn_util.py:1439-1553, batched), launched eagerly (what a script calling the module gets)
    and as ONE captured hipGraph.  Same bracket as the headline: synchronize, K steps, synchronize.  No overlap between batches."""
    from m3dssd_amd.host.detect import detect_from_outputs, select_block
    dev = x.device
    B = x.shape[0]
    eng = net.engine()
    plan = eng.plan_for(B, x.shape[2], x.shape[3])

    def seq():
        with torch.no_grad():
            cls, prob, b2, b3, feat_size, rois = net(x)
            return select_block(*detect_from_outputs(eng, plan, prob, b2, b3, rois, conf), conf)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"value": round(B * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3)}

    out = {"steps": steps, "eager": timed(seq)}
    graph = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(dev)
    cap.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(cap):
        seq()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=cap):
            g_block, g_counts = seq()
    torch.cuda.current_stream(dev).wait_stream(cap)
    out["graph"] = timed(graph.replay)
    del graph
    return out


def feed_u8_leg(net, conf, B, dev, steps, resident_ms):
    """The fed-input form of the same step: raw uint8 BGR frames [B, 375, 1242, 3] -- a DIFFERENT set every step --
    go from pinned host memory to the device on a copy stream, double buffered against the graph of the previous batch, and the
    stem reads them directly (Preprocess fused into its loads).  Replaces the reference's per-frame host preprocessing + im.cuda()
    (lib/rpn_util.py:1427-1429, lib/dataloader.py:934-950, lib/augmentations.py:472-501).  Never part of `value`."""
    import numpy as np
    from m3dssd_amd.pipeline import PipelinedDetector
    fh, fw = 375, 1242                                     # KITTI frame size, padded to the 384 x 1280 crop inside the stem
    rng = np.random.RandomState(7)
    pool = [torch.from_numpy(rng.randint(0, 256, size=(B, fh, fw, 3)).astype(np.uint8)).pin_memory() for _ in range(4)]
    pipe = PipelinedDetector(net, conf, B, CROP[0], CROP[1], u8_frame=(fh, fw))
    nbytes = pool[0].numel()
    # raw H2D rate of one frame set, alone on the copy stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst = torch.empty_like(pool[0], device=dev)
    dst.copy_(pool[0], non_blocking=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(8):
        dst.copy_(pool[i % 4], non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    h2d_alone_ms = e0.elapsed_time(e1) / 8
    # warm-up, then the timed loop: upload of batch k + 1 is issued before the graph of batch k
    pipe.feed(pool[0])
    for k in range(3):
        pipe.feed(pool[(k + 1) % 4])
        pipe.step_fed(as_block=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        pipe.feed(pool[k % 4])                             # queued: graph k uploads it (a kernel on its side branch) for step k + 1
        pipe.step_fed(as_block=True)
    pipe.step_fed(as_block=True)                           # the batch fed during warm-up keeps the count at `steps` + 1 submitted;
    pipe.flush(as_block=True)                              # flush drains the last one
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = 1e3 * dt / (steps + 1)
    return {"value": round(B * (steps + 1) / dt, 2), "unit": "images/sec", "steps": steps + 1, "ms_per_step": round(ms, 3),
            "vs_resident": round(resident_ms / ms, 4), "frame": [fw, fh], "input": "uint8 BGR [B, 375, 1242, 3] from pinned host "
            "memory, 4 distinct frame sets cycled, uploaded by a kernel on the side branch of the previous batch's graph (m3d_upload_indirect), Preprocess fused into the stem (m3d_stem_conv7x7_u8)",
            "h2d_bytes_per_step": int(nbytes), "h2d_gbs_effective": round(nbytes / (ms * 1e-3) / 1e9, 2),
            "h2d_alone_ms": round(h2d_alone_ms, 4), "h2d_alone_gbs": round(nbytes / (h2d_alone_ms * 1e-3) / 1e9, 2)}


def rccl_one_rank_allgather_us(B, post, dev):
    """The collective of SURVEY 8e on its real block size ([B, post + 1, 14] fp32) through the nccl (= RCCL) backend with ONE rank --
    what a single-GPU lease can measure of it: launch + protocol latency without a peer.  None when no process group can be made."""
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        return None
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    try:
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
        blk = torch.zeros(B, post + 1, 14, device=dev, dtype=torch.float32)
        out = torch.empty_like(blk)
        for _ in range(10):
            dist.all_gather_into_tensor(out, blk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            dist.all_gather_into_tensor(out, blk)
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) * 10.0, 2)
    except Exception:                  # noqa: BLE001
        return None
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def shard_leg(dtype, B, steps, dev):
    """One (dtype, per-GPU batch) point with the step definition and timed bracket of the headline line (pipelined hipGraph replays,
    `steps` batches forwarded AND detected inside the bracket), without the instrumentation passes: BASELINE.json configs[3] is
    bs = 256 sharded 32 per GPU -- this is its per-GPU shard on one MI355X."""
    from m3dssd_amd import synth
    from m3dssd_amd.pipeline import PipelinedDetector
    from model.M3d_inference_align import build
    conf = synth.synth_conf(CROP, 0, batch_size=B, device=str(dev))
    net = build(conf, "test")
    net.load_state_dict(synth.synth_state_dict(0), strict=True)
    net = net.to(dev).set_compute_dtype(dtype)
    x = synth.synth_frames(B, CROP, 1234).to(dev)
    pipe = PipelinedDetector(net, conf, B, CROP[0], CROP[1])
    pipe.input.copy_(x)
    for _ in range(3):
        pipe.step(as_block=True)
    pipe.flush(as_block=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.step(as_block=True)
    r = pipe.flush(as_block=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    counts = r[1]
    out = {"value": round(B * steps / dt, 2), "unit": "images/sec", "per_gpu_batch": B, "steps": steps,
           "ms_per_step": round(1e3 * dt / steps, 3), "dtype": dtype, "detections_last_batch": int(counts.sum().item()),
           "algorithmic_tflops": round(B * ALG_GFLOP_PER_IMAGE * 1e9 * steps / dt / 1e12, 1)}
    del pipe, net
    torch.cuda.empty_cache()
    return out


def run_config(args, dtype, steps, batch, rank, world, dev, mdist, dump_layers=None, feed=False):
    """Warm up, instrument, time `steps` steps of one (dtype, batch) configuration; returns (JSON dict or None off rank 0, state dict)."""
    from m3dssd_amd import synth
    from m3dssd_amd.host.detect import detect_device, select_block
    from model.M3d_inference_align import build
    bf16 = dtype == "bf16"
    if steps is None:
        steps = 60 if bf16 else 200
    B = batch if batch is not None else (64 if bf16 else PER_GPU_BATCH)
    conf = synth.synth_conf(CROP, 0, batch_size=B, device=str(dev))
    sd = synth.synth_state_dict(0)
    net = build(conf, "test")
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).set_compute_dtype(dtype)
    x = synth.synth_frames(B, CROP, 1234 + rank).to(dev)          # inputs resident in HBM before the timed region
    eng = net.engine()

    def step():
        block, counts = select_block(*detect_device(net, x, conf), conf)
        if world > 1:
            return mdist.gather_block(block)
        return block[:, :-1], counts

    # ---- warm-up: first step builds the plan; one fully instrumented step finds the dominant kernel ----
    for _ in range(max(1, args.warmup - 1)):
        step()
    eng.profile = []
    step()
    torch.cuda.synchronize()
    eng.flush_profile()
    if dump_layers and rank == 0:
        with open(dump_layers, "w") as f:
            f.write("name,kernel,gflop,ms,tflops\n")
            for name, kind, flops, ms in eng.profile:
                f.write("%s,\"%s\",%.3f,%.4f,%.1f\n" % (name, kind, flops / 1e9, ms, flops / 1e9 / max(ms, 1e-6)))
    if args.dump_launches and rank == 0:
        seq = []
        for name, kind, flops, ms in eng.profile:
            if kind.startswith(MFMA_FAMILIES):
                seq.append([kind, kernel_symbol(kind).split("(", 1)[0].split("<", 1)[0].replace("void ", "").strip()])
                if kind.startswith("bf16_dcn_patch"):        # the gated implicit-GEMM fallback is launched right behind it
                    seq.append([kind + "/fallback", "bf16_conv_kernel"])
        json.dump(seq, open(args.dump_launches, "w"))
    per_kind = {}
    for name, kind, flops, ms in eng.profile:
        a = per_kind.setdefault(kind, [0.0, 0.0, 0])
        a[0] += ms
        a[1] += flops
        a[2] += 1
    # MFMA-bound kernel families (everything else is a small HBM/latency-bound helper)
    igemm = {k: v for k, v in per_kind.items() if k.startswith(MFMA_FAMILIES)}
    # The dominant KERNEL is chosen by symbol, not by engine family: one symbol serves several families (conv_wave_kernel<true, 4>
    # = the full-size deformable layers and their three split-K forms), and the headline roofline belongs to the symbol with the
    # largest share of GPU time (VERDICT r4 #4 / #3a: by family the second-largest kernel won).
    by_symbol = {}
    for k, v in igemm.items():
        by_symbol.setdefault(kernel_symbol(k), []).append(k)
    dom_symbol = max(by_symbol, key=lambda sy: sum(igemm[k][0] for k in by_symbol[sy]))
    dom_kinds = sorted(by_symbol[dom_symbol], key=lambda k: -igemm[k][0])
    dominant = dom_kinds[0]                     # the largest family of the dominant symbol (labels, Winograd divisor)
    gpu_ms_all = sum(v[0] for v in per_kind.values())
    breakdown = {k: round(v[0], 3) for k, v in sorted(per_kind.items(), key=lambda kv: -kv[1][0])}

    # ---- roofline pass: a few eager steps with HIP events around the dominant kernel's launches ---------
    eng.profile, eng.profile_kinds = [], set(dom_kinds)
    ROOF_REPS = 5
    for _ in range(ROOF_REPS):
        step()
    torch.cuda.synchronize()
    eng.flush_profile()
    roof_rows = eng.profile
    eng.profile, eng.profile_kinds = None, None

    # ---- timed region: K steps.  The launch-bound sequence (~170 launches) is captured once in a hipGraph
    # and replayed; the graph contains exactly the work of step() (forward, bundle, top-k, decode, NMS, and
    # for N > 1 the all-gather runs after each replay).
    use_graph = not args.no_graph
    use_pipe = use_graph and not args.no_pipeline
    flush = None
    if use_pipe:
        # throughput mode (m3dssd_amd/pipeline.py): one graph replay = forward(batch k) || detect(batch k-1); the K timed
        # steps submit K batches and the final flush() inside the timed region drains the last one, so exactly K
        # batches are forwarded AND detected in the measured time.
        from m3dssd_amd.pipeline import PipelinedDetector
        pipe = PipelinedDetector(net, conf, B, CROP[0], CROP[1])
        pipe.input.copy_(x)

        def timed_step():
            r = pipe.step(as_block=True)
            if world > 1 and r is not None:
                return mdist.gather_block(r[0])
            return r

        def flush():
            r = pipe.flush(as_block=True)
            if world > 1 and r is not None:
                return mdist.gather_block(r[0])
            return r
    elif use_graph:
        graph = torch.cuda.CUDAGraph()
        cap_stream = torch.cuda.Stream()
        cap_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap_stream):
            select_block(*detect_device(net, x, conf), conf)  # warm the side stream (allocator pools)
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=cap_stream):
                g_block, g_counts = select_block(*detect_device(net, x, conf), conf)
        torch.cuda.current_stream().wait_stream(cap_stream)

        def timed_step():
            graph.replay()
            if world > 1:
                return mdist.gather_block(g_block)
            return g_block[:, :-1], g_counts
    else:
        timed_step = step
    # the W untimed warm-up steps, in the launch mode that is timed (graph replays), directly in front of the timed region: the
    # instrumented eager passes above end in host synchronisations, and a K = 20 region (the driver's command) that starts on a
    # chip just back from idle read 4 % slower per step than K = 200 (6.10 vs 5.85 ms, round 6)
    for _ in range(max(1, args.warmup)):
        timed_step()
    if flush is not None:
        flush()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        timed_step()
    if flush is not None:
        flush()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    # ---- shader clock under the step (outside the timed region): one wave on a side stream samples the cycle counter against the
    # 100 MHz wall clock over a 30 ms window while the step replays next to it
    sclk = None
    try:
        import ctypes
        from m3dssd_amd import _hip as _h
        probe = torch.zeros(4, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        _h.check(_h.lib().m3d_clock_probe(ctypes.c_void_p(probe.data_ptr()), 0.03, ctypes.c_void_p(side.cuda_stream)))
        for _ in range(max(3, int(0.05 / max(dt / steps, 1e-4)))):
            timed_step()
        if flush is not None:
            flush()
        torch.cuda.synchronize()
        pc = probe.cpu().tolist()
        if pc[3] > pc[1]:
            sclk = (pc[2] - pc[0]) / (pc[3] - pc[1]) * 0.1
    except Exception:
        sclk = None
    eng.profile = roof_rows
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # per launch of the dominant family: the MEDIAN of its HIP-event durations over the instrumented steps (an eager step now and
    # then catches a launch behind a host hiccup: the mean read 5 % above rocprofv3's average on one lease, the median agrees)
    by_name = {}
    for r in eng.profile:
        by_name.setdefault(r[0], []).append(r)
    dom_ms = sum(sorted(x[3] for x in rows)[len(rows) // 2] for rows in by_name.values())
    dom_flops = sum(rows[0][2] for rows in by_name.values())
    dom_n = len(by_name)
    eng.profile, eng.profile_kinds = None, None

    # algorithmic HBM bytes per launch, per MFMA kernel family (dominant one included)
    plan = eng.plan_for(B, CROP[0], CROP[1])
    alg = {}
    for op in plan.ops:
        ab = algorithmic_bytes(op)
        if ab is not None:
            a = alg.setdefault(op[1], [0.0, 0])
            a[0] += ab
            a[1] += 1
    dom_alg = [alg[k] for k in dom_kinds if k in alg]
    alg_bytes = int(sum(a[0] for a in dom_alg) / sum(a[1] for a in dom_alg)) if dom_alg else None
    peak_tf = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    from m3dssd_amd import _hip as _hl
    lib_hashes = _hl.lib_source_hashes()
    families = {}
    for k, (ms, fl, cnt) in sorted(igemm.items(), key=lambda kv: -kv[1][0]):
        div = 4.0 if k.startswith("wino44") else (2.25 if k.startswith("wino") else 1.0)
        tf = fl / (ms * 1e-3) / 1e12 / div if ms > 0 else 0.0
        tr, src, stale = pmc_traffic(k, lib_hashes)
        abl = int(alg[k][0] / alg[k][1]) if k in alg else None
        gbs = alg[k][0] / (ms * 1e-3) / 1e9 if (k in alg and ms > 0) else None
        families[k] = {"kernel": kernel_symbol(k), "launches_per_step": cnt, "ms_per_step": round(ms, 3),
                       "executed_tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / peak_tf, 3),
                       "algorithmic_bytes_per_launch": abl,
                       "algorithmic_gbs": round(gbs, 1) if gbs is not None else None,
                       "hbm_frac": round(gbs / PEAK_HBM_GBS, 3) if gbs is not None else None,
                       # which roof the family sits closer to (fraction of the MFMA peak vs fraction of the HBM peak of its
                       # algorithmic bytes): the larger fraction names the binding resource
                       "nearer_roof": ("hbm" if (gbs is not None and gbs / PEAK_HBM_GBS > tf / peak_tf) else "mfma"),
                       "traffic": tr, "traffic_ratio": round(tr / abl, 2) if (tr and abl) else None,
                       "traffic_source": src, "traffic_stale": stale}
    # the dominant symbol's traffic: launch-weighted mean over its families
    dom_tr = [(families[k]["traffic"], families[k]["launches_per_step"]) for k in dom_kinds if families[k]["traffic"]]
    dom_traffic = int(sum(t * n for t, n in dom_tr) / sum(n for _, n in dom_tr)) if dom_tr else None
    dom_stale = any(families[k]["traffic_stale"] for k in dom_kinds if families[k]["traffic"]) if dom_tr else None

    rccl = None
    if world > 1:
        rccl = rccl_proof(args, mdist, net, conf, B, rank, world, dev, step())
    # the launches outside the MFMA families (stem, level0's own F(4x4) kernel, pooling, up-sampling, bundling, ...): each against
    # the roof that bounds it -- HBM bytes it has to move once / HIP-event time, and the arithmetic rate where it executes FLOPs
    helpers = {}
    for k, (ms, fl, cnt) in sorted(per_kind.items(), key=lambda kv: -kv[1][0]):
        if k in igemm:
            continue
        div = 4.0 if k.startswith("wino44") else 1.0
        gbs = alg[k][0] / (ms * 1e-3) / 1e9 if (k in alg and ms > 0) else None
        tf = fl / (ms * 1e-3) / 1e12 / div if (fl > 0 and ms > 0) else None
        helpers[k] = {"launches_per_step": cnt, "ms_per_step": round(ms, 3),
                      "algorithmic_bytes_per_launch": int(alg[k][0] / alg[k][1]) if k in alg else None,
                      "algorithmic_gbs": round(gbs, 1) if gbs is not None else None,
                      "hbm_frac_of_peak": round(gbs / PEAK_HBM_GBS, 3) if gbs is not None else None,
                      "executed_tflops": round(tf, 2) if tf is not None else None,
                      "frac_of_mfma_peak": round(tf / peak_tf, 3) if (tf is not None and k.startswith("wino44")) else None,
                      "bound": "hbm" if (gbs is not None and (tf is None or gbs / PEAK_HBM_GBS >= (tf or 0) / peak_tf)) else
                               ("mfma" if k.startswith("wino44") else ("valu" if tf is not None else "latency"))}
    mfma_ms = sum(v["ms_per_step"] for v in families.values())
    mfma_weighted = sum(v["ms_per_step"] * v["frac_of_mfma_peak"] for v in families.values()) / mfma_ms if mfma_ms > 0 else 0.0

    if rank == 0:
        value = world * B * steps / dt
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        wino_div = 4.0 if dominant.startswith("wino44") else (2.25 if dominant.startswith("wino") else 1.0)
        out = {
            "metric": ("images/sec at 1280x384 bs=8, 1/2/4/8 MI355X; 3D-box Linf vs ref" if not bf16 else
                       "images/sec at 1280x384 bs=64 bf16 (BASELINE.json configs[2]), 1/2/4/8 MI355X"),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: full M3d_inference_align (DLA-34 + DCNv2 align + ANAB) "
                                   "forward + decode + top-3000 + NMS, bs=%d/GPU, 1280x384, %s, random-init "
                                   "synthetic weights and frames" % (3 if B == 32 else (2 if bf16 else 1), B,
                                                                     "bf16 storage + bf16 MFMA, fp32 "
                                                                     "accumulation / epilogues" if bf16 else "fp32"),
                       "per_gpu_batch": B, "global_batch": B * world, "resolution": [CROP[1], CROP[0]],
                       "parallelism": "dp%d (batch sharded, 1 all-gather of [B,40,14] detections)" % world},
            "launch": ("hipGraph replay, detect(k-1) overlapped with forward(k); the timed graph decodes from the planar head outputs (sort keys "
                       "written by anchor_select) where the eager per-kernel breakdown below shows `bundle`" if use_pipe else
                       "hipGraph replay" if use_graph else "eager"),
            # Winograd F(2x2,3x3) launches: `achieved` counts the MFMA FLOPs the kernel EXECUTES (16 multiplies per 2x2 output
            # tile and channel pair = the direct-convolution count / 2.25), so frac is the MFMA-pipe utilisation and cannot
            # exceed 1; the direct-convolution-equivalent rate is reported next to it.
            "roofline": {"bound": "mfma", "kernel": kernel_symbol(dominant), "achieved": round(achieved / wino_div, 2),
                         "peak": peak_tf, "unit": "TFLOP/s",
                         "frac": round(achieved / wino_div / peak_tf, 4),
                         "note": (("achieved = MFMA FLOPs executed by the Winograd F(4x4,3x3) launches (2*36*Cin*Cout per 4x4 output "
                                   "tile = direct-convolution FLOPs of SURVEY 8d / 4) / HIP-event time; peak = dense fp32 MFMA at "
                                   "the 2.4 GHz boost clock") if wino_div == 4.0 else
                                  ("achieved = MFMA FLOPs executed by the Winograd F(2x2,3x3) launches (2*16*Cin*Cout per 2x2 output "
                                   "tile = direct-convolution FLOPs of SURVEY 8d / 2.25) / HIP-event time; peak = dense fp32 MFMA at "
                                   "the 2.4 GHz boost clock")) if wino_div > 1 else
                                 "achieved = algorithmic FLOPs of the launches / HIP-event time",
                         "direct_conv_equivalent_tflops": round(achieved, 2),
                         "traffic": dom_traffic, "traffic_source": families[dominant]["traffic_source"],
                         "traffic_stale": dom_stale,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC passes committed under profiles/, not this run; "
                                         "traffic_stale = the loaded library was built from other sources of this kernel than the profiled one)",
                         "families_of_symbol": dom_kinds,
                         "algorithmic_bytes_per_launch": alg_bytes, "launches_timed": dom_n, "timing": "median of %d HIP-event measurements per launch" % ROOF_REPS,
                         "algorithmic_gbs": round(alg_bytes / (dom_ms / max(dom_n, 1) * 1e-3) / 1e9, 1) if alg_bytes and dom_ms > 0 else None,
                         "hbm_frac_of_peak": round(alg_bytes / (dom_ms / max(dom_n, 1) * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if alg_bytes and dom_ms > 0 else None,
                         "avg_launch_ms": round(dom_ms / max(dom_n, 1), 4),
                         "avg_launch_gflop": round(dom_flops / max(dom_n, 1) / 1e9, 3),
                         "share_of_gpu_time": round(sum(igemm[k][0] for k in dom_kinds) / gpu_ms_all, 3)},
            "gpu_ms_by_kernel_one_step": breakdown,
            "mfma_kernel_families": families,
            "mfma_time_weighted_frac": round(mfma_weighted, 4),
            "helper_kernels": helpers,
            # the clock the chip held while the step replayed (s_memtime / s_memrealtime of a probe wave): `peak` above is the
            # 2.4 GHz datasheet figure; peak x sclk / 2.4 is what the matrix pipes offered in this run
            "sclk_under_step_ghz": round(sclk, 3) if sclk else None,
            "roofline_frac_at_held_clock": round(achieved / wino_div / (peak_tf * sclk / 2.4), 4) if sclk else None,
        }
        out["work_in_timed_region"] = (
            "forward + decode + top-%d + NMS + select of K batches; bundle_outputs %s" %
            (int(conf.nms_topN_pre), "NOT run (decode reads the planar head outputs; `dropin` is the bundled reference call sequence)"
             if use_pipe else "run"))
        if world == 1 and not args.no_dropin:
            try:                                   # (side leg: never part of `value`)
                out["dropin"] = dropin_leg(net, x, conf, args.dropin_steps or max(5, min(steps, 60 if not bf16 else 30)))
                out["dropin"]["vs_headline_graph"] = round(out["dropin"]["graph"]["value"] / value, 4)
            except Exception as e:                 # noqa: BLE001
                out["dropin"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if feed:
            try:                                   # (side leg: never part of `value`)
                out["feed_u8"] = feed_u8_leg(net, conf, B, dev, steps, 1e3 * dt / steps)
            except Exception as e:                 # noqa: BLE001
                out["feed_u8"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
        if rccl is not None:
            out["rccl"] = rccl
            if not rccl["shards_match"]:
                raise SystemExit("bench.py: the gathered block does not hold rank r's detections at rows [r*B, (r+1)*B)")
        step_s = dt / steps
        if not bf16:
            # whole-step figures against both roofs.  In fp32 the network is MFMA-bound (~97 FLOP/B vs a ridge of ~20);
            # algorithmic TFLOP/s may exceed executed TFLOP/s because the Winograd layers execute 1/4 (1/2.25) of their
            # direct-convolution FLOPs.
            alg_b = B * ALG_MB_PER_IMAGE_F32 * 1e6 + ALG_MB_WEIGHTS_F32 * 1e6
            out["step_roofline"] = {
                "algorithmic_tflops": round(B * ALG_GFLOP_PER_IMAGE * 1e9 / step_s / 1e12, 1),
                "mfma_frac": round(B * ALG_GFLOP_PER_IMAGE * 1e9 / step_s / 1e12 / peak_tf, 4),
                "executed_tflops_mfma_families": round(sum(v["executed_tflops"] * v["ms_per_step"] for v in families.values())
                                                       / (step_s * 1e3), 1),
                "mfma_time_weighted_frac": round(mfma_weighted, 4),
                "algorithmic_gbs_f32": round(alg_b / step_s / 1e9, 1), "hbm_frac": round(alg_b / step_s / 1e9 / PEAK_HBM_GBS, 4),
                "counter_bytes_per_step": step_counter_bytes(families, helpers),
                "note": "SURVEY 8d algorithmic work per image: 105.8 GFLOP; 1003.6 MB fp32 activations + 21 MB outputs + 5.9 MB input, "
                        "weights 82.6 MB per batch; counter_bytes_per_step = sum over the kernel families of launches x PMC bytes per "
                        "launch from the committed passes under profiles/ (families without a pass counted at their algorithmic bytes)"}
        if bf16:
            # whole-step figures against both roofs (in bf16 the network is HBM-bound unless fused)
            out["step_roofline"] = {
                "algorithmic_tflops": round(B * 105.8e9 / step_s / 1e12, 1), "mfma_frac": round(B * 105.8e9 / step_s / 1e12 / peak_tf, 4),
                "algorithmic_gbs_bf16": round((B * (501.8e6 + 21e6 + 5.9e6) + 41.3e6) / step_s / 1e9, 1),
                "hbm_frac": round((B * (501.8e6 + 21e6 + 5.9e6) + 41.3e6) / step_s / 1e9 / PEAK_HBM_GBS, 4),
                "note": "SURVEY 8d algorithmic work per image: 105.8 GFLOP; 501.8 MB bf16 activations + 21 MB outputs + 5.9 MB input, "
                        "weights 41.3 MB (bf16) per batch"}
        return out, sd
    return None, sd


if __name__ == "__main__":
    main()
