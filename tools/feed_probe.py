"""Where does the fed-input leg lose time against the resident one?   python tools/feed_probe.py [steps]
(a) resident fp32 frames, one graph; (b) resident uint8 frames, the two graphs alternating, nothing uploaded; (c) the event
protocol of feed() / step_fed() without the copy; (d) the full fed form (bench.py's feed_u8 leg)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from m3dssd_amd import synth                                  # noqa: E402
from m3dssd_amd.pipeline import PipelinedDetector             # noqa: E402
from model.M3d_inference_align import build                   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
CROP, B, dev = (384, 1280), 8, torch.device("cuda:0")
conf = synth.synth_conf(CROP, 0, batch_size=B, device="cuda:0")
net = build(conf, "test")
net.load_state_dict(synth.synth_state_dict(0))
net = net.to(dev)
fh, fw = 375, 1242
rng = np.random.RandomState(7)
pool = [torch.from_numpy(rng.randint(0, 256, size=(B, fh, fw, 3)).astype(np.uint8)).pin_memory() for _ in range(4)]


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


pf = PipelinedDetector(net, conf, B, CROP[0], CROP[1])
pf.input.copy_(synth.synth_frames(B, CROP, 1234).to(dev))
print("(a) resident fp32, one graph:            %.3f ms" % timed(lambda: pf.step(as_block=True), steps), flush=True)
pu = PipelinedDetector(net, conf, B, CROP[0], CROP[1], u8_frame=(fh, fw))
pu.inputs_u8[0].copy_(pool[0])
pu.inputs_u8[1].copy_(pool[1])
torch.cuda.synchronize()
k = [0]


def alt():
    pu._graphs[k[0] & 1].replay()
    k[0] += 1
print("(b) resident uint8, two graphs alternate: %.3f ms" % timed(alt, steps), flush=True)
print("    resident uint8, graph 0 only:         %.3f ms" % timed(lambda: pu._graphs[0].replay(), steps), flush=True)
main = torch.cuda.current_stream(dev)


def proto():
    i = k[0] & 1
    k[0] += 1
    with torch.cuda.stream(pu._copy_stream):
        pu._copy_stream.wait_event(pu._done[i])
        pu._ready[i].record(pu._copy_stream)
    main.wait_event(pu._ready[i ^ 1])
    pu._graphs[i ^ 1].replay()
    pu._done[i ^ 1].record(main)
pu._done[0].record(main); pu._done[1].record(main); pu._ready[0].record(main); pu._ready[1].record(main)
print("(c) event protocol, no copy:              %.3f ms" % timed(proto, steps), flush=True)
pu._fed, pu._next_buf, pu._pending = [], 0, False
pu.feed(pool[0])
j = [0]


def fed():
    pu.feed(pool[j[0] & 3])
    j[0] += 1
    pu.step_fed(as_block=True)
print("(d) fed uint8 (upload k+1 || graph k):    %.3f ms" % timed(fed, steps), flush=True)
# (e) the same with the upload issued on the MAIN stream in front of the graph (serial, no events)
buf = pu.inputs_u8[0]


def serial():
    buf.copy_(pool[j[0] & 3], non_blocking=True)
    j[0] += 1
    pu._graphs[0].replay()
print("(e) upload on the main stream, serial:    %.3f ms" % timed(serial, steps), flush=True)

# (f) zero-copy: the two input buffers are pinned host memory, the stem reads them over PCIe
pz = PipelinedDetector(net, conf, B, CROP[0], CROP[1], u8_frame=(fh, fw), u8_zero_copy=True)
pz.inputs_u8[0].copy_(pool[0]); pz.inputs_u8[1].copy_(pool[1])
kz = [0]


def altz():
    pz._graphs[kz[0] & 1].replay()
    kz[0] += 1
print("(f) zero-copy, graphs alternate, no host write: %.3f ms" % timed(altz, steps), flush=True)
torch.cuda.synchronize()
pz._fed, pz._next_buf, pz._pending, pz._used = [], 0, False, [False, False]
pz.feed(pool[0])
jz = [0]


def fedz():
    pz.feed(pool[jz[0] & 3])
    jz[0] += 1
    pz.step_fed(as_block=True)
print("(g) zero-copy fed (host memcpy of a fresh frame set every step): %.3f ms" % timed(fedz, steps), flush=True)
