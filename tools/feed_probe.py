"""Where does the fed-input leg lose time against the resident one?   python tools/feed_probe.py [steps]
(a) resident fp32 frames, one graph; (b) resident uint8 frames, the two graphs alternating, nothing uploaded; (d) the full fed
form (bench.py's feed_u8 leg: the upload of batch k + 1 is a kernel on the side branch of graph k); (e) a serial hipMemcpyAsync
in front of every replay.  Round 4, first version (hipMemcpyAsync on a copy stream, events both ways): (d) was 6.47-6.60 ms against
6.13 / 6.06 resident and 6.36-6.44 serial -- the asynchronous copy cost MORE than the serial one."""
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
j = [0]
pu.feed(pool[0])


def fed():
    pu.feed(pool[(j[0] + 1) & 3])
    j[0] += 1
    pu.step_fed(as_block=True)
print("(d) fed uint8 (in-graph upload kernel of batch k+1 || forward k): %.3f ms" % timed(fed, steps), flush=True)
torch.cuda.synchronize()
# (e) the same frames by hipMemcpyAsync on the MAIN stream in front of the graph (serial)
buf = pu.inputs_u8[0]
pu._slots.zero_()


def serial():
    buf.copy_(pool[j[0] & 3], non_blocking=True)
    j[0] += 1
    pu._graphs[0].replay()
print("(e) hipMemcpyAsync on the main stream, serial:    %.3f ms" % timed(serial, steps), flush=True)
