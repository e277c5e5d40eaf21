"""Experiment: the headline batch (256 frames of 1080p through crt-royale) on ONE engine and stream against TWO engine instances
on two HIP streams (128 frames each, enqueued back to back from one host thread), to see whether kernels of the two streams
overlap (HBM-bound passes of one under the VALU / LDS-bound passes of the other).  Run on the GPU box from the repo root:
    python3 profiles/dev_two_streams.py
"""
import os, sys, time, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, chain_specs
from retrocapture_amd.engine import ShaderEngine

tmp = tempfile.mkdtemp(); tree = chain_specs.write_tree(tmp)
W, H, N = 1920, 1080, 256
g = torch.Generator(device="cuda"); g.manual_seed(1234)
frames = torch.randint(0, 256, (N, H, W, 4), dtype=torch.uint8, device="cuda", generator=g); frames[..., 3] = 255

def make(stream):
    e = ShaderEngine()
    assert e.init(0, stream.cuda_stream)
    e.setAllowMissingSources(True)
    assert e.loadPresetStatus(tree["crt-royale"]) == 0
    e.setViewport(W, H)
    return e

def timed(fn, steps=60, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return N * steps / (time.perf_counter() - t0)

s0 = torch.cuda.current_stream()
e0 = make(s0)
print("one engine, one stream      : %.0f frames/s" % timed(lambda: e0.applyShaderBatch(frames, N, W, H)), flush=True)
for parts in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    engines = [make(s) for s in streams]
    per = N // parts
    chunks = [frames[i * per:(i + 1) * per] for i in range(parts)]
    for e in engines: e.setChunkFrames(per if per <= 128 else 128)
    def fn():
        for e, c in zip(engines, chunks):
            e.applyShaderBatch(c, per, W, H)
    print("%d engines on %d streams       : %.0f frames/s" % (parts, parts, timed(fn)), flush=True)
    del engines

# the two halves rendered by two engines on two streams at once are the bytes one engine renders (fresh engines: the frame
# counter - crt-royale's field parity - starts at the same parity for frame 128 either way)
import numpy as np
ref = make(s0)
ref.applyShaderBatch(frames, N, W, H); ref.sync()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ea, eb = make(sa), make(sb)
ea.applyShaderBatch(frames[:128], 128, W, H); eb.applyShaderBatch(frames[128:], 128, W, H)
ea.sync(); eb.sync()
last = ref.passCount() - 1
bad = 0
for k in (0, 1, 63, 127):
    bad += int(np.count_nonzero(ref.readPass(last, k) != ea.readPass(last, k)))
    bad += int(np.count_nonzero(ref.readPass(last, 128 + k) != eb.readPass(last, k)))
print("two engines vs one: %d bytes differ in 8 frames" % bad, flush=True)
