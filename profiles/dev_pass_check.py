"""Development loop for the crt-royale kernels at 1920x1080 (run on the GPU box from the repo root):
every pass of the specialised forms against the general per-pixel forms (bytes, three frames: noise / smooth / bars, both
mask modes), then microseconds per frame and pass.  PASSES="10,1" limits the comparison printout; SKIPCMP=1 only times."""
import os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, chain_specs
from gpu_util import make_engine, run_engine, to_device_rgba
from test_royale_fullsize import frames3
tmp = tempfile.mkdtemp(); tree = chain_specs.write_tree(tmp)
W, H = int(os.environ.get('W', 1920)), int(os.environ.get('H', 1080))
bad_total = 0
if not os.environ.get('SKIPCMP'):
    fr = frames3() if (W, H) == (1920, 1080) else np.random.default_rng(1).integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    for mask in (False, True):
        e = make_engine(tree['crt-royale'], W, H)
        e.setUndefinedVaryingZero(mask)
        run_engine(e, fr)
        mine = [[e.readPass(i, k) for i in range(12)] for k in range(3)]
        e.setGeneralKernelsOnly(True)
        run_engine(e, fr)
        for i in range(12):
            bad = sum(int((e.readPass(i, k) != mine[k][i]).sum()) for k in range(3))
            bad_total += bad
            if bad:
                k = [k for k in range(3) if (e.readPass(i, k) != mine[k][i]).any()][0]
                d = np.argwhere((e.readPass(i, k) != mine[k][i]).any(-1))
                print("mask %d pass %d: %d bytes differ; frame %d first (y,x) %s rows %s cols %s" % (mask, i, bad, k, d[:4].tolist(), np.unique(d[:, 0])[:12].tolist(), np.unique(d[:, 1])[:12].tolist()))
        e.shutdown()
    print("specialised vs general: %d bytes differ in total" % bad_total)
for mask in (False, True):
    e = make_engine(tree['crt-royale'], W, H)
    e.setUndefinedVaryingZero(mask)
    NB = int(os.environ.get('BATCH', 8))   # frames per batch (= per launch while it does not exceed the engine's chunk)
    fr = np.random.default_rng(0).integers(0, 256, (NB, H, W, 3), dtype=np.uint8)
    d = to_device_rgba(fr)
    if NB > 8: e.setChunkFrames(NB)
    for _ in range(3): e.applyShaderBatch(d, NB, W, H)
    e.sync(); e.setProfiling(True)
    for _ in range(10): e.applyShaderBatch(d, NB, W, H)
    t = [round(e.passProfile(i)['total_ms'] / max(1, e.passProfile(i)['frames']) * 1000, 1) for i in range(12)]
    print("mask %d us/frame per pass %s  total %.1f" % (mask, t, sum(t)))
    e.shutdown()
sys.exit(1 if bad_total else 0)
