"""rc_engine_set_lanes(1) against (2) on several bench workloads (256 frames per step, frames resident, uniform noise; MASK=1 renders
crt-royale's mask; LANES="2" runs one setting only, e.g. under rocprofv3): frames/s each way.  Run on the GPU box from the repo root:  python3 profiles/dev_lanes.py
"""
import os, sys, time, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, chain_specs
import bench
from retrocapture_amd.engine import ShaderEngine

tmp = tempfile.mkdtemp(); tree = chain_specs.write_tree(tmp)
N = 256
for wl in os.environ.get("WORKLOADS", "crt-royale crt-royale-fake-bloom crt-hyllian-glow crt-pi zfast-crt crt-easymode ntsc").split():
    key, w, h, vw, vh, _ = bench.WORKLOADS[wl]
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    frames = torch.randint(0, 256, (N, h, w, 4), dtype=torch.uint8, device="cuda", generator=g); frames[..., 3] = 255
    e = ShaderEngine(); assert e.init(0, torch.cuda.current_stream().cuda_stream)
    e.setAllowMissingSources(True); assert e.loadPresetStatus(tree[key]) == 0
    e.setViewport(vw, vh)
    if os.environ.get("MASK"): e.setUndefinedVaryingZero(True)
    res = []
    for lanes in [int(v) for v in os.environ.get("LANES", "1 2").split()]:
        e.setLanes(lanes)
        for _ in range(3): e.applyShaderBatch(frames, N, w, h)
        torch.cuda.synchronize()
        steps = 30
        t0 = time.perf_counter()
        for _ in range(steps): e.applyShaderBatch(frames, N, w, h)
        torch.cuda.synchronize()
        res.append(N * steps / (time.perf_counter() - t0))
    if len(res) == 2:
        print("%-24s one lane %9.0f   two lanes %9.0f   %+5.1f %%" % (wl, res[0], res[1], (res[1] / res[0] - 1) * 100), flush=True)
    else:
        print("%-24s LANES=%s %9.0f frames/s" % (wl, os.environ.get("LANES"), res[0]), flush=True)
    e.shutdown(); del frames
