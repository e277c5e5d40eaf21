"""Per-pass time of the crt-royale chain (one lane, mask rendered) on three kinds of frames: uniform noise (every table gather of a
wave hits 64 different entries), a constant colour (every gather a broadcast) and a two-colour checkerboard.  Development:
how much of a pass is the bank conflicts of its gathers?   python3 profiles/dev_frame_statistics.py"""
import os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, chain_specs
from gpu_util import make_engine
tree = chain_specs.write_tree(tempfile.mkdtemp())
W, H, N = 1920, 1080, 128
g = torch.Generator(device="cuda"); g.manual_seed(7)
kinds = {"noise": torch.randint(0, 256, (N, H, W, 4), dtype=torch.uint8, device="cuda", generator=g)}
kinds["constant"] = torch.full((N, H, W, 4), 0, dtype=torch.uint8, device="cuda"); kinds["constant"][..., 0] = 180; kinds["constant"][..., 1] = 97; kinds["constant"][..., 2] = 33
cb = torch.zeros((N, H, W, 4), dtype=torch.uint8, device="cuda")
yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
cb[:] = torch.where(((yy + xx) % 2 == 0)[None, ..., None], torch.tensor([200, 60, 120, 255], dtype=torch.uint8, device="cuda"), torch.tensor([40, 170, 90, 255], dtype=torch.uint8, device="cuda"))
kinds["checkerboard"] = cb
e = make_engine(tree['crt-royale'], W, H)
e.setLanes(1)
e.setUndefinedVaryingZero(True)
for name, fr in kinds.items():
    for _ in range(2): e.applyShaderBatch(fr, N, W, H)
    e.sync()
    e.setProfiling(True)
    for _ in range(3): e.applyShaderBatch(fr, N, W, H)
    e.sync()
    prof = [e.passProfile(i) for i in range(e.passCount())]
    e.setProfiling(False)
    print(name, [round(q["total_ms"] / max(1, q["frames"]) * 1e3, 2) for q in prof], flush=True)
e.shutdown()
