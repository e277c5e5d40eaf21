"""One crt-royale batch at a given size / mode (development: reproducing a failure in isolation).
   W=1872 H=1053 N=3 FOLD=0 LANES=1 MASK=1 python3 profiles/dev_case.py"""
import os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, chain_specs
from gpu_util import make_engine
tree = chain_specs.write_tree(tempfile.mkdtemp())
W, H, N = int(os.environ.get('W', 1920)), int(os.environ.get('H', 1080)), int(os.environ.get('N', 3))
g = torch.Generator(device="cuda"); g.manual_seed(81)
frames = torch.randint(0, 256, (N, H, W, 4), dtype=torch.uint8, device="cuda", generator=g)
e = make_engine(tree['crt-royale'], W, H)
e.setFoldPasses(os.environ.get('FOLD', '1') == '1')
e.setLanes(int(os.environ.get('LANES', 1)))
e.setUndefinedVaryingZero(os.environ.get('MASK', '0') == '1')
if os.environ.get('GENERAL'): e.setGeneralKernelsOnly(True)
e.applyShaderBatch(frames, N, W, H)
e.sync()
print("ok", W, H, N, "fold", os.environ.get('FOLD', '1'), "lanes", os.environ.get('LANES', 1), "checksum", int(e.readPass(e.passCount() - 1, 0).astype(np.int64).sum()))
e.shutdown()
