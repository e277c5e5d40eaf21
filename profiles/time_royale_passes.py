"""Per-pass device time of crt-royale at 1920x1080 (batch 8): MASK=1 renders the mask passes, PROF=1 prints microseconds
per frame for passes 0..11 (engine pass profile), REPS = untimed repetitions, PARAMS="name=value,..." sets shader
parameters (e.g. geom_mode_runtime=1 for the curved last pass).  Run from the repo root on the GPU box:
    PROF=1 python3 profiles/time_royale_passes.py
"""
import sys, os, tempfile
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, torch, chain_specs
from gpu_util import make_engine, run_engine, to_device_rgba
tmp=tempfile.mkdtemp(); tree=chain_specs.write_tree(tmp)
e=make_engine(tree['crt-royale'],1920,1080)
if os.environ.get('MASK'): e.setUndefinedVaryingZero(True)
for kv in filter(None, os.environ.get('PARAMS','').split(',')):
    k,v=kv.split('='); assert e.setShaderParameter(k,float(v))
fr=np.random.default_rng(0).integers(0,256,(8,1080,1920,3),dtype=np.uint8)
d=to_device_rgba(fr)
for _ in range(int(os.environ.get('REPS','3'))):
    e.applyShaderBatch(d,8,1920,1080)
e.sync()
if os.environ.get('PROF'):
    e.setProfiling(True)
    for _ in range(10): e.applyShaderBatch(d,8,1920,1080)
    print([round(e.passProfile(i)['total_ms']/max(1,e.passProfile(i)['frames'])*1000,1) for i in range(12)])
