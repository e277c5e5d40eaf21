import numpy as np, sys, torch
sys.path[:0]=['/root/repo','/root/repo/tests']
from test_frame_io import o_ingest
from retrocapture_amd import engine
rng=np.random.default_rng(5)
w,h,n=64,4,1
src=rng.integers(0,256,n*h*w*2,dtype=np.uint8)
d_src=torch.from_numpy(src).cuda(); d_dst=torch.zeros(n*h*w*4,dtype=torch.uint8,device='cuda')
engine.ingest(d_src,'yuyv422',w,h,n,d_dst); torch.cuda.synchronize()
got=d_dst.cpu().numpy().reshape(-1,4); want=o_ingest(src,'yuyv422',n*h*w).reshape(-1,4)
bad=np.argwhere((got!=want).any(1)).ravel()
print(len(bad), bad[:10])
for p in bad[:6]:
    m=src[4*(p//2):4*(p//2)+4]; print(p, m, got[p], want[p])
