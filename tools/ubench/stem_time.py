import sys, os
sys.path.insert(0, '/root/repo')
import torch
from gridnext_amd import _lib as L
DEV='cuda:0'
n=4992
x=torch.rand(n,3,128,128,device=DEV); W=torch.randn(64,3,7,7,device=DEV)*0.1
sc=torch.rand(64,device=DEV)+0.5; sh=torch.randn(64,device=DEV)*0.2
out=torch.empty(n*32*32,256,device=DEV)
def run():
    L.call('gnx_conv_stem_bnrelu_maxpool', L.ptr(x), L.ptr(W), L.ptr(out), 256, n, 3,128,128,64,7,7,2,3,L.ptr(sc),L.ptr(sh),L.stream())
for _ in range(3): run()
torch.cuda.synchronize()
s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
print('fused stem ms', s.elapsed_time(e)/20)
