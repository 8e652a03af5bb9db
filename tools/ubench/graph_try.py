import sys, time, torch, torch.nn as nn
sys.path.insert(0, '/root/repo')
import gridnext_amd as ga
from gridnext_amd import functional as GF
DEV='cuda:0'
torch.manual_seed(0)
f = ga.DenseNet(num_classes=8, small_inputs=False, growth_rate=32, block_config=(6,12,24,16), num_init_features=64, bn_size=4).to(DEV).train()
x = torch.rand(32,3,128,128, device=DEV); y = torch.randint(0,8,(32,), device=DEV)
opt = torch.optim.Adam(f.parameters(), lr=1e-3)
def step_eager():
    opt.zero_grad()
    out = f(x); loss,_,_ = GF.masked_cross_entropy(out, y, 1, label_base=0); loss.backward(); opt.step(); return loss
for _ in range(3): l = step_eager()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): l = step_eager()
torch.cuda.synchronize(); print('eager ms/step', (time.perf_counter()-t)/10*1e3, float(l))
g = torch.cuda.make_graphed_callables(f, (x,))
def step_graph():
    opt.zero_grad()
    out = g(x); loss,_,_ = GF.masked_cross_entropy(out, y, 1, label_base=0); loss.backward(); opt.step(); return loss
for _ in range(3): l = step_graph()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): l = step_graph()
torch.cuda.synchronize(); print('graphed ms/step', (time.perf_counter()-t)/10*1e3, float(l))
