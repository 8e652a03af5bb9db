import sys, time, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, '/root/repo')
import gridnext_amd as ga
from gridnext_amd import functional as GF
DEV='cuda:0'
torch.manual_seed(0)
cfg = dict(num_classes=8, small_inputs=False, growth_rate=32, block_config=(6,12,24,16), num_init_features=64, bn_size=4)
f = ga.DenseNet(**cfg).to(DEV).eval()
x = torch.rand(32,3,128,128, device=DEV)
with torch.no_grad():
    for _ in range(2): f(x)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f(x)
    torch.cuda.current_stream().wait_stream(s)
    print('side stream ok', flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = f(x)
    print('captured eval fwd', flush=True)
    g.replay(); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); print('graph replay eval fwd ms', (time.perf_counter()-t)/20*1e3, flush=True)
    t=time.perf_counter()
    for _ in range(20): f(x)
    torch.cuda.synchronize(); print('eager eval fwd ms', (time.perf_counter()-t)/20*1e3, flush=True)
# training fwd+bwd manual capture
f.train()
y = torch.randint(0,8,(32,), device=DEV)
def fb():
    out = f(x); loss,_,_ = GF.masked_cross_entropy(out, y, 1, label_base=0); loss.backward(); return loss
for _ in range(3):
    f.zero_grad(set_to_none=True); fb()
torch.cuda.synchronize(); print('eager train ok', flush=True)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        f.zero_grad(set_to_none=True); fb()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
print('side-stream train ok', flush=True)
g2 = torch.cuda.CUDAGraph()
f.zero_grad(set_to_none=True)
with torch.cuda.graph(g2):
    loss = fb()
print('captured train fwd+bwd', flush=True)
g2.replay(); torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(20): g2.replay()
torch.cuda.synchronize(); print('graph replay train fwd+bwd ms', (time.perf_counter()-t)/20*1e3, float(loss), flush=True)
