"""Diagnostic: BASELINE config 2 (DenseNet-121 @128 px, train_spotwise, batch 32) - steps for a rocprofv3 kernel trace."""
import contextlib, io, os, sys, time
import torch, torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gridnext_amd as ga
DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 32 * B
g = torch.Generator(device=DEV).manual_seed(0)
x = torch.rand((n + 2 * B, 3, 128, 128), generator=g, device=DEV)
y = torch.randint(0, 8, (n + 2 * B,), device=DEV)
dl = {'train': DataLoader(TensorDataset(x[:n], y[:n]), batch_size=B, shuffle=True),
      'val': DataLoader(TensorDataset(x[n:], y[n:]), batch_size=B)}
f = ga.DenseNet(num_classes=8, small_inputs=False, growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64,
                bn_size=4, drop_rate=0)
opt = torch.optim.Adam(f.parameters(), lr=1e-3)
with contextlib.redirect_stdout(io.StringIO()):
    ga.train_spotwise(f, {'train': DataLoader(TensorDataset(x[:2 * B], y[:2 * B]), batch_size=B), 'val': dl['val']},
                      nn.CrossEntropyLoss(), opt, num_epochs=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ga.train_spotwise(f, dl, nn.CrossEntropyLoss(), opt, num_epochs=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("batch %d: %.2f ms per step (34 steps incl. 2 val), %.0f spots/s" % (B, dt / 34 * 1e3, (n + 2 * B) / dt))
