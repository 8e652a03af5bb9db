"""Diagnostic: where the host side of the feed spends its time (collate, pinned staging, H2D) for one 128-px array."""
import time, torch
x8 = torch.randint(0, 256, (78, 64, 3, 128, 128), dtype=torch.uint8)
xc = torch.rand(2000, 78, 64)
print("threads", torch.get_num_threads())
def t(fn, n=5):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
print("stack u8 -> pageable      %.1f ms" % t(lambda: torch.stack([x8])))
pin8 = torch.empty((1,) + x8.shape, dtype=torch.uint8, pin_memory=True)
print("stack u8 -> pinned (out=) %.1f ms" % t(lambda: torch.stack([x8], out=pin8)))
print("copy_ u8 -> pinned        %.1f ms" % t(lambda: pin8[0].copy_(x8)))
pinc = torch.empty((1,) + xc.shape, pin_memory=True)
print("stack f32 counts -> pinned %.1f ms" % t(lambda: torch.stack([xc], out=pinc)))
dev = torch.empty(pin8.shape, dtype=torch.uint8, device='cuda')
def h2d():
    dev.copy_(pin8, non_blocking=True); torch.cuda.synchronize()
print("H2D 245 MB pinned         %.1f ms" % t(h2d))
pg = torch.stack([x8])
def h2d_pg():
    dev.copy_(pg); torch.cuda.synchronize()
print("H2D 245 MB pageable       %.1f ms" % t(h2d_pg))
import numpy as np
a = x8.numpy(); b = pin8[0].numpy()
print("numpy copyto -> pinned    %.1f ms" % t(lambda: np.copyto(b, a)))
