"""Probe: does running the image classifier of step i+1 on a side stream hide the g / count-f / CE / backward work of step i?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
import bench
from gridnext_amd import functional as GF
from gridnext_amd.gridnet_models import GridNetHexOddr
from gridnext_amd.synthetic import visium_array
dev = torch.device('cuda:0')
model = bench.build_model(dev)
for p in model.patch_classifier.parameters():
    p.requires_grad = False
opt = torch.optim.Adam(model.corrector.parameters(), lr=1e-3)
arrays = []
for a in range(2):
    xi, xc, y = visium_array(a, 2000, 8, 128, device=dev)
    arrays.append((xi.unsqueeze(0), xc.unsqueeze(0), y.unsqueeze(0)))
model.train(); model.patch_classifier.eval()
f_img = model.image_classifier

def g_step(rows_img, xc, y):
    model._set_mode('count')
    g_count = GridNetHexOddr._grid_nhwc(model, xc)
    model._set_mode('image'); model._set_mode('concat')
    grid = torch.cat((g_count, rows_img.reshape(1, 78, 64, 8)), dim=3)
    logits = model._correct_nhwc(grid)
    loss, stats, _ = GF.masked_cross_entropy(logits.reshape(-1, 8), y, 1)
    loss.backward(); opt.step(); opt.zero_grad()
    return loss

def f_rows(xi):
    with torch.no_grad():
        return f_img(xi.reshape(-1, 3, 128, 128))

def run_serial(n):
    for i in range(n):
        xi, xc, y = arrays[i % 2]
        g_step(f_rows(xi), xc, y)

side = torch.cuda.Stream()
def run_overlap(n):
    main = torch.cuda.current_stream()
    rows = f_rows(arrays[0][0])
    for i in range(n):
        xi, xc, y = arrays[i % 2]
        nxt = None
        if i + 1 < n:
            side.wait_stream(main)                     # (the inputs are resident; nothing to wait for but ordering)
            with torch.cuda.stream(side):
                nxt = f_rows(arrays[(i + 1) % 2][0])
        g_step(rows, xc, y)
        if nxt is not None:
            main.wait_stream(side)
            nxt.record_stream(main)
        rows = nxt

for fn in (run_serial, run_overlap, run_serial, run_overlap):
    fn(5); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(20); torch.cuda.synchronize()
    print(fn.__name__, "%.2f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
