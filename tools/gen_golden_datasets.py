#!/usr/bin/env python3
"""Golden vectors for the file-backed count datasets: writes small synthetic Visium-style files under
tests/golden/files/ and records what the REFERENCE's CountDataset / CountGridDataset return for them
(tests/golden/count_datasets.npz).  Build container only (imports /root/reference, read-only).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_datasets.py
"""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True
from gridnext.count_datasets import CountDataset, CountGridDataset      # noqa: E402

FILES = os.path.join(ROOT, 'tests', 'golden', 'files')
os.makedirs(FILES, exist_ok=True)
H, W, G = 8, 6, 5
rng = np.random.RandomState(0)
names = ['Layer1', 'Layer2', 'WM']
genes = ['G%d' % i for i in range(G)]
arrays = []
for a in range(2):
    spots = []            # (barcode, array_row, array_col)
    for row in range(H):
        for x in range(W):
            if rng.rand() < 0.7:
                col = 2 * x + (row % 2)                     # Visium pseudo-hex column
                spots.append(('BC%d_%02d%02d-1' % (a, row, col), row, col))
    counts = rng.poisson(3.0, size=(G, len(spots)))
    cfile = os.path.join(FILES, 'array%d.counts.tsv' % a)
    with open(cfile, 'w') as fh:
        fh.write('\t'.join([''] + ['%d_%d' % (c, r) for _, r, c in spots]) + '\n')
        for gi, gname in enumerate(genes):
            fh.write('\t'.join([gname] + [str(v) for v in counts[gi]]) + '\n')
    pfile = os.path.join(FILES, 'array%d.tissue_positions.csv' % a)
    with open(pfile, 'w') as fh:
        fh.write('barcode,in_tissue,array_row,array_col,pxl_row_in_fullres,pxl_col_in_fullres\n')
        for bc, r, c in spots:
            fh.write('%s,1,%d,%d,%d,%d\n' % (bc, r, c, 100 * r, 50 * c))
    afile = os.path.join(FILES, 'array%d.loupe.csv' % a)
    with open(afile, 'w') as fh:
        fh.write('Barcode,AARs\n')
        for i, (bc, r, c) in enumerate(spots):
            if i % 5 == 4:
                continue                                     # un-annotated spot
            fh.write('%s,%s\n' % (bc, names[(r + c + a) % (3 if a == 0 else 2)]))
    arrays.append((cfile, afile, pfile))

cfiles, afiles, pfiles = [list(t) for t in zip(*arrays)]
out = {}
with contextlib.redirect_stdout(io.StringIO()) as buf:
    ds = CountDataset(cfiles, afiles, pfiles, Visium=True)
out['spot_stdout'] = np.array(buf.getvalue())
out['spot_classes'] = np.array(list(ds.classes))
out['spot_len'] = len(ds)
xs, ys = zip(*[ds[i] for i in range(len(ds))])
out['spot_x'] = np.stack([x.numpy() for x in xs])
out['spot_y'] = np.array([int(y) for y in ys])
ds_sel = CountDataset(cfiles, afiles, pfiles, Visium=True, select_genes=['G3', 'G1'])
with contextlib.redirect_stdout(io.StringIO()):
    pass
out['spot_sel_x0'] = ds_sel[0][0].numpy()
gd = CountGridDataset(cfiles, afiles, pfiles, Visium=True, h_st=H, w_st=W)
out['grid_classes'] = np.array(list(gd.classes))
gx, gy = zip(*[gd[i] for i in range(len(gd))])
out['grid_x'] = np.stack([x.numpy() for x in gx])
out['grid_y'] = np.stack([y.numpy() for y in gy])
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'count_datasets.npz'), **out)
print('spots', out['spot_len'], 'classes', out['spot_classes'], 'grid', out['grid_x'].shape, out['grid_y'].shape,
      'fg', int((out['grid_y'] > 0).sum()))

# ---- classic-ST / Splotch mode (Visium=False): coordinates are Cartesian floats "x_y" rounded to the nearest grid position
# (utils.py:147-149), annotations a one-hot matrix (annotations x spot coordinate strings, utils.py:234-244).  As shipped, the
# reference's reader keeps only annotation ROWS whose sum is exactly one (utils.py:238) and labels every spot by the argmax over
# the rows that survive - all-zero columns therefore get class 0.  The fixture pins that behaviour as it is: rows 'A' and 'C'
# mark one spot each and survive, row 'B' marks three spots and is dropped.
st_out = {}
Hs, Ws = 6, 5
rng2 = np.random.RandomState(7)
coords = [(0.98, 1.03), (2.04, 0.97), (3.01, 3.96), (1.02, 4.99), (3.97, 2.02), (0.03, 0.01)]
cstrs = ['%.2f_%.2f' % c for c in coords]
st_counts = rng2.poisson(4.0, size=(G, len(coords)))
st_cfile = os.path.join(FILES, 'st0.counts.tsv')
with open(st_cfile, 'w') as fh:
    fh.write('\t'.join([''] + cstrs) + '\n')
    for gi, gname in enumerate(genes):
        fh.write('\t'.join([gname] + [str(v) for v in st_counts[gi]]) + '\n')
onehot = np.zeros((3, len(coords)), dtype=int)
onehot[0, 1] = 1                      # A: one spot
onehot[1, [0, 2, 4]] = 1              # B: three spots (row sum 3: dropped by the reader)
onehot[2, 3] = 1                      # C: one spot
st_afile = os.path.join(FILES, 'st0.annot.tsv')
with open(st_afile, 'w') as fh:
    fh.write('\t'.join([''] + cstrs) + '\n')
    for name, row in zip(['A', 'B', 'C'], onehot):
        fh.write('\t'.join([name] + [str(v) for v in row]) + '\n')
gd_st = CountGridDataset([st_cfile], [st_afile], Visium=False, h_st=Hs, w_st=Ws)
x, y = gd_st[0]
st_out['st_grid_x'], st_out['st_grid_y'] = x.numpy(), y.numpy()
try:                                   # without annotation files the reference's reader never creates its label grid
    CountGridDataset([st_cfile], None, Visium=False, h_st=Hs, w_st=Ws)[0]
    st_out['st_grid_noannot_error'] = np.array('')
except Exception as exc:               # (utils.py:164: UnboundLocalError) - recorded, not reproduced
    st_out['st_grid_noannot_error'] = np.array(type(exc).__name__)
gd_sel = CountGridDataset([st_cfile], [st_afile], Visium=False, h_st=Hs, w_st=Ws, select_genes=['G4', 'G0'])
st_out['st_grid_sel_x'] = gd_sel[0][0].numpy()
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'count_datasets_splotch.npz'), **st_out)
print('splotch grid', st_out['st_grid_x'].shape, 'labels', np.unique(st_out['st_grid_y']), 'filled', int((st_out['st_grid_x'].sum(0) > 0).sum()))
