"""CPU: host-side logic of the product package (no kernels run here).

 * the C-ABI library loads and exports every symbol include/gridnext_hip.h declares, with ctypes signatures
   for each one;
 * the product path fails loudly without a HIP device (no CPU fallback);
 * gridnext_amd.training's loops - generic path, driven with the CPU oracle models - reproduce the histories
   the reference's own loops produced (tests/golden);
 * module surfaces: state_dict key order of the drop-in classes equals the reference's.
"""
import contextlib
import io
import os
import re

import numpy as np
import pytest
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

from conftest import ROOT, load_golden, sub

torch.set_num_threads(1)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        r = fn(*a, **k)
    return r, buf.getvalue()


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'gridnext_hip.h')).read()
    return sorted(set(re.findall(r'\b(gnx_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from gridnext_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = _lib.lib()
    declared = header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(handle, name), "libgridnext_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "no ctypes signature for %s" % name
    assert sorted(_lib.SIGNATURES) == declared, "header and ctypes table disagree"


def test_product_path_fails_loudly_without_hip():
    if torch.cuda.is_available():
        pytest.skip("needs a CPU-only box")
    import gridnext_amd as ga
    from gridnext_amd import functional as GF
    m = ga.DenseNet(growth_rate=4, block_config=(2,), num_init_features=8, bn_size=2, num_classes=3, small_inputs=True)
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.rand(1, 3, 8, 8))
    with pytest.raises(RuntimeError, match="HIP device"):
        GF.masked_cross_entropy(torch.rand(4, 3), torch.ones(4, dtype=torch.long), 1)
    with pytest.raises(RuntimeError, match="HIP device"):
        GF.hexconv(torch.rand(1, 4, 4, 2), torch.rand(3, 2, 3, 1), torch.rand(3, 2, 2, 2), None, True)


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, 'gridnext_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, re.M), f


def test_state_dict_keys_match_reference_order():
    import gridnext_amd as ga
    from gridnext_amd.synthetic import count_mlp
    g = load_golden('densenet_tiny_large')
    m = ga.DenseNet(growth_rate=4, block_config=(2, 2), num_init_features=8, bn_size=2, num_classes=5,
                    small_inputs=False)
    assert list(m.state_dict()) == [k[3:] for k in g if k.startswith('sd/')]
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(g['sd/' + k].shape), k
    g = load_golden('gridwise_hexmm_tutorial')
    mm = ga.GridNetHexMM(m, count_mlp(20, 5), (3, 32, 32), (20,), (6, 4), 5)
    assert list(mm.state_dict()) == [k[5:] for k in g if k.startswith('init/')]
    g = load_golden('gridwise_cartesian')
    gn = ga.GridNet(count_mlp(24, 5), (24,), (7, 6), 5)
    assert list(gn.state_dict()) == [k[5:] for k in g if k.startswith('init/')]
    # reference quirk: the MM model leaves patch_classifier on the image network (gridnet_models.py:229-233)
    assert mm.patch_classifier is mm.image_classifier


def test_densenet_init_statistics_follow_reference():
    import gridnext_amd as ga
    torch.manual_seed(0)
    m = ga.DenseNet(growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64, bn_size=4, num_classes=8,
                    small_inputs=False)
    assert sum(p.numel() for p in m.parameters()) == 6962056
    w = m.features.denseblock1.denselayer1.conv2.weight               # N(0, sqrt(2/(out*kh*kw))), densenet.py:142-144
    assert abs(w.std().item() - (2.0 / (32 * 9)) ** 0.5) < 0.01
    assert float(m.features.norm0.weight.detach().min()) == 1.0
    assert float(m.classifier.bias.detach().abs().max()) == 0.0


def _loaders(x, y, n_train, batch):
    return {'train': DataLoader(TensorDataset(x[:n_train], y[:n_train]), batch_size=batch, shuffle=False),
            'val': DataLoader(TensorDataset(x[n_train:], y[n_train:]), batch_size=batch, shuffle=False)}


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU walk of the generic loop path")
def test_product_spot_loop_generic_path_reproduces_reference_history(tmp_path):
    from gridnext_amd.training import train_spotwise
    from oracle import densenet as odn
    g = load_golden('spotwise_densenet_tiny')
    f = odn.DenseNet(growth_rate=4, block_config=(2, 2), num_init_features=8, bn_size=2, num_classes=5,
                     small_inputs=False)
    f.load_named_state(sub(g, 'init'))
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    dl = _loaders(x, y, int(g['n_train']), int(g['batch']))
    opt = torch.optim.Adam(f.parameters(), lr=float(g['lr']))
    out = str(tmp_path / 'f.pth')
    (f, vh, th), text = quiet(train_spotwise, f, dl, nn.CrossEntropyLoss(), opt, num_epochs=int(g['epochs']),
                              outfile=out)
    np.testing.assert_allclose(th, g['train_history'], rtol=1e-4)
    np.testing.assert_allclose(vh, g['val_history'], rtol=1e-4)
    assert os.path.exists(out)
    # printed lines have the reference's format (training.py:76)
    ref_lines = [l for l in str(g['stdout']).splitlines() if 'Loss:' in l]
    got_lines = [l for l in text.splitlines() if 'Loss:' in l]
    assert [l.split()[0] for l in ref_lines] == [l.split()[0] for l in got_lines]
    assert all(re.fullmatch(r'(train|val) Loss: \d+\.\d{4} Acc: \d+\.\d{4}', l) for l in got_lines)
    assert text.splitlines()[0] == 'Epoch 0/%d' % (int(g['epochs']) - 1)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU walk of the generic loop path")
@pytest.mark.parametrize("name,accum,limit,fopt,ntrain", [("gridwise_hexoddr", 3, None, False, 5),
                                                          ("gridwise_hexoddr_fopt", 1, 7, True, 3)])
def test_product_grid_loop_generic_path_reproduces_reference_history(tmp_path, name, accum, limit, fopt, ntrain):
    from gridnext_amd.training import train_gridwise
    from oracle import gridnet as ogn
    from oracle.mlp import count_mlp
    g = load_golden(name)
    G, H, W, C = 24, 8, 6, 5
    m = ogn.GridNetHexOddr(count_mlp(G, C), (G,), (H, W), C, use_bn=True, atonce_patch_limit=limit)
    m.load_state_dict(sub(g, 'init'))
    if not fopt:
        for p in m.patch_classifier.parameters():
            p.requires_grad = False
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    dl = _loaders(x, y, ntrain, 1)
    opt = torch.optim.Adam(m.corrector.parameters(), lr=1e-3)
    f_opt = torch.optim.Adam(m.patch_classifier.parameters(), lr=1e-4) if fopt else None
    out = str(tmp_path / 'g.pth')
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        (m, vh, th), text = quiet(train_gridwise, m, dl, nn.CrossEntropyLoss(), opt, num_epochs=2, outfile=out,
                                  f_opt=f_opt, accum_iters=accum)
    np.testing.assert_allclose(th, g['train_history'], rtol=1e-4)
    np.testing.assert_allclose(vh, g['val_history'], rtol=1e-4)
    assert os.path.exists(out) and os.path.exists(str(tmp_path / 'g.opt'))       # training.py:187-195
    saved = torch.load(str(tmp_path / 'g.opt'))
    assert (set(saved) == {'g_opt', 'f_opt'}) if fopt else ('state' in saved and 'param_groups' in saved)
    assert str(g['stdout']).count('Loss:') == text.count('Loss:')


def test_mmstack_dataset_contract():
    from gridnext_amd import MMStackDataset
    xi = torch.rand(3, 4, 4, 3, 8, 8)
    xc = torch.rand(3, 10, 4, 4)
    y1 = torch.randint(0, 4, (3, 4, 4))
    y2 = y1.clone()
    y2[0, 0, 0] = (y2[0, 0, 0] + 1) % 4
    ds = MMStackDataset(TensorDataset(xi, y1), TensorDataset(xc, y2))
    (a, b), y = ds[0]
    assert a.shape == (4, 4, 3, 8, 8) and b.shape == (10, 4, 4)
    assert y[0, 0] == 0 and torch.equal(y.flatten()[1:], y1[0].flatten()[1:])       # disagreeing labels are zeroed
    inputs, labels = next(iter(DataLoader(ds, batch_size=2)))
    assert isinstance(inputs, list) and len(inputs) == 2                              # what the loops test for
    with pytest.raises(AssertionError):
        MMStackDataset(TensorDataset(xi, y1), TensorDataset(xc[:2], y2[:2]))


def test_visium_coordinate_maps_match_reference_formulas():
    # reference utils.py:64-79: even rows x = col/2 ; odd rows x = (col-1)/2 ; inverse 2*col (+1 on odd rows)
    from gridnext_amd.utils import oddr_to_pseudo_hex, pseudo_hex_to_oddr, pseudo_to_true_hex
    for row in range(6):
        for col in range(5):
            pc, pr = oddr_to_pseudo_hex(col, row)
            assert (pc, pr) == ((2 * col if row % 2 == 0 else 2 * col + 1), row)
            assert pseudo_hex_to_oddr(pc, pr) == (col, row)
    x, y = pseudo_to_true_hex(3, 2)
    assert x == 1.5 and abs(y - 3 ** 0.5) < 1e-12


def test_count_datasets_match_reference_on_visium_files():
    """File-backed CountDataset / CountGridDataset (reference count_datasets.py:77-303) on the small Visium-style
    files under tests/golden/files/; expected tensors were produced by the reference's own classes
    (tools/gen_golden_datasets.py)."""
    from gridnext_amd.count_datasets import CountDataset, CountGridDataset
    g = load_golden('count_datasets')
    files = os.path.join(ROOT, 'tests', 'golden', 'files')
    cf = [os.path.join(files, 'array%d.counts.tsv' % a) for a in range(2)]
    af = [os.path.join(files, 'array%d.loupe.csv' % a) for a in range(2)]
    pf = [os.path.join(files, 'array%d.tissue_positions.csv' % a) for a in range(2)]
    (ds, text) = quiet(CountDataset, cf, af, pf, Visium=True)
    assert text == str(g['spot_stdout'])                                   # "N un-annotated spots"
    assert list(ds.classes) == list(g['spot_classes']) and len(ds) == int(g['spot_len'])
    for i in range(len(ds)):
        x, y = ds[i]
        assert x.dtype == torch.float32 and y.dtype == torch.int64 and y.dim() == 0
        assert np.array_equal(x.numpy(), g['spot_x'][i]) and int(y) == int(g['spot_y'][i])
    sel, _ = quiet(CountDataset, cf, af, pf, Visium=True, select_genes=['G3', 'G1'])
    assert np.array_equal(sel[0][0].numpy(), g['spot_sel_x0'])
    gd = CountGridDataset(cf, af, pf, Visium=True, h_st=8, w_st=6)
    assert list(gd.classes) == list(g['grid_classes']) and len(gd) == 2
    for i in range(2):
        x, y = gd[i]
        assert x.shape == (5, 8, 6) and x.dtype == torch.float32 and y.dtype == torch.int64
        assert np.array_equal(x.numpy(), g['grid_x'][i]) and np.array_equal(y.numpy(), g['grid_y'][i])
    with pytest.raises(ValueError, match='Length of count_files and annot_files must match'):
        CountGridDataset(cf, af[:1], pf)
    with pytest.raises(ValueError, match='Must provide Spaceranger position files'):
        CountDataset(cf, af, None)


def test_patch_datasets_contract(tmp_path):
    """PatchDataset / PatchGridDataset item contract (reference image_datasets.py:113-122, :192-232): the reference
    classes need torchvision (absent here), so this checks the documented contract on generated PNG patches that are
    named after the spots of the Visium fixture files."""
    from PIL import Image
    from gridnext_amd.image_datasets import PatchDataset, PatchGridDataset
    g = load_golden('count_datasets')
    files = os.path.join(ROOT, 'tests', 'golden', 'files')
    af = [os.path.join(files, 'array%d.loupe.csv' % a) for a in range(2)]
    pf = [os.path.join(files, 'array%d.tissue_positions.csv' % a) for a in range(2)]
    dirs = []
    rng = np.random.RandomState(3)
    pixels = {}
    for a in range(2):
        d = tmp_path / ('array%d' % a)
        d.mkdir()
        dirs.append(str(d))
        header = open(os.path.join(files, 'array%d.counts.tsv' % a)).readline().strip('\n').split('\t')[1:]
        for cstr in header:
            img = rng.randint(0, 256, size=(8, 8, 3), dtype=np.uint8)
            pixels[(a, cstr)] = img
            Image.fromarray(img).save(str(d / ('spot_%s.png' % cstr)))
    ds = PatchDataset(dirs, af, pf, Visium=True, img_ext='png')
    assert list(ds.classes) == list(g['spot_classes']) and len(ds) == int(g['spot_len'])
    x, y = ds[0]
    assert x.shape == (3, 8, 8) and x.dtype == torch.float32 and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    assert y.dtype == torch.int64
    gd = PatchGridDataset(dirs, af, pf, Visium=True, img_ext='png', h_st=8, w_st=6)
    for a in range(2):
        grid, labels = gd[a]
        assert grid.shape == (8, 6, 3, 8, 8) and labels.shape == (8, 6)
        assert torch.equal(labels, torch.from_numpy(g['grid_y'][a]))            # same labels as the count grid
        has_img = (grid.abs().sum(dim=(2, 3, 4)) > 0)
        assert int(has_img.sum()) == sum(1 for k in pixels if k[0] == a)
        assert bool((has_img | (labels == 0)).all())                            # background = no label
        cstr = next(k[1] for k in pixels if k[0] == a)
        cx, cy = map(int, cstr.split('_'))
        from gridnext_amd.utils import pseudo_hex_to_oddr
        ox, oy = pseudo_hex_to_oddr(cx, cy)
        want = torch.from_numpy(pixels[(a, cstr)]).permute(2, 0, 1).float() / 255
        assert torch.equal(grid[oy, ox], want)


def test_densenet_derived_weight_cache_is_invalidated_by_every_kind_of_write():
    """the eval forward's derived tensors (folded BN, repacked conv weights) are keyed on (_version, data_ptr)
    of their sources plus a model epoch; writes that bump no version counter must go through invalidate_cache()."""
    import gridnext_amd as ga
    m = ga.DenseNet(growth_rate=4, block_config=(2, 2), num_init_features=8, bn_size=2, num_classes=5, small_inputs=False)
    bn = m.features.norm0
    srcs = [bn.weight, bn.bias, bn.running_mean, bn.running_var]
    k0 = m._key(srcs)
    with torch.no_grad():
        bn.weight.mul_(2.0)                                   # in-place op: version counter
    k1 = m._key(srcs)
    assert k1 != k0
    bn.running_mean.data.fill_(3.0)                           # .data write: no version bump ...
    assert m._key(srcs) == k1
    m._cache['fold'] = ('stale',)
    m.invalidate_cache()                                      # ... which is what invalidate_cache() is for
    assert m._cache == {} and m._key(srcs) != k1
    k2 = m._key(srcs)
    m._cache['fold'] = ('stale',)
    m.load_state_dict(m.state_dict())                         # load_state_dict hook
    assert m._cache == {} and m._key(srcs) != k2
    m._cache['fold'] = ('stale',)
    m.double()                                                # _apply (.to / .float / .double / .cuda)
    assert m._cache == {}


def test_grid_forward_does_not_change_the_users_densenet_chunk_setting():
    """GridNet used to leave `f.atonce = atonce_patch_limit` behind on the user's DenseNet."""
    import inspect
    import gridnext_amd.gridnet_models as gm
    src = inspect.getsource(gm.GridNet._f_rows)
    assert 'finally' in src and 'f.atonce = keep' in src


def test_bench_launcher_parent_stays_off_the_gpu_and_reports_failed_ranks():
    """`python bench.py --gpus N` starts N fresh worker processes; the parent imports neither torch nor the package, and a
    failing rank (here: every rank, there is no HIP device in this container) makes it exit non-zero instead of hanging."""
    import subprocess
    import sys
    code = ("import sys; sys.argv=['bench.py']; import bench; "
            "assert 'torch' not in sys.modules and 'gridnext_amd' not in sys.modules; "
            "a = bench.parse(['--gpus', '4', '--steps', '3']); assert a.gpus == 4 and not a.worker; print('ok')")
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                          '--no-cpu-baseline'], cwd=ROOT, capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode != 0
    assert 'exited with code' in out.stderr


def _splotch_tree(tmp_path, rng, H, W, G, P, ext='jpg'):
    """One synthetic array in the formats MultiModal*Dataset read: count TSV (genes x 'x_y' spots), Splotch one-hot
    annotation TSV (annotations x spots), one '<x>_<y>.jpg' per spot that has image data."""
    from PIL import Image
    spots = [(2 * x + (row % 2), row) for row in range(H) for x in range(W) if rng.rand() < 0.8]      # pseudo-hex (col, row)
    names = ['%d_%d' % s for s in spots]
    counts = rng.poisson(3.0, size=(G, len(spots)))
    cfile = str(tmp_path / 'counts.tsv')
    with open(cfile, 'w') as fh:
        fh.write('\t'.join([''] + names) + '\n')
        for gi in range(G):
            fh.write('\t'.join(['G%d' % gi] + [str(v) for v in counts[gi]]) + '\n')
    labels = {}
    afile = str(tmp_path / 'annot.tsv')
    annotated = [n for i, n in enumerate(names) if i % 4 != 3]
    with open(afile, 'w') as fh:
        fh.write('\t'.join([''] + annotated) + '\n')
        for k in range(3):
            row = []
            for n in annotated:
                labels.setdefault(n, int(rng.randint(0, 3)))
                row.append('1' if labels[n] == k else '0')
            fh.write('\t'.join(['AAR%d' % k] + row) + '\n')
    imdir = tmp_path / 'patches'
    imdir.mkdir()
    pixels = {}
    for i, n in enumerate(names):
        if i % 5 == 0:
            continue                                              # spot without image data
        img = rng.randint(1, 256, size=(P, P, 3), dtype=np.uint8)
        Image.fromarray(img).save(str(imdir / ('%s.%s' % (n, ext))), quality=95)
        pixels[n] = np.array(Image.open(str(imdir / ('%s.%s' % (n, ext)))))      # what a JPEG decoder returns
    return cfile, afile, str(imdir), names, counts, labels, pixels


def test_multimodal_file_datasets_contract(tmp_path):
    """MultiModalDataset / MultiModalGridDataset (reference multimodal_datasets.py:141-246, DEFUNCT there): constructor
    signatures, skip rules and item tuples on a hand-built tree, JPEG patches through the ToTensor contract
    (uint8 HWC -> float32 CHW / 255)."""
    import inspect
    from gridnext_amd import MultiModalDataset, MultiModalGridDataset
    from gridnext_amd.utils import pseudo_hex_to_oddr
    assert list(inspect.signature(MultiModalDataset.__init__).parameters)[1:8] == \
        ['count_files', 'img_files', 'annot_files', 'select_genes', 'img_transforms', 'cfile_delim', 'afile_delim']
    assert list(inspect.signature(MultiModalGridDataset.__init__).parameters)[1:11] == \
        ['count_files', 'img_files', 'annot_files', 'select_genes', 'h_st', 'w_st', 'Visium', 'img_transforms',
         'cfile_delim', 'afile_delim']
    rng = np.random.RandomState(11)
    H, W, G, P = 6, 5, 4, 8
    cfile, afile, imdir, names, counts, labels, pixels = _splotch_tree(tmp_path, rng, H, W, G, P)
    with pytest.raises(ValueError):
        MultiModalDataset([cfile], [imdir, imdir], [afile])
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        ds = MultiModalDataset([cfile], [imdir], [afile])
    usable = [n for n in names if n in labels and n in pixels]
    assert len(ds) == len(usable) and 'no image data' in buf.getvalue()
    for i in (0, len(ds) - 1):
        cvec, img, y = ds[i]
        n = usable[i]
        assert torch.equal(cvec, torch.from_numpy(counts[:, names.index(n)].astype(np.float32)))
        assert img.dtype == torch.float32 and img.shape == (3, P, P)
        assert torch.equal(img, torch.from_numpy(pixels[n]).permute(2, 0, 1).float() / 255)          # ToTensor on a JPEG
        assert int(y) == labels[n] and y.dtype == torch.int64
    gd = MultiModalGridDataset([cfile], [imdir], [afile], h_st=H, w_st=W)
    cg, pg, ag = gd[0]
    assert cg.shape == (G, H, W) and pg.shape == (H, W, 3, P, P) and ag.shape == (H, W) and ag.dtype == torch.int64
    want_fg = 0
    for n in names:
        x, y = pseudo_hex_to_oddr(*map(int, n.split('_')))
        if n in labels and n in pixels:
            want_fg += 1
            assert int(ag[y, x]) == labels[n] + 1
            assert torch.equal(cg[:, y, x], torch.from_numpy(counts[:, names.index(n)].astype(np.float32)))
            assert torch.equal(pg[y, x], torch.from_numpy(pixels[n]).permute(2, 0, 1).float() / 255)
        else:                                                     # no image or no annotation: background everywhere
            assert int(ag[y, x]) == 0 and float(pg[y, x].abs().max()) == 0.0
            if n not in pixels:
                assert float(cg[:, y, x].abs().max()) == 0.0
    assert int((ag > 0).sum()) == want_fg
    (xi, xc), y2 = MultiModalGridDataset([cfile], [imdir], [afile], h_st=H, w_st=W, training_pairs=True)[0]
    assert torch.equal(xi, pg) and torch.equal(xc, cg) and torch.equal(y2, ag)
    batch = next(iter(DataLoader(MultiModalGridDataset([cfile], [imdir], [afile], h_st=H, w_st=W, training_pairs=True),
                                 batch_size=1)))
    assert isinstance(batch[0], list) and batch[0][0].shape == (1, H, W, 3, P, P)       # what train_gridwise tests for


def test_patch_dataset_reads_jpeg_through_the_totensor_contract(tmp_path):
    """image_datasets.py:102-105: the default transform is ToTensor - JPEG file -> decoder's uint8 HWC -> float CHW / 255."""
    from PIL import Image
    from gridnext_amd.image_datasets import PatchDataset
    rng = np.random.RandomState(5)
    d = tmp_path / 'arr'
    d.mkdir()
    img = rng.randint(0, 256, size=(16, 16, 3), dtype=np.uint8)
    Image.fromarray(img).save(str(d / 'spot_3_4.jpg'), quality=90)
    decoded = np.array(Image.open(str(d / 'spot_3_4.jpg')))
    ds = PatchDataset([str(d)], annot_files=None, Visium=False)
    x, _ = ds[0]
    assert x.dtype == torch.float32 and torch.equal(x, torch.from_numpy(decoded).permute(2, 0, 1).float() / 255)


def test_prefetcher_is_transparent_on_the_cpu_and_keeps_batch_structure():
    """prefetch.DevicePrefetcher: on a CPU 'device' it is the loader itself; helper functions keep (list, tensor) nesting."""
    from gridnext_amd import prefetch
    data = [((torch.rand(2, 3), torch.rand(4)), torch.tensor(i)) for i in range(5)]
    dl = DataLoader(data, batch_size=2)
    assert prefetch.wrap(dl, torch.device('cpu')) is dl
    got = list(prefetch.DevicePrefetcher(dl, 'cpu'))
    assert len(got) == 3 and isinstance(got[0][0], list) and got[0][0][0].shape == (2, 2, 3)
    flat = prefetch._tensors(got[0], [])
    assert len(flat) == 3
    doubled = prefetch._map(got[0], lambda t: t * 2)
    assert isinstance(doubled[0], list) and torch.equal(doubled[1], got[0][1] * 2)


def test_prefetcher_draws_an_epoch_in_the_loaders_own_order():
    """prefetch._epoch_index_batches (the own-collate path of the pinned prefetcher) == the batches a plain DataLoader
    iterator yields for the same generator state, epoch after epoch, and it leaves the generator where the loader would."""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    from gridnext_amd import prefetch
    data = TensorDataset(torch.arange(23))
    for shuffle in (True, False):
        g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
        plain = DataLoader(data, batch_size=4, shuffle=shuffle, generator=g1)
        mine = DataLoader(data, batch_size=4, shuffle=shuffle, generator=g2)
        for _ in range(3):
            want = [b[0].tolist() for b in plain]
            got = prefetch._epoch_index_batches(mine)
            assert got == want
        assert torch.equal(g1.get_state(), g2.get_state())


def test_prefetcher_leaves_iterable_dataset_loaders_alone():
    """A DataLoader over an IterableDataset has a batch_sampler too (over torch's endless _InfiniteConstantSampler): the
    own-collate path must refuse it - drawing 'an epoch' of index batches from it would never return."""
    from torch.utils.data import DataLoader, IterableDataset, TensorDataset
    from gridnext_amd import prefetch

    class Stream(IterableDataset):
        def __iter__(self):
            for k in range(5):
                yield torch.full((3,), float(k)), torch.tensor(k)

    it_loader = DataLoader(Stream(), batch_size=2)
    assert it_loader.batch_sampler is not None                   # the trap
    assert not prefetch._default_collate_loader(it_loader)
    map_loader = DataLoader(TensorDataset(torch.zeros(4, 3), torch.zeros(4)), batch_size=2)
    assert prefetch._default_collate_loader(map_loader)
    # residency is read off the tensors a dataset HOLDS: no sample is drawn (no decode, no transform, no RNG draw)
    class Exploding(TensorDataset):
        def __getitem__(self, i):
            raise AssertionError("the residency probe must not index the dataset")
    pf = prefetch.DevicePrefetcher(DataLoader(Exploding(torch.zeros(4, 3), torch.zeros(4)), batch_size=2), 'cpu')
    assert pf._resident() is False
    pf.close()


def test_count_grid_dataset_matches_reference_in_splotch_mode():
    """Classic-ST / Splotch mode (`Visium=False`, reference count_datasets.py:281-303 -> utils.py:88-166, :234-244): Cartesian
    float coordinates rounded to grid positions, one-hot annotation matrix.  Expected tensors were produced by the reference's
    own CountGridDataset (tools/gen_golden_datasets.py) - including its reader's row-sum filter (utils.py:238: only annotation
    rows marking exactly one spot survive, every other spot gets the first surviving class), pinned as it is.  Without
    annotation files the reference raises (its label grid is never created, utils.py:164); here that case returns the counts
    with an all-background label grid - recorded as a deliberate difference."""
    from gridnext_amd.count_datasets import CountGridDataset
    g = load_golden('count_datasets_splotch')
    files = os.path.join(ROOT, 'tests', 'golden', 'files')
    cf, af = [os.path.join(files, 'st0.counts.tsv')], [os.path.join(files, 'st0.annot.tsv')]
    x, y = CountGridDataset(cf, af, Visium=False, h_st=6, w_st=5)[0]
    assert x.dtype == torch.float32 and y.dtype == torch.int64 and x.shape == (5, 6, 5) and y.shape == (6, 5)
    assert np.array_equal(x.numpy(), g['st_grid_x']) and np.array_equal(y.numpy(), g['st_grid_y'])
    assert int((y > 0).sum()) == 6 and sorted(np.unique(y.numpy())) == [0, 1, 2]
    xs, _ = CountGridDataset(cf, af, Visium=False, h_st=6, w_st=5, select_genes=['G4', 'G0'])[0]
    assert np.array_equal(xs.numpy(), g['st_grid_sel_x'])
    assert str(g['st_grid_noannot_error']) == 'UnboundLocalError'          # what the reference does without annotations
    x0, y0 = CountGridDataset(cf, None, Visium=False, h_st=6, w_st=5)[0]
    assert np.array_equal(x0.numpy(), g['st_grid_x']) and int(y0.abs().sum()) == 0


def test_best_weights_keeper_snapshots_like_deepcopy():
    """training.py:87-89 / :197-199 of the reference keep `copy.deepcopy(model.state_dict())` of the best epoch; the loops here
    refresh ONE set of snapshot buffers with multi-tensor copies - same contents, independent of the live parameters."""
    import copy
    import torch
    import torch.nn as nn
    from gridnext_amd.training import _BestKeeper
    torch.manual_seed(0)
    m = nn.Sequential(nn.Linear(6, 5), nn.BatchNorm1d(5), nn.ReLU(), nn.Linear(5, 3))
    k = _BestKeeper(m)
    first = copy.deepcopy(m.state_dict())
    m.train()
    m(torch.randn(8, 6))                                       # running statistics and num_batches_tracked move
    with torch.no_grad():
        for p in m.parameters():
            p.add_(1.0)
    for key, v in first.items():
        assert torch.equal(k.best_wts[key], v), key             # the snapshot did not follow the live tensors
    assert k.offer(0.5) and not k.offer(0.7)
    second = copy.deepcopy(m.state_dict())
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(2.0)
    assert list(k.best_wts.keys()) == list(second.keys())
    for key, v in second.items():
        assert torch.equal(k.best_wts[key], v) and k.best_wts[key].dtype == v.dtype, key
    m.load_state_dict(k.best_wts)
    for key, v in second.items():
        assert torch.equal(m.state_dict()[key], v), key
