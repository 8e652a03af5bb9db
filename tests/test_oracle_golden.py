"""CPU: pin the oracle (oracle/) against fixtures produced by the reference's own code
(tools/gen_golden.py).  No GPU, no reference import at run time."""
import copy
import io
import contextlib

import numpy as np
import pytest
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

from conftest import load_golden, sub
from oracle import densenet as odn
from oracle import gridnet as ogn
from oracle import loops as oloops
from oracle import masked_ce as oce
from oracle.mlp import count_mlp, sequential_forward

torch.set_num_threads(1)

TINY_LARGE = dict(growth_rate=4, block_config=(2, 2), num_init_features=8, bn_size=2, num_classes=5,
                  small_inputs=False)
TINY_SMALL = dict(growth_rate=6, block_config=(2, 3, 2), num_init_features=10, bn_size=2, num_classes=7,
                  small_inputs=True, classify=False, compression=0.5)


def close(a, b, rtol=1e-5, atol=1e-6):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    tol = atol + rtol * b.abs().max().item() if b.numel() else atol
    assert err <= tol, "max abs err %.3e > tol %.3e" % (err, tol)


@pytest.mark.parametrize("name,kw", [("densenet_tiny_large", TINY_LARGE), ("densenet_tiny_small", TINY_SMALL)])
def test_densenet_oracle_matches_reference(name, kw):
    g = load_golden(name)
    cfg = odn.DenseNetCfg(**{k: (tuple(v) if k == 'block_config' else v) for k, v in kw.items()})
    assert [k for k, _ in odn.state_layout(cfg)] == [k[3:] for k in g if k.startswith('sd/')]
    x = torch.from_numpy(g['x'])
    labels = torch.from_numpy(g['labels'])
    # eval forward
    sd = sub(g, 'sd')
    close(odn.forward(sd, x, cfg, training=False), g['eval_out'])
    # eval-mode gradients
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone())
          for k, v in sub(g, 'sd').items()}
    xg = x.clone().requires_grad_(True)
    loss = nn.functional.cross_entropy(odn.forward(sd, xg, cfg, training=False), labels)
    loss.backward()
    assert abs(loss.item() - float(g['eval_loss'])) < 1e-5
    close(xg.grad, g['evalgrad/x'], rtol=1e-4)
    for k, ref in sub(g, 'evalgrad').items():
        if k != 'x':
            close(sd[k].grad, ref, rtol=1e-4)
    # train-mode forward, gradients, running statistics
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone())
          for k, v in sub(g, 'sd').items()}
    y = odn.forward(sd, x, cfg, training=True)
    close(y, g['train_out'], rtol=1e-4)
    loss = nn.functional.cross_entropy(y, labels)
    loss.backward()
    assert abs(loss.item() - float(g['train_loss'])) < 1e-5
    for k, ref in sub(g, 'traingrad').items():
        close(sd[k].grad, ref, rtol=2e-4, atol=1e-6)
    for k, ref in sub(g, 'post').items():
        close(sd[k], ref, rtol=1e-5)


def test_densenet121_closed_form_outputs():
    g = load_golden('densenet121_closedform')
    cfg = odn.DenseNetCfg(num_classes=8, **odn.DENSENET121)
    sd = odn.closed_form_state(cfg)
    assert len(sd) == 727                                    # reference state_dict has 727 entries
    n_params = sum(v.numel() for k, v in sd.items() if 'running' not in k and 'num_batches' not in k)
    assert n_params == 6962056
    with torch.no_grad():
        close(odn.forward(sd, odn.closed_form_images(3, 64), cfg), g['eval_out_p64'], rtol=2e-4)
        close(odn.forward(sd, odn.closed_form_images(2, 128), cfg), g['eval_out_p128'], rtol=2e-4)
        sd2 = {k: v.clone() for k, v in sd.items()}
        close(odn.forward(sd2, odn.closed_form_images(4, 64), cfg, training=True), g['train_out_p64'], rtol=5e-4)


def test_count_mlp_oracle_matches_reference_stack():
    g = load_golden('mlp_count')
    f = count_mlp(200, 8)
    f.load_state_dict(sub(g, 'sd'))
    x = torch.from_numpy(g['x'])
    f.eval()
    close(f(x), g['eval_out'])
    close(sequential_forward(f, x, training=False)[0], g['eval_out'], rtol=1e-4)
    f.train()
    xg = x.clone().requires_grad_(True)
    y = f(xg)
    close(y, g['train_out'], rtol=1e-4)
    close(sequential_forward(f, x, training=True)[0], g['train_out'], rtol=1e-4)
    loss = nn.functional.cross_entropy(y, torch.from_numpy(g['labels']))
    loss.backward()
    assert abs(loss.item() - float(g['train_loss'])) < 1e-5
    close(xg.grad, g['traingrad/x'], rtol=1e-4)
    for k, p in f.named_parameters():
        close(p.grad, g['traingrad/' + k], rtol=1e-4)
    for k, ref in sub(g, 'post').items():
        close(f.state_dict()[k], ref)


def _loaders(x, y, n_train, batch):
    return {'train': DataLoader(TensorDataset(x[:n_train], y[:n_train]), batch_size=batch, shuffle=False),
            'val': DataLoader(TensorDataset(x[n_train:], y[n_train:]), batch_size=batch, shuffle=False)}


def test_spotwise_loop_mlp_history():
    g = load_golden('spotwise_mlp')
    f = count_mlp(64, 8)
    f.load_state_dict(sub(g, 'init'))
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    dl = _loaders(x, y, int(g['n_train']), int(g['batch']))
    opt = torch.optim.Adam(f.parameters(), lr=float(g['lr']))
    f, vh, th = oloops.train_spotwise(f, dl, nn.CrossEntropyLoss(), opt, num_epochs=int(g['epochs']))
    np.testing.assert_allclose(th, g['train_history'], rtol=2e-5)
    np.testing.assert_allclose(vh, g['val_history'], rtol=2e-5)
    for k, ref in sub(g, 'final').items():
        close(f.state_dict()[k], ref, rtol=1e-3, atol=1e-5)


def test_spotwise_loop_tiny_densenet_history():
    g = load_golden('spotwise_densenet_tiny')
    f = odn.DenseNet(**TINY_LARGE)
    f.load_named_state(sub(g, 'init'))
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    dl = _loaders(x, y, int(g['n_train']), int(g['batch']))
    opt = torch.optim.Adam(f.parameters(), lr=float(g['lr']))
    f, vh, th = oloops.train_spotwise(f, dl, nn.CrossEntropyLoss(), opt, num_epochs=int(g['epochs']))
    np.testing.assert_allclose(th, g['train_history'], rtol=1e-4)
    np.testing.assert_allclose(vh, g['val_history'], rtol=1e-4)
    final = f.named_state()
    for k, ref in sub(g, 'final').items():
        close(final[k], ref, rtol=2e-3, atol=2e-5)


def _load_grid_state(model, init, densenet_prefixes=()):
    """Load a reference state_dict into an oracle grid model (oracle DenseNet keeps flat names)."""
    own = model.state_dict()
    mapped = {}
    for k, v in init.items():
        kk = k
        for pre in densenet_prefixes:
            if k.startswith(pre + '.'):
                kk = pre + '.' + k[len(pre) + 1:].replace('.', '__')
        mapped[kk] = v
    missing = set(own) - set(mapped)
    assert not missing, missing
    model.load_state_dict(mapped)


def test_gridwise_cartesian_history():
    g = load_golden('gridwise_cartesian')
    G, H, W, C = 24, 7, 6, 5
    f = count_mlp(G, C)
    m = ogn.GridNet(f, (G,), (H, W), C, use_bn=True)
    _load_grid_state(m, sub(g, 'init'))
    for p in m.patch_classifier.parameters():
        p.requires_grad = False
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    m.eval()
    with torch.no_grad():
        close(m(x[:2]), g['fwd0'], rtol=1e-4)
    dl = _loaders(x, y, 4, 2)
    opt = torch.optim.Adam(m.corrector.parameters(), lr=1e-3)
    m, vh, th = oloops.train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=2)
    np.testing.assert_allclose(th, g['train_history'], rtol=1e-4)
    np.testing.assert_allclose(vh, g['val_history'], rtol=1e-4)


@pytest.mark.parametrize("name,accum,limit,fopt,ntrain", [
    ("gridwise_hexoddr", 3, None, False, 5),
    ("gridwise_hexoddr_fopt", 1, 7, True, 3)])
def test_gridwise_hexoddr_history(name, accum, limit, fopt, ntrain):
    g = load_golden(name)
    G, H, W, C = 24, 8, 6, 5
    f = count_mlp(G, C)
    m = ogn.GridNetHexOddr(f, (G,), (H, W), C, use_bn=True, atonce_patch_limit=limit)
    _load_grid_state(m, sub(g, 'init'))
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    if not fopt:
        for p in m.patch_classifier.parameters():
            p.requires_grad = False
        m.eval()
        with torch.no_grad():
            close(m.patch_predictions(x[:2]), g['pp0'], rtol=1e-4)
            close(m(x[:2]), g['fwd0'], rtol=1e-4)
    dl = _loaders(x, y, ntrain, 1)
    opt = torch.optim.Adam(m.corrector.parameters(), lr=1e-3)
    f_opt = torch.optim.Adam(m.patch_classifier.parameters(), lr=1e-4) if fopt else None
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m, vh, th = oloops.train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=2, f_opt=f_opt,
                                          accum_iters=accum)
    np.testing.assert_allclose(th, g['train_history'], rtol=1e-4)
    np.testing.assert_allclose(vh, g['val_history'], rtol=1e-4)
    for k, ref in sub(g, 'final').items():
        close(m.state_dict()[k], ref, rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize("name,fopt,accum", [("gridwise_hexmm_tutorial", False, 1), ("gridwise_hexmm_fopt", True, 2)])
def test_gridwise_hexmm_history_and_quirks(name, fopt, accum):
    g = load_golden(name)
    G, H, W, P, C = 20, 6, 4, 32, 5
    f_img = odn.DenseNet(**TINY_LARGE)
    f_cnt = count_mlp(G, C)
    m = ogn.GridNetHexMM(f_img, f_cnt, (3, P, P), (G,), (H, W), C)
    _load_grid_state(m, sub(g, 'init'), densenet_prefixes=('patch_classifier', 'image_classifier'))
    xi, xc, y = torch.from_numpy(g['x_img']), torch.from_numpy(g['x_cnt']), torch.from_numpy(g['y'])
    m.eval()
    with torch.no_grad():
        close(m.patch_predictions([xi[:2], xc[:2]]), g['pp0'], rtol=1e-4)
        close(m([xi[:2], xc[:2]]), g['fwd0'], rtol=1e-4)
    data = [((xi[i], xc[i]), y[i]) for i in range(4)]
    dl = {'train': DataLoader(data[:3], batch_size=1, shuffle=False),
          'val': DataLoader(data[3:], batch_size=1, shuffle=False)}
    opt = torch.optim.Adam(m.corrector.parameters(), lr=1e-3)
    f_opt = None
    if fopt:
        f_opt = torch.optim.Adam(list(m.image_classifier.parameters()) + list(m.count_classifier.parameters()),
                                 lr=1e-4)
    else:
        for p in m.patch_classifier.parameters():
            p.requires_grad = False
    m, vh, th = oloops.train_gridwise(m, dl, nn.CrossEntropyLoss(), opt, num_epochs=2, f_opt=f_opt,
                                      accum_iters=accum)
    np.testing.assert_allclose(th, g['train_history'], rtol=2e-4)
    np.testing.assert_allclose(vh, g['val_history'], rtol=2e-4)
    # SURVEY a-8 quirks, as recorded from the reference run
    assert int(g['patch_classifier_is_image']) == 1 and m.patch_classifier is m.image_classifier
    assert int(g['count_training_flag']) == int(m.count_classifier.training)
    ref_final = sub(g, 'final')
    got = m.state_dict()
    for k in ('count_classifier.2.running_mean', 'count_classifier.2.running_var', 'corrector.0.kernel0',
              'corrector.8.bias_tensor', 'corrector.2.running_var'):
        close(got[k], ref_final[k], rtol=2e-3, atol=2e-5)


def test_masked_ce_on_reference_saved_maps():
    g = load_golden('masked_ce_maynard')
    z = torch.from_numpy(g['logits']).unsqueeze(0)
    lab = torch.from_numpy(g['labels']).unsqueeze(0)
    for accum in (1, 4):
        zz = z.clone().requires_grad_(True)
        loss, preds, t = oce.masked_ce(zz, lab, accum)
        assert abs(loss.item() - float(g['loss_accum%d' % accum])) < 2e-6
        loss.backward()
        close(zz.grad[0], g['grad_accum%d' % accum], rtol=1e-5, atol=1e-9)
        close(oce.masked_ce_grad(z, lab, accum)[0], g['grad_accum%d' % accum], rtol=1e-5, atol=1e-9)
    # the reference's saved softmax map of the same logits pins the softmax of all_fgd_predictions
    fg = torch.from_numpy(g['labels']) > 0
    sm = torch.softmax(z[0], dim=0)
    close(sm[:, fg], torch.from_numpy(g['smax_saved'])[:, fg], rtol=1e-5, atol=1e-7)
    t, p, s = oce.fgd_softmax_argmax(z, lab)
    assert t.numel() == int(fg.sum()) and s.shape == (t.numel(), 7)


@pytest.mark.parametrize("tag", ["gridwise_hexoddr_fopt", "gridwise_hexmm_tutorial"])
def test_oracle_running_statistics_after_one_epoch(tag):
    """The oracle's loop and models against the reference's one-epoch BatchNorm buffers (tools/gen_golden_epoch0.py): the
    statistics the validation phase runs on, pinned directly instead of through a loss tolerance."""
    from oracle import densenet as odn, gridnet as ogn, loops
    from oracle.mlp import count_mlp
    g, e0 = load_golden(tag), load_golden(tag + '_epoch0')
    C = 5
    torch.set_num_threads(1)
    if tag == 'gridwise_hexoddr_fopt':
        G, H, W = 24, 8, 6
        m = ogn.GridNetHexOddr(count_mlp(G, C), (G,), (H, W), C, use_bn=True, atonce_patch_limit=7)
        m.load_state_dict(sub(g, 'init'))
        x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
        data = [(x[i], y[i]) for i in range(5)]
        kw = dict(f_opt=torch.optim.Adam(m.patch_classifier.parameters(), lr=1e-4))
    else:
        G, H, W, P = 20, 6, 4, 32
        f_img = odn.DenseNet(growth_rate=4, block_config=(2, 2), num_init_features=8, bn_size=2, num_classes=C, small_inputs=False)
        m = ogn.GridNetHexMM(f_img, count_mlp(G, C), (3, P, P), (G,), (H, W), C)
        _load_grid_state(m, sub(g, 'init'), densenet_prefixes=('patch_classifier', 'image_classifier'))
        xi, xc, y = torch.from_numpy(g['x_img']), torch.from_numpy(g['x_cnt']), torch.from_numpy(g['y'])
        data = [((xi[i], xc[i]), y[i]) for i in range(4)]
        for p in m.patch_classifier.parameters():
            p.requires_grad = False
        kw = {}
    from torch.utils.data import DataLoader
    dl = {'train': DataLoader(data[:3], batch_size=1, shuffle=False), 'val': DataLoader(data[3:], batch_size=1, shuffle=False)}
    opt = torch.optim.Adam(m.corrector.parameters(), lr=1e-3)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        m, vh, th = loops.train_gridwise(m, dl, torch.nn.CrossEntropyLoss(), opt, num_epochs=1, **kw)
    assert abs(th[0] - float(e0['train_loss'])) <= 1e-6 and abs(vh[0] - float(e0['val_loss'])) <= 1e-5
    got = m.state_dict()
    for k, ref in sub(e0, 'buf').items():
        for pre in ('patch_classifier', 'image_classifier'):      # the oracle DenseNet keeps flat buffer names
            if k.startswith(pre + '.features'):
                k = pre + '.' + k[len(pre) + 1:].replace('.', '__')
        np.testing.assert_allclose(got[k].numpy(), ref, rtol=1e-5, atol=1e-7, err_msg=k)
