#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own code.

Runs only in the build container (needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

The reference's modules are imported from /root/reference and executed on CPU;
only INPUTS and OUTPUTS (arrays) are written to tests/golden/*.npz - no reference
source or bytecode is copied.  `gridnext.gridnet_models` imports the third-party
`hexagdly`, which is not installed; for the fixtures that need those classes a
module object exposing `Conv2d = oracle.hexconv.HexConv2d` is registered under
that name first, so the reference's wrapper logic (reshape/permute order, concat
order, train/eval quirks, masking, stepping schedule) runs unmodified while the hex
arithmetic comes from the oracle (hex arithmetic stays "parity unpinned").
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from oracle import densenet as odn            # noqa: E402
from oracle import hexconv as ohex            # noqa: E402
from oracle.mlp import count_mlp              # noqa: E402

standin = types.ModuleType('hexagdly')
standin.Conv2d = ohex.HexConv2d
sys.modules['hexagdly'] = standin

from gridnext.densenet import DenseNet as RefDenseNet                  # noqa: E402
from gridnext.training import train_spotwise as ref_spotwise           # noqa: E402
from gridnext.training import train_gridwise as ref_gridwise           # noqa: E402
from gridnext import gridnet_models as ref_gm                          # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(1)          # deterministic summation order


def npy(t):
    return t.detach().cpu().clone().numpy()     # clone: state_dict() tensors alias live parameters


def sd_arrays(prefix, sd):
    return {prefix + k: npy(v) for k, v in sd.items()}


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        r = fn(*a, **k)
    return r, buf.getvalue()


def randomize_bn(model, gen):
    for m in model.modules():
        if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
            with torch.no_grad():
                m.weight.copy_(1 + 0.3 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=gen))
                m.running_mean.copy_(0.2 * torch.randn(m.running_mean.shape, generator=gen))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=gen))


# ----------------------------------------------------------------------------- DenseNet
def densenet_case(name, cfg_kwargs, n, p, seed):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    model = RefDenseNet(efficient=False, **cfg_kwargs)
    randomize_bn(model, gen)
    x = torch.rand((n, 3, p, p), generator=gen)
    out_dim = cfg_kwargs.get('num_classes', 10) if cfg_kwargs.get('classify', True) else None
    arrays = {'x': npy(x), 'n': n, 'p': p}
    arrays.update(sd_arrays('sd/', model.state_dict()))

    model.eval()
    y_eval = model(x)
    arrays['eval_out'] = npy(y_eval)
    if out_dim is None:
        out_dim = y_eval.shape[1]
    labels = torch.randint(0, out_dim, (n,), generator=gen)
    arrays['labels'] = npy(labels)
    # eval-mode BN gradients (what train_gridwise + f_opt produces: training.py:126 keeps f in eval)
    model.zero_grad()
    xg = x.clone().requires_grad_(True)
    loss = nn.functional.cross_entropy(model(xg), labels)
    loss.backward()
    arrays['eval_loss'] = loss.item()
    arrays['evalgrad/x'] = npy(xg.grad)
    for k, p_ in model.named_parameters():
        if p_.grad is not None:                    # classify=False leaves the classifier unused
            arrays['evalgrad/' + k] = npy(p_.grad)
    # train-mode BN forward + gradients + running-stat update (train_spotwise)
    model.train()
    model.zero_grad()
    y_train = model(x)
    loss = nn.functional.cross_entropy(y_train, labels)
    loss.backward()
    arrays['train_out'] = npy(y_train)
    arrays['train_loss'] = loss.item()
    for k, p_ in model.named_parameters():
        if p_.grad is not None:
            arrays['traingrad/' + k] = npy(p_.grad)
    arrays.update(sd_arrays('post/', {k: v for k, v in model.state_dict().items()
                                     if 'running' in k or 'num_batches' in k}))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrays)
    print(name, 'eval_out', arrays['eval_out'].shape, 'train_loss %.6f' % arrays['train_loss'])


def densenet121_closed_form():
    cfg = odn.DenseNetCfg(num_classes=8, **odn.DENSENET121)
    model = RefDenseNet(efficient=False, num_classes=8, **odn.DENSENET121)
    model.load_state_dict(odn.closed_form_state(cfg))
    model.eval()
    arrays = {}
    with torch.no_grad():
        for p, n in ((128, 2), (64, 3)):
            x = odn.closed_form_images(n, p)
            arrays['eval_out_p%d' % p] = npy(model(x))
    model.train()
    x = odn.closed_form_images(4, 64)
    with torch.no_grad():
        arrays['train_out_p64'] = npy(model(x))
    np.savez_compressed(os.path.join(OUT, 'densenet121_closedform.npz'), **arrays)
    print('densenet121_closedform', {k: (v.shape, float(np.abs(v).max())) for k, v in arrays.items()})


# ----------------------------------------------------------------------------- count MLP
def mlp_case():
    gen = torch.Generator().manual_seed(11)
    torch.manual_seed(11)
    G, C, N = 200, 8, 96
    f = count_mlp(G, C)
    randomize_bn(f, gen)
    x = torch.randint(0, 10, (N, G), generator=gen).float()
    x[::7] = 0.0                                   # background rows are all-zero (gridnet_models.py:83-86)
    labels = torch.randint(0, C, (N,), generator=gen)
    arrays = {'x': npy(x), 'labels': npy(labels)}
    arrays.update(sd_arrays('sd/', f.state_dict()))
    f.eval()
    arrays['eval_out'] = npy(f(x))
    f.train()
    f.zero_grad()
    xg = x.clone().requires_grad_(True)
    y = f(xg)
    loss = nn.functional.cross_entropy(y, labels)
    loss.backward()
    arrays['train_out'] = npy(y)
    arrays['train_loss'] = loss.item()
    arrays['traingrad/x'] = npy(xg.grad)
    for k, p_ in f.named_parameters():
        arrays['traingrad/' + k] = npy(p_.grad)
    arrays.update(sd_arrays('post/', {k: v for k, v in f.state_dict().items()
                                     if 'running' in k or 'num_batches' in k}))
    np.savez_compressed(os.path.join(OUT, 'mlp_count.npz'), **arrays)
    print('mlp_count train_loss %.6f' % arrays['train_loss'])


# ----------------------------------------------------------------------------- train_spotwise
def spotwise_cases():
    # (1) count MLP - BASELINE config 1 in miniature
    gen = torch.Generator().manual_seed(21)
    torch.manual_seed(21)
    G, C = 64, 8
    xs = torch.randint(0, 10, (320, G), generator=gen).float()
    ys = torch.randint(0, C, (320,), generator=gen)
    dl = {'train': DataLoader(TensorDataset(xs[:256], ys[:256]), batch_size=32, shuffle=False),
          'val': DataLoader(TensorDataset(xs[256:], ys[256:]), batch_size=32, shuffle=False)}
    f = count_mlp(G, C)
    arrays = {'x': npy(xs), 'y': npy(ys), 'n_train': 256, 'batch': 32, 'lr': 1e-3, 'epochs': 3}
    arrays.update(sd_arrays('init/', f.state_dict()))
    opt = torch.optim.Adam(f.parameters(), lr=1e-3)
    (f, vh, th), text = quiet(ref_spotwise, f, dl, nn.CrossEntropyLoss(), opt, num_epochs=3)
    arrays['val_history'], arrays['train_history'] = np.array(vh), np.array(th)
    arrays['stdout'] = np.array(text)
    arrays.update(sd_arrays('final/', f.state_dict()))
    np.savez_compressed(os.path.join(OUT, 'spotwise_mlp.npz'), **arrays)
    print('spotwise_mlp', th, vh)

    # (2) tiny DenseNet - BASELINE config 2 in miniature
    gen = torch.Generator().manual_seed(22)
    torch.manual_seed(22)
    cfgk = dict(growth_rate=4, block_config=(2, 2), num_init_features=8, bn_size=2, num_classes=5,
                small_inputs=False)
    xs = torch.rand((24, 3, 32, 32), generator=gen)
    ys = torch.randint(0, 5, (24,), generator=gen)
    dl = {'train': DataLoader(TensorDataset(xs[:16], ys[:16]), batch_size=8, shuffle=False),
          'val': DataLoader(TensorDataset(xs[16:], ys[16:]), batch_size=8, shuffle=False)}
    f = RefDenseNet(efficient=False, **cfgk)
    arrays = {'x': npy(xs), 'y': npy(ys), 'n_train': 16, 'batch': 8, 'lr': 1e-3, 'epochs': 2}
    arrays.update(sd_arrays('init/', f.state_dict()))
    opt = torch.optim.Adam(f.parameters(), lr=1e-3)
    (f, vh, th), text = quiet(ref_spotwise, f, dl, nn.CrossEntropyLoss(), opt, num_epochs=2)
    arrays['val_history'], arrays['train_history'] = np.array(vh), np.array(th)
    arrays['stdout'] = np.array(text)
    arrays.update(sd_arrays('final/', f.state_dict()))
    np.savez_compressed(os.path.join(OUT, 'spotwise_densenet_tiny.npz'), **arrays)
    print('spotwise_densenet_tiny', th, vh)


# ----------------------------------------------------------------------------- train_gridwise
def _grid_count_data(gen, n, G, H, W, C):
    cdat = torch.randint(0, 10, (n, G, H, W), generator=gen).float()
    ldat = torch.randint(0, C + 1, (n, H, W), generator=gen)
    cdat = cdat * (ldat > 0).unsqueeze(1).float()          # zero background spots
    return cdat, ldat


def gridwise_cases():
    C = 5
    # (a) Cartesian GridNet: every line of arithmetic is the reference's / torch's
    gen = torch.Generator().manual_seed(31)
    torch.manual_seed(31)
    G, H, W = 24, 7, 6
    cdat, ldat = _grid_count_data(gen, 6, G, H, W, C)
    cdat_hw = cdat.permute(0, 2, 3, 1).contiguous()         # GridNet wants (B, H, W, feats)
    dl = {'train': DataLoader(TensorDataset(cdat_hw[:4], ldat[:4]), batch_size=2, shuffle=False),
          'val': DataLoader(TensorDataset(cdat_hw[4:], ldat[4:]), batch_size=2, shuffle=False)}
    f = count_mlp(G, C)
    g = ref_gm.GridNet(f, (G,), (H, W), C, use_bn=True)
    for p_ in g.patch_classifier.parameters():
        p_.requires_grad = False
    arrays = {'x': npy(cdat_hw), 'y': npy(ldat), 'epochs': 2, 'lr': 1e-3, 'accum_iters': 1}
    arrays.update(sd_arrays('init/', g.state_dict()))
    g.eval()
    with torch.no_grad():
        arrays['fwd0'] = npy(g(cdat_hw[:2]))
    opt = torch.optim.Adam(g.corrector.parameters(), lr=1e-3)
    (g, vh, th), text = quiet(ref_gridwise, g, dl, nn.CrossEntropyLoss(), opt, num_epochs=2)
    arrays['val_history'], arrays['train_history'], arrays['stdout'] = np.array(vh), np.array(th), np.array(text)
    arrays.update(sd_arrays('final/', g.state_dict()))
    np.savez_compressed(os.path.join(OUT, 'gridwise_cartesian.npz'), **arrays)
    print('gridwise_cartesian', th, vh)

    # (b) GridNetHexOddr, count f frozen, accum_iters=3 (reference wrapper + oracle hex arithmetic)
    gen = torch.Generator().manual_seed(32)
    torch.manual_seed(32)
    G, H, W = 24, 8, 6
    cdat, ldat = _grid_count_data(gen, 7, G, H, W, C)
    dl = {'train': DataLoader(TensorDataset(cdat[:5], ldat[:5]), batch_size=1, shuffle=False),
          'val': DataLoader(TensorDataset(cdat[5:], ldat[5:]), batch_size=1, shuffle=False)}
    f = count_mlp(G, C)
    randomize_bn(f, gen)
    g = ref_gm.GridNetHexOddr(f, (G,), (H, W), C, use_bn=True)
    for p_ in g.patch_classifier.parameters():
        p_.requires_grad = False
    arrays = {'x': npy(cdat), 'y': npy(ldat), 'epochs': 2, 'lr': 1e-3, 'accum_iters': 3}
    arrays.update(sd_arrays('init/', g.state_dict()))
    g.eval()
    with torch.no_grad():
        arrays['pp0'] = npy(g.patch_predictions(cdat[:2]))
        arrays['fwd0'] = npy(g(cdat[:2]))
    opt = torch.optim.Adam(g.corrector.parameters(), lr=1e-3)
    (g, vh, th), text = quiet(ref_gridwise, g, dl, nn.CrossEntropyLoss(), opt, num_epochs=2, accum_iters=3)
    arrays['val_history'], arrays['train_history'], arrays['stdout'] = np.array(vh), np.array(th), np.array(text)
    arrays.update(sd_arrays('final/', g.state_dict()))
    np.savez_compressed(os.path.join(OUT, 'gridwise_hexoddr.npz'), **arrays)
    print('gridwise_hexoddr', th, vh)

    # (e) GridNetHexOddr with a TRAINABLE count f (f_opt) and atonce_patch_limit=7: checkpointed chunks
    gen = torch.Generator().manual_seed(34)
    torch.manual_seed(34)
    G, H, W = 24, 8, 6
    cdat, ldat = _grid_count_data(gen, 5, G, H, W, C)
    dl = {'train': DataLoader(TensorDataset(cdat[:3], ldat[:3]), batch_size=1, shuffle=False),
          'val': DataLoader(TensorDataset(cdat[3:], ldat[3:]), batch_size=1, shuffle=False)}
    f = count_mlp(G, C)
    randomize_bn(f, gen)
    g = ref_gm.GridNetHexOddr(f, (G,), (H, W), C, use_bn=True, atonce_patch_limit=7)
    arrays = {'x': npy(cdat), 'y': npy(ldat), 'epochs': 2, 'lr': 1e-3, 'accum_iters': 1, 'atonce_patch_limit': 7}
    arrays.update(sd_arrays('init/', g.state_dict()))
    opt = torch.optim.Adam(g.corrector.parameters(), lr=1e-3)
    f_opt = torch.optim.Adam(g.patch_classifier.parameters(), lr=1e-4)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        (g, vh, th), text = quiet(ref_gridwise, g, dl, nn.CrossEntropyLoss(), opt, num_epochs=2, f_opt=f_opt)
    arrays['val_history'], arrays['train_history'], arrays['stdout'] = np.array(vh), np.array(th), np.array(text)
    arrays.update(sd_arrays('final/', g.state_dict()))
    np.savez_compressed(os.path.join(OUT, 'gridwise_hexoddr_fopt.npz'), **arrays)
    print('gridwise_hexoddr_fopt', th, vh)

    # (c) GridNetHexMM in tutorial mode (image f frozen, corrector optimised) and
    # (d) with f_opt over both classifiers, accum_iters=2.  (atonce_patch_limit cannot be combined with a
    #     trainable f in GridNetHexMM: the checkpoint recompute runs after patch_classifier has been
    #     re-pointed at the image network and the reference raises - so chunking is covered by (e).)
    for tag, use_fopt, accum, limit in (('gridwise_hexmm_tutorial', False, 1, None),
                                        ('gridwise_hexmm_fopt', True, 2, None)):
        gen = torch.Generator().manual_seed(33)
        torch.manual_seed(33)
        G, H, W, P = 20, 6, 4, 32
        n = 4
        imdat = torch.rand((n, H, W, 3, P, P), generator=gen)
        cdat = torch.randint(0, 10, (n, G, H, W), generator=gen).float()
        ldat = torch.randint(0, C + 1, (n, H, W), generator=gen)
        dset = [((imdat[i], cdat[i]), ldat[i]) for i in range(n)]
        dl = {'train': DataLoader(dset[:3], batch_size=1, shuffle=False),
              'val': DataLoader(dset[3:], batch_size=1, shuffle=False)}
        cfgk = dict(growth_rate=4, block_config=(2, 2), num_init_features=8, bn_size=2, num_classes=C,
                    small_inputs=False)
        f_img = RefDenseNet(efficient=False, **cfgk)
        randomize_bn(f_img, gen)
        f_cnt = count_mlp(G, C)
        g = ref_gm.GridNetHexMM(f_img, f_cnt, (3, P, P), (G,), (H, W), C, atonce_patch_limit=limit)
        arrays = {'x_img': npy(imdat), 'x_cnt': npy(cdat), 'y': npy(ldat), 'epochs': 2, 'lr': 1e-3,
                  'accum_iters': accum, 'atonce_patch_limit': -1 if limit is None else limit}
        arrays.update(sd_arrays('init/', g.state_dict()))
        g.eval()
        with torch.no_grad():
            arrays['pp0'] = npy(g.patch_predictions([imdat[:2], cdat[:2]]))
            arrays['fwd0'] = npy(g([imdat[:2], cdat[:2]]))
        opt = torch.optim.Adam(g.corrector.parameters(), lr=1e-3)
        f_opt = None
        if use_fopt:
            f_opt = torch.optim.Adam(list(g.image_classifier.parameters()) +
                                     list(g.count_classifier.parameters()), lr=1e-4)
        else:
            for p_ in g.patch_classifier.parameters():          # Tutorial_multimodal.ipynb cell 27
                p_.requires_grad = False
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            (g, vh, th), text = quiet(ref_gridwise, g, dl, nn.CrossEntropyLoss(), opt, num_epochs=2,
                                      f_opt=f_opt, accum_iters=accum)
        arrays['val_history'], arrays['train_history'], arrays['stdout'] = np.array(vh), np.array(th), np.array(text)
        arrays['patch_classifier_is_image'] = int(g.patch_classifier is g.image_classifier)
        arrays['count_training_flag'] = int(g.count_classifier.training)
        arrays.update(sd_arrays('final/', g.state_dict()))
        np.savez_compressed(os.path.join(OUT, tag + '.npz'), **arrays)
        print(tag, th, vh)


# ----------------------------------------------------------------------------- masked CE on saved maps
class _LogitsAsModel(nn.Module):
    """Feeds a saved logit grid through the reference loop so its inline masking/CE runs."""
    def __init__(self, logits):
        super().__init__()
        self.z = nn.Parameter(logits.clone())
        self.patch_classifier = nn.Identity()

    def forward(self, x):
        return self.z.unsqueeze(0).expand(x.shape[0], -1, -1, -1)


class _GradTap(torch.optim.SGD):
    def step(self, closure=None):
        self.tapped = [p.grad.detach().clone() for g in self.param_groups for p in g['params']]
        return super().step(closure)


def masked_ce_case():
    base = os.path.join(REF, 'outputs', 'maynard_imgpred_maps', 'maynard_151507_%s_oddr.npy')
    logits = torch.from_numpy(np.load(base % 'logits')).float()
    smax = np.load(base % 'smax')
    true = torch.from_numpy(np.load(base % 'true')).long()
    arrays = {'logits': npy(logits), 'smax_saved': smax, 'labels': npy(true)}
    for accum in (1, 4):
        m = _LogitsAsModel(logits)
        opt = _GradTap(m.parameters(), lr=0.0)
        ds = TensorDataset(torch.zeros(1, 1), true.unsqueeze(0))
        dl = {'train': DataLoader(ds, batch_size=1), 'val': DataLoader(ds, batch_size=1)}
        (m, vh, th), text = quiet(ref_gridwise, m, dl, nn.CrossEntropyLoss(), opt, num_epochs=1,
                                  accum_iters=accum)
        arrays['loss_accum%d' % accum] = th[0]
        arrays['grad_accum%d' % accum] = npy(opt.tapped[0])
        arrays['stdout_accum%d' % accum] = np.array(text)
    np.savez_compressed(os.path.join(OUT, 'masked_ce_maynard.npz'), **arrays)
    print('masked_ce_maynard loss', arrays['loss_accum1'], 'n_fg', int((true > 0).sum()))


if __name__ == '__main__':
    densenet_case('densenet_tiny_large',
                  dict(growth_rate=4, block_config=(2, 2), num_init_features=8, bn_size=2,
                       num_classes=5, small_inputs=False), n=4, p=32, seed=1)
    densenet_case('densenet_tiny_small',
                  dict(growth_rate=6, block_config=(2, 3, 2), num_init_features=10, bn_size=2,
                       num_classes=7, small_inputs=True, classify=False, compression=0.5), n=3, p=16, seed=2)
    densenet121_closed_form()
    mlp_case()
    spotwise_cases()
    gridwise_cases()
    masked_ce_case()
    print('fixtures written to', OUT)
