"""Oracle for the foreground-masked softmax cross-entropy of the grid loop.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows gridnext/training.py:152-160 (train_gridwise) and the same masking in
gridnext/utils.py:37-41 (all_fgd_predictions):
  outputs (B, C, H, W) -> permute(0,2,3,1) -> (B*H*W, C); labels (B,H,W) -> flat;
  keep rows with label > 0; shift labels by -1; mean CE over kept rows,
  divided by accum_iters; preds = argmax over classes.
Pinned by tests/golden/masked_ce_maynard.npz (reference loop driven on the
reference's own saved logit/label grids) and the saved softmax maps.
"""
import torch


def masked_ce(outputs, labels, accum_iters=1):
    """Returns (loss, preds_fg, labels_fg) exactly as the loop sees them."""
    C = outputs.shape[1]
    flat = outputs.permute(0, 2, 3, 1).reshape(-1, C)
    lab = labels.reshape(-1)
    keep = lab > 0
    z = flat[keep]
    t = lab[keep] - 1
    lse = torch.logsumexp(z, dim=1)
    picked = z.gather(1, t.view(-1, 1)).squeeze(1)
    loss = (lse - picked).mean() / accum_iters
    preds = z.argmax(dim=1)
    return loss, preds, t


def masked_ce_grad(outputs, labels, accum_iters=1):
    """Closed-form d loss / d outputs, (softmax - onehot) / (n_fg * accum) on foreground."""
    B, C, H, W = outputs.shape
    lab = labels.reshape(B, 1, H, W)
    fg = (lab > 0)
    n_fg = int(fg.sum())
    p = torch.softmax(outputs, dim=1)
    onehot = torch.zeros_like(outputs).scatter_(1, (lab - 1).clamp(min=0), 1.0)
    g = (p - onehot) * fg.to(outputs.dtype) / (n_fg * accum_iters)
    return g


def fgd_softmax_argmax(outputs, labels):
    """utils.py:36-47: foreground rows -> (true, argmax, softmax)."""
    C = outputs.shape[1]
    flat = outputs.permute(0, 2, 3, 1).reshape(-1, C)
    lab = labels.reshape(-1)
    keep = lab > 0
    z = flat[keep]
    return lab[keep] - 1, z.argmax(dim=1), torch.softmax(z, dim=1)
