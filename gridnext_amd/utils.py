"""Evaluation and coordinate helpers of the f∘g path.

Mirrors the parts of /root/reference/gridnext/utils.py that sit on (or right next to) the hot path:
  * `all_fgd_predictions` (:20-57): forward every array, keep foreground spots, return (true, argmax, softmax).
    Here the forward runs on the HIP kernels and the softmax/argmax is one fused kernel on channels-last logits;
    list inputs (GridNetHexMM) are supported - the reference's helper crashes on them (`x.to(device)` at :29);
  * the Visium coordinate maps (:64-85).
File readers (`read_annotated_starray`, `read_annotfile`, Spaceranger finders) are host-side ETL and out of scope.
"""
import numpy as np
import torch

from . import _lib as L
from . import distributed as gdist


def softmax_argmax_rows(rows):
    """(probs [M, C], preds [M]) of channels-last logits [M, C] on a HIP device."""
    rows = rows.contiguous()
    M, C = rows.shape
    probs = torch.empty((M, C), device=rows.device, dtype=torch.float32)
    preds = torch.empty(M, device=rows.device, dtype=torch.int64)
    L.call('gnx_softmax_rows', L.ptr(rows), C, M, C, L.ptr(probs), C, L.ptr(preds, torch.int64), L.stream())
    return probs, preds


def all_fgd_predictions(dataloader, model, f_only=False):
    """Flattened predictions for all foreground spots: (true_vals, pred_vals, pred_smax)."""
    true_vals, pred_vals, pred_smax = [], [], []
    device = gdist.default_device()
    model.to(device)
    model.eval()
    for x, y in dataloader:
        x = [t.to(device) for t in x] if isinstance(x, (list, tuple)) else x.to(device)
        y = y.to(device)
        with torch.no_grad():
            if f_only:
                outputs = model.patch_predictions(x).permute(0, 2, 3, 1)
            elif hasattr(model, 'forward_nhwc'):
                outputs = model.forward_nhwc(x)
            else:
                outputs = model(x).permute(0, 2, 3, 1)
            rows = outputs.reshape(-1, outputs.shape[-1])
            labels = y.reshape(-1)
            if rows.is_cuda:
                probs, preds = softmax_argmax_rows(rows)
            else:                                   # a CPU model handed in by the caller: spell the reference out
                probs, preds = torch.softmax(rows, dim=1), torch.argmax(rows, dim=1)
            keep = (labels > 0).cpu()
            true_vals.append((labels.cpu()[keep] - 1).numpy())
            pred_vals.append(preds.cpu()[keep].numpy())
            pred_smax.append(probs.cpu()[keep].numpy())
    return np.concatenate(true_vals), np.concatenate(pred_vals), np.concatenate(pred_smax)


# ---- Visium coordinate maps (reference utils.py:64-85) ---------------------------------------------------------------
def pseudo_hex_to_oddr(col, row):
    """Visium pseudo-hex (col doubles along a row) -> odd-right (col, row)."""
    return int((col - (row % 2)) / 2), int(row)


def oddr_to_pseudo_hex(col, row):
    return int(2 * col + (row % 2)), int(row)


def pseudo_to_true_hex(col, row):
    """Cartesian centre with unit distance between neighbouring spots."""
    return col / 2, row * np.sqrt(3) / 2
