"""Oracle for the hexagonal convolution used by the corrector g.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

What is restated
----------------
The reference builds g from ``hexagdly.Conv2d(in, out, kernel_size=1, stride=1,
bias=True)`` (gridnext/gridnet_models.py:128-148).  ``hexagdly`` is a third-party
PyPI dependency, listed WITHOUT a version pin in /root/reference/requirements.txt:11
and absent from /root/reference and from this image, so its source cannot be
run here.  **PARITY UNPINNED** for the hex arithmetic itself: the reference
holds no golden vectors at this boundary.  What follows restates the package's
published algorithm (Steppa & Holch, "HexagDLy", SoftwareX 9 (2019); the
package's ``HexBase.operation_with_single_hexbase_stride`` for size-1 kernels):

* addressing: vertically aligned columns, every second column (0-based index
  1, 3, 5, ...) is shifted DOWN by half a cell.  The reference's own plotting
  helper states the same: hexagon centre ``(x*sqrt(3)/2, -(y + (x % 2)*0.5))``
  (gridnext/hexagdly_tools.py:68).
* parameters of a size-1 layer: ``kernel0 (O, I, 3, 1)`` = the cell's own
  column (row-1, row, row+1); ``kernel1 (O, I, 2, 2)`` = the two adjacent
  columns (left, right) x two rows; ``bias_tensor (O,)`` added once.
* evaluation: one ``conv2d`` of the row-padded input with ``kernel0`` (+bias),
  plus, separately for the un-shifted and the shifted columns, a ``conv2d`` with
  ``kernel1`` at column dilation 2 and column stride 2 on suitably padded
  input, re-interleaved and added.

Neighbour table (r = row, c = column, 0-based, hexagdly addressing):
  c even:  (r-1,c) (r,c) (r+1,c) | (r-1,c-1) (r,c-1) | (r-1,c+1) (r,c+1)
  c odd :  (r-1,c) (r,c) (r+1,c) | (r,  c-1) (r+1,c-1) | (r,  c+1) (r+1,c+1)
  kernel0[..., 0|1|2, 0] <-> row-1|row|row+1 ; kernel1[..., a, b] <-> a-th of the
  two rows above, b = 0 left / 1 right.

Two independent forms are provided (`hexconv_gather` = per-tap shifted gather,
`hexconv_subconv` = the package's sub-convolution form); tests require them to
agree to fp32 rounding, and tests/test_oracle_geometry.py checks the neighbour
table against the reference's coordinate helpers (gridnext/utils.py:64-85).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def hex_taps(col_parity):
    """(dr, dc, tensor_id, a, b) for the 7 taps of a column of given parity."""
    taps = [(-1, 0, 0, 0, 0), (0, 0, 0, 1, 0), (1, 0, 0, 2, 0)]
    base = -1 if col_parity == 0 else 0
    for a in (0, 1):
        for b in (0, 1):
            taps.append((base + a, -1 + 2 * b, 1, a, b))
    return taps


def hexconv_gather(x, kernel0, kernel1, bias=None):
    """Brute-force 7-tap gather, hexagdly addressing. x: (B, I, R, C)."""
    B, I, R, C = x.shape
    O = kernel0.shape[0]
    xp = F.pad(x, (1, 1, 1, 1))
    cols = torch.arange(C, device=x.device)
    out = x.new_zeros((B, O, R, C))
    for parity in (0, 1):
        acc = x.new_zeros((B, O, R, C))
        for dr, dc, tid, a, b in hex_taps(parity):
            w = (kernel0 if tid == 0 else kernel1)[:, :, a, b]          # (O, I)
            sh = xp[:, :, 1 + dr:1 + dr + R, 1 + dc:1 + dc + C]         # x[r+dr, c+dc]
            acc = acc + torch.einsum('oi,birc->borc', w, sh)
        mask = ((cols % 2) == parity).to(x.dtype).view(1, 1, 1, C)
        out = out + acc * mask
    if bias is not None:
        out = out + bias.view(1, O, 1, 1)
    return out


def hexconv_subconv(x, kernel0, kernel1, bias=None):
    """hexagdly's own decomposition for hexbase_size=1, stride=1."""
    B, I, R, C = x.shape
    # own column: pad one row top and bottom, 3x1 kernel (+ bias, added once)
    res = F.conv2d(F.pad(x, (0, 0, 1, 1)), kernel0, bias)
    # un-shifted columns 0,2,4,...: neighbours in adjacent columns sit at rows r-1, r
    pr = max(0, 1 - ((C - 1) % 2))
    a = F.conv2d(F.pad(x, (1, pr, 1, 0)), kernel1, None, stride=(1, 2), dilation=(1, 2))
    # shifted columns 1,3,5,...: neighbours at rows r, r+1
    inter = torch.zeros_like(res)
    inter[..., 0::2] = a
    if C >= 2:                      # a single-column grid has no shifted column
        pr = max(0, 1 - ((C - 2) % 2))
        b = F.conv2d(F.pad(x, (0, pr, 0, 1)), kernel1, None, stride=(1, 2), dilation=(1, 2))
        inter[..., 1::2] = b
    return res + inter


def hexconv_oddr(x, kernel0, kernel1, bias=None, form=hexconv_gather):
    """Hex conv on a Visium odd-right grid (B, I, H_ST, W_ST).

    gridnext/gridnet_models.py:178-185 wraps the corrector in
    rot90(k=1,[3,2])+flip([3]) and its inverse, which is a transpose of the two
    grid axes (checked in tests): hexagdly's "column" is the Visium row.
    """
    return form(x.transpose(2, 3), kernel0, kernel1, bias).transpose(2, 3)


class HexConv2d(nn.Module):
    """Module form with hexagdly's parameter names/shapes and default init
    (xavier-uniform per kernel tensor, bias 0.01)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, bias=True, debug=False):
        super().__init__()
        assert kernel_size == 1 and stride == 1, "oracle covers the size-1/stride-1 layers g uses"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel0 = nn.Parameter(torch.empty(out_channels, in_channels, 3, 1))
        self.kernel1 = nn.Parameter(torch.empty(out_channels, in_channels, 2, 2))
        self.bias_tensor = nn.Parameter(torch.empty(out_channels)) if bias else None
        if debug:
            nn.init.constant_(self.kernel0, 1.0)
            nn.init.constant_(self.kernel1, 1.0)
            if bias:
                nn.init.constant_(self.bias_tensor, 1.0)
        else:
            nn.init.xavier_uniform_(self.kernel0)
            nn.init.xavier_uniform_(self.kernel1)
            if bias:
                nn.init.constant_(self.bias_tensor, 0.01)

    def forward(self, x):
        return hexconv_subconv(x, self.kernel0, self.kernel1, self.bias_tensor)
