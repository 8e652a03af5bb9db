"""CPU: the hex-conv oracle's geometry, pinned by the reference's coordinate helpers.

hexagdly itself is absent (hex arithmetic "parity unpinned"), so what CAN be pinned is:
 * the two independent restatements (tap gather vs sub-convolution form) agree;
 * the neighbour set equals "cells at unit distance" under the reference's own
   odd-right -> pseudo-hex -> true-hex maps (gridnext/utils.py:64-85 formulas, restated below),
   which is also the adjacency rule of gridnext/graph_datasets.py:155-157 (<= 6 neighbours);
 * rot90(1,[3,2])+flip([3]) (gridnet_models.py:178-179) is a transpose and :184-185 its inverse.
"""
import math

import pytest
import torch

from oracle import hexconv as ohex


def oddr_to_true_hex(col, row):
    # utils.py:73-79 (oddr -> pseudo-hex) then utils.py:82-85 (pseudo-hex -> true hex)
    px = 2 * col if row % 2 == 0 else 2 * col + 1
    return px / 2.0, row * math.sqrt(3) / 2.0


@pytest.mark.parametrize("R,C", [(5, 4), (6, 7), (7, 7), (64, 78), (1, 1), (2, 3)])
def test_gather_and_subconv_forms_agree(R, C):
    g = torch.Generator().manual_seed(R * 100 + C)
    x = torch.randn(2, 3, R, C, generator=g, dtype=torch.float64)
    k0 = torch.randn(4, 3, 3, 1, generator=g, dtype=torch.float64)
    k1 = torch.randn(4, 3, 2, 2, generator=g, dtype=torch.float64)
    b = torch.randn(4, generator=g, dtype=torch.float64)
    a = ohex.hexconv_gather(x, k0, k1, b)
    s = ohex.hexconv_subconv(x, k0, k1, b)
    assert torch.allclose(a, s, rtol=1e-12, atol=1e-12)


def test_all_ones_counts_in_grid_neighbours():
    H, W = 78, 64
    x = torch.ones(1, 1, H, W, dtype=torch.float64)
    k0 = torch.ones(1, 1, 3, 1, dtype=torch.float64)
    k1 = torch.ones(1, 1, 2, 2, dtype=torch.float64)
    out = ohex.hexconv_oddr(x, k0, k1, torch.ones(1, dtype=torch.float64))[0, 0]
    # expected = 1 (bias) + 1 (self) + number of in-grid cells at unit distance (reference geometry)
    for (y, xx) in [(0, 0), (0, 63), (77, 0), (77, 63), (1, 0), (1, 63), (10, 10), (11, 10), (0, 5), (77, 30)]:
        cx, cy = oddr_to_true_hex(xx, y)
        n = 0
        for yy in range(max(0, y - 2), min(H, y + 3)):
            for x2 in range(max(0, xx - 2), min(W, xx + 3)):
                if (yy, x2) != (y, xx):
                    px, py = oddr_to_true_hex(x2, yy)
                    if abs(math.hypot(px - cx, py - cy) - 1.0) < 1e-9:
                        n += 1
        assert n <= 6
        assert out[y, xx].item() == 2 + n, (y, xx, out[y, xx].item(), n)
    assert out[10, 10].item() == 8.0


def test_neighbour_table_is_unit_distance_set():
    # Visium odd-right (row y, col x): after the transpose, hexagdly column = y, hexagdly row = x.
    H, W = 9, 8
    for y in range(H):
        for x in range(W):
            want = set()
            cx, cy = oddr_to_true_hex(x, y)
            for yy in range(H):
                for xx in range(W):
                    px, py = oddr_to_true_hex(xx, yy)
                    if abs(math.hypot(px - cx, py - cy) - 1.0) < 1e-9:
                        want.add((yy, xx))
            got = set()
            for dr, dc, tid, a, b in ohex.hex_taps(y % 2):
                yy, xx = y + dc, x + dr           # hexagdly (row, col) offsets -> Visium (x, y)
                if (dr, dc) != (0, 0) and 0 <= yy < H and 0 <= xx < W:
                    got.add((yy, xx))
            assert got == want, (y, x, got, want)


def test_reference_rotation_pair_is_a_transpose():
    x = torch.arange(2 * 3 * 5 * 4, dtype=torch.float32).reshape(2, 3, 5, 4)
    to_hex = torch.flip(torch.rot90(x, 1, [3, 2]), [3])
    assert torch.equal(to_hex, x.transpose(2, 3))
    back = torch.rot90(torch.flip(to_hex, [3]), 1, [2, 3])
    assert torch.equal(back, x)


def test_single_impulse_response_matches_weights():
    # an impulse at an interior cell spreads each tap weight to the mirrored neighbour
    k0 = torch.tensor([1., 2., 3.]).view(1, 1, 3, 1)
    k1 = torch.tensor([[4., 5.], [6., 7.]]).view(1, 1, 2, 2)
    x = torch.zeros(1, 1, 6, 6)
    x[0, 0, 2, 2] = 1.0                       # even column
    y = ohex.hexconv_subconv(x, k0, k1)[0, 0]
    # own column: output(r) += k0[a]*x(r-1+a): impulse at r=2 reaches r=3 (a=0), r=2 (a=1), r=1 (a=2)
    assert (y[3, 2], y[2, 2], y[1, 2]) == (1., 2., 3.)
    # neighbours of the impulse are ODD columns 1 and 3; an odd column's output at r reads (r, c±1) [a=0]
    # and (r+1, c±1) [a=1]; its left input (b=0) is column c-1, right input (b=1) is column c+1
    assert (y[2, 3], y[1, 3]) == (4., 6.)     # column 3 sees the impulse on its LEFT  (b=0)
    assert (y[2, 1], y[1, 1]) == (5., 7.)     # column 1 sees the impulse on its RIGHT (b=1)
    assert y.sum().item() == 28.0
