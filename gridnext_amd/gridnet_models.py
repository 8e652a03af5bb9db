"""Grid models: a spot classifier f applied to every position of an ST array, then a corrector g.

Drop-in for /root/reference/gridnext/gridnet_models.py: same class names, constructor signatures
(GridNet :24-25, GridNetHex :123-124, GridNetHexOddr, GridNetHexMM :194-195), same attributes used by the
training loops and notebooks (`patch_classifier`, `corrector`, `patch_predictions`, `forward`) and the same
state_dict keys (`bg_const`, `dummy_tensor`, `patch_classifier.*`, `corrector.N.{kernel0,kernel1,bias_tensor}`,
`corrector.N.{weight,bias,running_*}`; GridNetHexMM additionally `image_classifier.*`, `count_classifier.*`).

How it runs here (MI355X-first, not a translation):
  * everything after f is channels-last: f writes rows [spot][feature]; g, BN and the masked CE read rows.  The
    (B, C, H, W) tensors the API returns are zero-copy permuted views of that storage.
  * GridNetHexOddr's rot90/flip copies (:178-185) are replaced by running the hex stencil directly on the
    odd-right grid (kernel `mode 1`); a 4-D count grid (B, genes, H, W) is consumed in place by the first
    Linear's GEMM (no permute+copy of the 40 MB array, :167-169).
  * a stock `nn.Sequential` of Linear/BatchNorm1d/ReLU (the tutorials' count MLP) is executed by the HIP MLP
    pipeline using the user's own parameters; `gridnext_amd.DenseNet` runs its own HIP forward.  Any other
    classifier module is simply called (torch's kernels) - outside the north-star path.
Reference quirks kept on purpose: f runs on background spots too; GridNetHexMM re-points
`patch_classifier`/`patch_shape`/`f_dim` per modality and leaves them on the image network.
"""
import torch
import torch.nn as nn
import torch.utils.checkpoint as cp

from . import functional as GF
from . import hexconv as hexagdly
from .densenet import DenseNet


def _spot_rows(classifier, spots, count_grid=None):
    """f over a flat batch of spots -> rows [n_spots, f_dim].
    `count_grid` (B, genes, H*W) is the un-permuted source of `spots` when available."""
    if GF.is_hip_sequential(classifier) and (spots if count_grid is None else count_grid).is_cuda:
        if count_grid is not None:
            return GF.sequential_forward(classifier, count_grid, kmajor=True)
        return GF.sequential_forward(classifier, spots.reshape(spots.shape[0], -1))
    return classifier(spots)


class GridNet(nn.Module):
    """Cartesian-grid model (reference :23-117).  f goes through the HIP kernels; the Cartesian Conv2d
    corrector (:51-66) is kept as stock torch layers - the Visium path (GridNetHex*) is the accelerated one."""

    def __init__(self, patch_classifier, patch_shape, grid_shape, n_classes,
                 use_bn=True, atonce_patch_limit=None, f_dim=None):
        super().__init__()
        self.patch_shape = tuple(patch_shape)
        self.grid_shape = tuple(grid_shape)
        self.n_classes = n_classes
        self.patch_classifier = patch_classifier
        self.use_bn = use_bn
        self.atonce_patch_limit = atonce_patch_limit
        self.f_dim = n_classes if f_dim is None else f_dim
        self.corrector = self._init_corrector()
        self.register_buffer("bg_const", torch.zeros((1, self.f_dim)))
        self.register_buffer("dummy_tensor", torch.ones(1, dtype=torch.float32))

    def _init_corrector(self):
        c = self.n_classes
        layers = []
        for idx, (c_in, k) in enumerate(((self.f_dim, 3), (c, 5), (c, 5), (c, 3))):
            layers.append(nn.Conv2d(c_in, c, k, padding=k // 2))
            if idx < 3:
                if self.use_bn:
                    layers.append(nn.BatchNorm2d(c))
                layers.append(nn.ReLU())
        return nn.Sequential(*layers)

    # -- f over the grid ------------------------------------------------------------------------------
    def _f_rows(self, x, count_grid=None):
        """rows [B*H*W, f_dim] of f over every grid position (background included, reference :83-86)."""
        f = self.patch_classifier
        spots = None if count_grid is not None else x.reshape((-1,) + tuple(self.patch_shape))
        n = count_grid.shape[0] * count_grid.shape[2] if count_grid is not None else spots.shape[0]
        lim = self.atonce_patch_limit
        if isinstance(f, DenseNet):
            on_tape = f.training or (torch.is_grad_enabled() and any(p.requires_grad for p in f.parameters()))
            if not on_tape:
                keep, f.atonce = f.atonce, lim     # chunking happens inside the HIP eval forward
                try:
                    return f(spots)
                finally:
                    f.atonce = keep                # the user's own setting is not ours to change
            if lim is None or lim >= n:
                return f(spots)
            # gradient / train-mode path: chunks of `atonce_patch_limit` spots, each CHECKPOINTED as in the reference (:88-104):
            # its forward keeps no tape, its backward recomputes it - one tape alive at a time.  In train mode BatchNorm
            # batch statistics are per chunk there too (running statistics move once per chunk here, twice there).
            from .densenet_train import densenet_recompute
            if spots.dtype not in (torch.uint8, torch.float32):
                spots = spots.float()
            return torch.cat([densenet_recompute(f, spots.narrow(0, s0, min(lim, n - s0))) for s0 in range(0, n, lim)], 0)
        if lim is None or lim >= n:
            return _spot_rows(f, spots, count_grid)
        if spots is None:
            spots = count_grid.permute(0, 2, 1).reshape(n, -1)
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in f.parameters())
        parts = []
        for s0 in range(0, n, lim):
            piece = spots.narrow(0, s0, min(lim, n - s0))
            if needs_grad and not GF.is_hip_sequential(f):
                # generic modules keep the reference's recompute-in-backward behaviour (:95-98)
                parts.append(cp.checkpoint(lambda p, d: f(p), piece, self.dummy_tensor.requires_grad_(True),
                                           use_reentrant=True))
            else:
                parts.append(_spot_rows(f, piece))
        return torch.cat(parts, 0)

    def _grid_nhwc(self, x):
        rows = self._f_rows(x)
        return rows.reshape((-1,) + self.grid_shape + (self.f_dim,))

    def patch_predictions(self, x):
        return self._grid_nhwc(x).permute(0, 3, 1, 2)

    def forward(self, x):
        return self.corrector(self.patch_predictions(x))


class GridNetHex(GridNet):
    """Hexagonal corrector on a grid given in hexagdly addressing (reference :122-148)."""
    _oddr = False

    def _init_corrector(self):
        layers = [hexagdly.Conv2d(self.f_dim, 32, kernel_size=1, stride=1, bias=True),
                  hexagdly.Conv2d(32, 32, kernel_size=1, stride=1, bias=True)]
        if self.use_bn:
            layers.append(nn.BatchNorm2d(32))
        layers.append(nn.ReLU())
        layers += [hexagdly.Conv2d(32, 32, kernel_size=1, stride=1, bias=True),
                   hexagdly.Conv2d(32, 32, kernel_size=1, stride=1, bias=True)]
        if self.use_bn:
            layers.append(nn.BatchNorm2d(32))
        layers.append(nn.ReLU())
        layers.append(hexagdly.Conv2d(32, self.n_classes, kernel_size=1, stride=1, bias=True))
        return nn.Sequential(*layers)

    def _correct_nhwc(self, grid):
        """Run the corrector's layers on channels-last data [B, H, W, C]."""
        mods = list(self.corrector)
        i = 0
        while i < len(mods):
            m = mods[i]
            B, H, W, C = grid.shape
            if isinstance(m, hexagdly.Conv2d):
                grid = m.forward_nhwc(grid, self._oddr)
                i += 1
            elif isinstance(m, nn.BatchNorm2d):
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                grid = GF.batch_norm_relu(grid.reshape(-1, C), m, relu=fuse).reshape(B, H, W, C)
                i += 2 if fuse else 1
            elif isinstance(m, nn.ReLU):
                grid = GF.relu_rows(grid.reshape(-1, C)).reshape(B, H, W, C)
                i += 1
            else:                                   # user-inserted layer: fall back to its own forward
                grid = m(grid.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
                i += 1
        return grid

    def forward_nhwc(self, x):
        """Channels-last logits [B, H, W, n_classes] (what the fused masked CE consumes)."""
        return self._correct_nhwc(self._grid_nhwc(x))

    def forward(self, x):
        return self.forward_nhwc(x).permute(0, 3, 1, 2)


class GridNetHexOddr(GridNetHex):
    """Visium odd-right grids (reference :159-187): 1-D spot features arrive as (B, feats, H, W),
    n-D ones as (B, H, W, feats...); output (B, n_classes, H, W)."""
    _oddr = True

    def _grid_nhwc(self, x):
        if x.dim() == 4:
            B, G, H, W = x.shape
            f = self.patch_classifier
            lim = self.atonce_patch_limit
            if GF.is_hip_sequential(f) and x.is_cuda and (lim is None or lim >= B * H * W):
                rows = self._f_rows(None, count_grid=x.reshape(B, G, H * W))
            else:
                rows = self._f_rows(x.permute(0, 2, 3, 1))
            return rows.reshape((-1,) + self.grid_shape + (self.f_dim,))
        return super()._grid_nhwc(x)


class GridNetHexMM(GridNetHexOddr):
    """Image + count classifiers, concatenated count-first along features (reference :193-235)."""

    def __init__(self, image_classifier, count_classifier, image_shape, count_shape, grid_shape, n_classes,
                 use_bn=True, atonce_patch_limit=None, image_f_dim=None, count_f_dim=None):
        image_f_dim = n_classes if image_f_dim is None else image_f_dim
        count_f_dim = n_classes if count_f_dim is None else count_f_dim
        super().__init__(image_classifier, image_shape, grid_shape, n_classes, use_bn, atonce_patch_limit,
                         image_f_dim + count_f_dim)
        self.image_classifier = image_classifier
        self.count_classifier = count_classifier
        self.image_shape, self.count_shape = tuple(image_shape), tuple(count_shape)
        self.image_f_dim, self.count_f_dim = image_f_dim, count_f_dim

    def _set_mode(self, mode):
        if mode == 'image':
            self.patch_classifier, self.patch_shape, self.f_dim = \
                self.image_classifier, self.image_shape, self.image_f_dim
        elif mode == 'count':
            self.patch_classifier, self.patch_shape, self.f_dim = \
                self.count_classifier, self.count_shape, self.count_f_dim
        else:
            self.f_dim = self.count_f_dim + self.image_f_dim

    def _grid_nhwc(self, x):
        x_image, x_count = x
        self._set_mode('count')
        g_count = GridNetHexOddr._grid_nhwc(self, x_count)
        self._set_mode('image')
        g_image = GridNetHexOddr._grid_nhwc(self, x_image)
        self._set_mode('concat')
        return torch.cat((g_count, g_image), dim=3)
