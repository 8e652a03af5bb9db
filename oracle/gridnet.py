"""Oracle for the grid models (f applied over every spot, then the corrector g).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates gridnext/gridnet_models.py:
  GridNet          :23-117   flatten grid -> f on EVERY position (background too, :83-86)
                             -> (B, f_dim, H, W) -> Cartesian Conv2d corrector (:51-66)
  GridNetHex       :122-148  corrector = 5 hex convs (f_dim->32->32, BN, ReLU, 32->32->32, BN, ReLU, 32->C)
  GridNetHexOddr   :159-187  4-D (count) inputs permuted (0,2,3,1) first; corrector wrapped in
                             rot90(1,[3,2])+flip([3]) ... flip([3])+rot90(1,[2,3])
  GridNetHexMM     :193-235  two classifiers; `patch_classifier`, `patch_shape`, `f_dim` are
                             RE-POINTED per modality, count first then image, left on image

State-dict keys match the reference (`bg_const`, `dummy_tensor`, `patch_classifier.*`,
`corrector.N.*`, and for MM additionally `image_classifier.*`, `count_classifier.*`).
Pinned by tests/golden/gridwise_*.npz: tools/gen_golden.py drives the reference's own
classes (with `oracle.hexconv.HexConv2d` injected in place of the absent hexagdly module,
so everything but the hex arithmetic is the reference's code).
"""
import torch
import torch.nn as nn
import torch.utils.checkpoint as cp

from .hexconv import HexConv2d


class GridNet(nn.Module):
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
        # gridnet_models.py:42-48: two buffers that exist only to keep checkpointing alive
        self.register_buffer("bg_const", torch.zeros((1, self.f_dim)))
        self.register_buffer("dummy_tensor", torch.ones(1, dtype=torch.float32))

    def _init_corrector(self):
        c = self.n_classes
        spec = [(self.f_dim, 3), (c, 5), (c, 5), (c, 3)]          # (in_channels, kernel)
        layers = []
        for i, (cin, k) in enumerate(spec):
            layers.append(nn.Conv2d(cin, c, k, padding=k // 2))
            if i < len(spec) - 1:
                if self.use_bn:
                    layers.append(nn.BatchNorm2d(c))
                layers.append(nn.ReLU())
        return nn.Sequential(*layers)

    def _run_f(self, spots, dummy=None):
        return self.patch_classifier(spots)

    def patch_predictions(self, x):
        spots = x.reshape((-1,) + tuple(self.patch_shape))
        lim = self.atonce_patch_limit
        if lim is None:
            preds = self._run_f(spots)
        else:
            needs_grad = any(p.requires_grad for p in self.patch_classifier.parameters())
            parts = []
            for start in range(0, len(spots), lim):
                piece = spots.narrow(0, start, min(lim, len(spots) - start))
                if needs_grad:      # gridnet_models.py:95-98 (recompute-in-backward)
                    parts.append(cp.checkpoint(self._run_f, piece, self.dummy_tensor.requires_grad_(True),
                                               use_reentrant=True))
                else:
                    parts.append(self._run_f(piece))
            preds = torch.cat(parts, 0)
        grid = preds.reshape((-1,) + tuple(self.grid_shape) + (self.f_dim,))
        return grid.permute(0, 3, 1, 2)

    def forward(self, x):
        return self.corrector(self.patch_predictions(x))


class GridNetHex(GridNet):
    def _init_corrector(self):
        layers = [HexConv2d(self.f_dim, 32), HexConv2d(32, 32)]
        if self.use_bn:
            layers.append(nn.BatchNorm2d(32))
        layers += [nn.ReLU(), HexConv2d(32, 32), HexConv2d(32, 32)]
        if self.use_bn:
            layers.append(nn.BatchNorm2d(32))
        layers += [nn.ReLU(), HexConv2d(32, self.n_classes)]
        return nn.Sequential(*layers)


class GridNetHexOddr(GridNetHex):
    def patch_predictions(self, x):
        if x.dim() == 4:                       # (B, feats, H, W) -> (B, H, W, feats)
            x = x.permute(0, 2, 3, 1)
        return GridNet.patch_predictions(self, x)

    def forward(self, x):
        grid = self.patch_predictions(x)
        to_hex = torch.flip(torch.rot90(grid, 1, [3, 2]), [3])      # :178-179
        corrected = self.corrector(to_hex)
        return torch.rot90(torch.flip(corrected, [3]), 1, [2, 3])   # :184-185


class GridNetHexMM(GridNetHexOddr):
    def __init__(self, image_classifier, count_classifier, image_shape, count_shape, grid_shape,
                 n_classes, use_bn=True, atonce_patch_limit=None, image_f_dim=None, count_f_dim=None):
        image_f_dim = n_classes if image_f_dim is None else image_f_dim
        count_f_dim = n_classes if count_f_dim is None else count_f_dim
        super().__init__(image_classifier, image_shape, grid_shape, n_classes, use_bn,
                         atonce_patch_limit, image_f_dim + count_f_dim)
        self.image_classifier = image_classifier
        self.count_classifier = count_classifier
        self.image_shape, self.count_shape = tuple(image_shape), tuple(count_shape)
        self.image_f_dim, self.count_f_dim = image_f_dim, count_f_dim

    def _point_at(self, which):
        if which == 'image':
            self.patch_classifier, self.patch_shape, self.f_dim = \
                self.image_classifier, self.image_shape, self.image_f_dim
        elif which == 'count':
            self.patch_classifier, self.patch_shape, self.f_dim = \
                self.count_classifier, self.count_shape, self.count_f_dim
        else:
            self.f_dim = self.count_f_dim + self.image_f_dim

    def patch_predictions(self, x):
        x_image, x_count = x
        self._point_at('count')
        g_count = GridNetHexOddr.patch_predictions(self, x_count)
        self._point_at('image')
        g_image = GridNetHexOddr.patch_predictions(self, x_image)
        self._point_at('concat')
        return torch.cat((g_count, g_image), dim=1)
