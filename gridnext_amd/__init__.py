"""gridnext_amd - MI355X-native implementation of GridNext's f∘g training hot path.

Public surface mirrors the reference package for that path:
    from gridnext_amd.densenet import DenseNet
    from gridnext_amd.gridnet_models import GridNet, GridNetHex, GridNetHexOddr, GridNetHexMM
    from gridnext_amd.training import train_spotwise, train_gridwise
    from gridnext_amd.multimodal_datasets import MMStackDataset, MultiModalDataset, MultiModalGridDataset
    import gridnext_amd.hexconv as hexagdly          # Conv2d(kernel_size=1, stride=1)
All arithmetic runs in hand-written gfx950 kernels behind the C ABI of include/gridnext_hip.h
(libgridnext_hip.so, built in-tree by `__graft_entry__.build()`); there is no CPU fallback.
"""
__version__ = "0.1.0"

from .densenet import DenseNet                                                     # noqa: F401
from .gridnet_models import GridNet, GridNetHex, GridNetHexOddr, GridNetHexMM      # noqa: F401
from .training import train_spotwise, train_gridwise                               # noqa: F401
from .multimodal_datasets import MMStackDataset, MultiModalDataset, MultiModalGridDataset  # noqa: F401
from .count_datasets import CountDataset, CountGridDataset                         # noqa: F401
from .image_datasets import PatchDataset, PatchGridDataset                         # noqa: F401
from .utils import all_fgd_predictions                                             # noqa: F401
