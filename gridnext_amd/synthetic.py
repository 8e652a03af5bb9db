"""Synthetic Visium-shaped inputs, following the reference's own dummy-data pattern
(/root/reference/notebooks/Tutorial_multimodal.ipynb cell 19): uniform [0,1) patches, integer counts 0..9,
integer labels with 0 = background; background spots are zeroed as real grids are
(/root/reference/gridnext/multimodal_datasets.py:238-244).  Seeded on a CPU generator so every rank/device
sees the same bytes for the same (seed, index)."""
import torch

H_VISIUM, W_VISIUM = 78, 64          # /root/reference/gridnext/imgprocess.py:21-22


def visium_array(seed, n_genes=2000, n_classes=8, patch=128, h=H_VISIUM, w=W_VISIUM, image=True, counts=True,
                 device=None, zero_background=True):
    """One synthetic array: (x_image (h,w,3,P,P) | None, x_count (genes,h,w) | None, labels (h,w))."""
    g = torch.Generator().manual_seed(seed)
    labels = torch.randint(0, n_classes + 1, (h, w), generator=g)
    x_cnt = x_img = None
    if counts:
        x_cnt = torch.randint(0, 10, (n_genes, h, w), generator=g).float()
        if zero_background:
            x_cnt *= (labels > 0).float().unsqueeze(0)
    if image:
        if device is not None and torch.device(device).type == 'cuda':
            # 981 MB per 128-px array: generate on the device (seeded per array) instead of shipping over PCIe
            gd = torch.Generator(device=device).manual_seed(seed)
            x_img = torch.rand((h, w, 3, patch, patch), generator=gd, device=device)
        else:
            x_img = torch.rand((h, w, 3, patch, patch), generator=g)
        if zero_background:
            x_img *= (labels > 0).to(x_img.device, x_img.dtype).view(h, w, 1, 1, 1)
    if device is not None:
        labels = labels.to(device)
        x_cnt = None if x_cnt is None else x_cnt.to(device)
        x_img = None if x_img is None else x_img.to(device)
    return x_img, x_cnt, labels


def count_mlp(n_genes, n_classes):
    """The spot head of the tutorials (Tutorial_visium_count.ipynb cell 12): stock torch layers; the grid
    models and loops route it through the HIP MLP kernels."""
    import torch.nn as nn
    return nn.Sequential(
        nn.Linear(n_genes, 500), nn.Linear(500, 100), nn.BatchNorm1d(100), nn.ReLU(),
        nn.Linear(100, 100), nn.Linear(100, 50), nn.BatchNorm1d(50), nn.ReLU(),
        nn.Linear(50, n_classes))
