"""Image-patch datasets feeding the f∘g path (API surface of /root/reference/gridnext/image_datasets.py).

`PatchDataset` (:20-122): one item per annotated spot image -> (patch (3, P, P) float32 in [0, 1], label int64);
`PatchGridDataset` (:125-232): one item per array -> (patches (H_ST, W_ST, 3, P, P) float32, zeros where the array has
no spot image; labels (H_ST, W_ST) int64, class id + 1, 0 = background).  Spot images are files named
"*_<array_x>_<array_y>.<ext>" inside one directory per array; Visium array coordinates are mapped to odd-right grid
positions.  The default transform is torchvision's `ToTensor()` semantics (uint8 HWC -> float CHW / 255) written out
here because torchvision is not a dependency of this package; any callable taking a PIL image can be passed instead.
"""
import glob
import os
import re
from pathlib import Path

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from .count_datasets import _check_files, _label_names, read_annotfile
from .utils import pseudo_hex_to_oddr

Image.MAX_IMAGE_PIXELS = None


def to_tensor(img):
    """PIL image -> float32 CHW in [0, 1] (what torchvision.transforms.ToTensor does for 8-bit images)."""
    arr = np.array(img)                       # copy: PIL's buffer is read-only
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
    return t.float().div(255) if t.dtype == torch.uint8 else t.float()


def to_tensor_u8(img):
    """PIL image -> uint8 CHW, NOT divided: the / 255 of ToTensor then happens on the device, inside the stem kernel
    (`gridnext_amd.DenseNet` takes uint8 patches; a quarter of the bytes over PCIe and out of HBM)."""
    arr = np.array(img)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    if arr.dtype != np.uint8:
        raise TypeError("raw_uint8 needs 8-bit images (got %s)" % arr.dtype)
    return torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).contiguous()


def _spot_labels(annot_file, position_file, Visium, afile_delim, class_names):
    if Visium:
        coords, strs = read_annotfile(annot_file, position_file=position_file, Visium=True, afile_delim=afile_delim)
        return dict(zip(coords, np.searchsorted(class_names, strs)))
    coords, ids = read_annotfile(annot_file, Visium=False, afile_delim=afile_delim)
    return dict(zip(coords, ids))


class PatchDataset(Dataset):
    def __init__(self, img_files, annot_files=None, position_files=None, Visium=True,
                 img_transforms=None, afile_delim=',', img_ext='jpg', verbose=False, raw_uint8=False):
        super().__init__()
        self.raw_uint8 = raw_uint8            # addition: yield uint8 patches (ToTensor's / 255 is fused into the stem kernel)
        _check_files(img_files, annot_files, position_files, Visium, 'img_files')
        names = None
        if Visium and annot_files is not None:
            names = _label_names(annot_files, position_files, ',')
            self.classes = names
        self.afile_delim = afile_delim
        self.imgpath_mapping, self.annotations = [], []
        skipped = 0
        if annot_files is not None:
            for i, (imdir, afile) in enumerate(zip(img_files, annot_files)):
                lookup = _spot_labels(afile, position_files[i] if Visium else None, Visium, afile_delim, names)
                for imfile in glob.glob(os.path.join(imdir, '*.' + img_ext)):
                    cstr = '_'.join(Path(imfile).stem.split('_')[-2:])
                    if cstr not in lookup:
                        skipped += 1
                        if verbose:
                            print(cstr, 'image patch missing annotation (skipping)')
                        continue
                    self.annotations.append(lookup[cstr])
                    self.imgpath_mapping.append(imfile)
        else:
            for imdir in img_files:
                self.imgpath_mapping += glob.glob(os.path.join(imdir, '*.' + img_ext))
        self.preprocess = (to_tensor_u8 if raw_uint8 else to_tensor) if img_transforms is None else img_transforms
        if annot_files is not None and verbose:
            print('%d image patches without annotation' % skipped)

    def __len__(self):
        return len(self.imgpath_mapping)

    def __getitem__(self, idx):
        img = self.preprocess(Image.open(self.imgpath_mapping[idx]))
        label = torch.tensor(self.annotations[idx]).long() if len(self.annotations) > 0 else torch.empty(0)
        return (img if self.raw_uint8 else img.float()), label


class PatchGridDataset(Dataset):
    def __init__(self, img_files, annot_files=None, position_files=None, Visium=True,
                 img_transforms=None, afile_delim=',', img_ext='jpg', h_st=78, w_st=64, raw_uint8=False):
        super().__init__()
        self.raw_uint8 = raw_uint8            # addition: yield a uint8 patch grid (245 MB instead of 981 MB at 128 px)
        _check_files(img_files, annot_files, position_files, Visium, 'img_files')
        if Visium and annot_files is not None:
            self.classes = _label_names(annot_files, position_files, ',')
        self.img_files, self.annot_files, self.position_files = img_files, annot_files, position_files
        self.h_st, self.w_st, self.Visium = h_st, w_st, Visium
        self.afile_delim, self.img_ext = afile_delim, img_ext
        self.preprocess = (to_tensor_u8 if raw_uint8 else to_tensor) if img_transforms is None else img_transforms

    def __len__(self):
        return len(self.img_files)

    def __getitem__(self, idx):
        lookup = None
        if self.annot_files is not None:
            lookup = _spot_labels(self.annot_files[idx], self.position_files[idx] if self.Visium else None,
                                  self.Visium, self.afile_delim, getattr(self, 'classes', None))
        grid = None
        labels = torch.zeros((self.h_st, self.w_st), dtype=torch.int64)
        pattern = re.compile(r".*_(\d+)_(\d+)\.%s$" % re.escape(self.img_ext))
        for fname in sorted(os.listdir(str(self.img_files[idx]))):
            hit = pattern.match(fname)
            if hit is None:
                continue
            ax, ay = int(hit.group(1)), int(hit.group(2))
            patch = self.preprocess(Image.open(os.path.join(self.img_files[idx], fname)))
            if grid is None:
                grid = torch.zeros((self.h_st, self.w_st) + tuple(patch.shape),
                                   dtype=torch.uint8 if self.raw_uint8 else torch.float32)
            x, y = pseudo_hex_to_oddr(ax, ay) if self.Visium else (ax, ay)
            if lookup is not None:
                cstr = '%d_%d' % (ax, ay)
                if cstr in lookup:
                    labels[y, x] = int(lookup[cstr]) + 1            # 0 is reserved for background
            grid[y, x] = patch
        return (grid if self.raw_uint8 else grid.float()), labels.long()
