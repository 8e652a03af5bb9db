"""Multimodal dataset glue for the f∘g path.

`MMStackDataset` mirrors /root/reference/gridnext/multimodal_datasets.py:21-37 - the feed the reference's
multimodal tutorial actually uses (Tutorial_multimodal.ipynb cell 19): it pairs an image dataset and a count
dataset of equal length and yields ((x_image, x_count), y) with y zeroed wherever the two label grids disagree.
torch's default collate turns the (x_image, x_count) tuple into a list, which is what the training loops test for.
"""
import torch
from torch.utils.data import Dataset


class MMStackDataset(Dataset):
    def __init__(self, image_dataset, count_dataset):
        assert len(count_dataset) == len(image_dataset), "Datasets must be of the same length!"
        self.count_dataset = count_dataset
        self.image_dataset = image_dataset

    def __len__(self):
        return len(self.count_dataset)

    def __getitem__(self, idx):
        x_img, y_img = self.image_dataset[idx]
        x_cnt, y_cnt = self.count_dataset[idx]
        y = torch.where(y_img != y_cnt, torch.zeros_like(y_img), y_img)
        return (x_img, x_cnt), y


# ----------------------------------------------------------------------------------------------------------------------
# The file-backed multimodal datasets of /root/reference/gridnext/multimodal_datasets.py:141-246.  The reference keeps
# them under a "CURRENTLY DEFUNCT" banner (:135) and they are broken as shipped: they call a CountDataset /
# CountGridDataset constructor that no longer exists (positional `select_genes, h_st, w_st, Visium` where the current
# classes take `position_files, Visium, select_genes`, :205-206 vs count_datasets.py:218-219) and they yield 3-tuples that
# `train_gridwise` cannot consume.  What is rebuilt here is their DOCUMENTED contract - same constructor signatures, same
# skip rules, same item tuples - on this package's own file readers, so the names north_star lists exist and work:
#   * Splotch-formatted annotation files (one-hot matrix, annotations x spot coordinate strings; :137-139),
#   * spot images `<img_dir>/<x>_<y>.jpg` named by the same coordinate strings as the count-file header,
#   * MultiModalDataset[i]      -> (count_vec (G,), image (3,P,P) float in [0,1], label int64)              (:196-199)
#   * MultiModalGridDataset[i]  -> (counts (G,H,W), patches (H,W,3,P,P), labels (H,W) int64, 0 = background) (:246),
#     a spot being foreground only when it has BOTH image data and an annotation (:237-244).
# `training_pairs=True` (an addition) yields ((x_image, x_count), y) instead - the MMStackDataset item that the training
# loops take.
import os                                                               # noqa: E402
import re                                                               # noqa: E402

import numpy as np                                                      # noqa: E402
from PIL import Image                                                   # noqa: E402

from .count_datasets import _read_counts, read_annotated_starray       # noqa: E402
from .image_datasets import to_tensor                                   # noqa: E402
from .utils import pseudo_hex_to_oddr                                   # noqa: E402


class MultiModalDataset(Dataset):
    def __init__(self, count_files, img_files, annot_files, select_genes=None, img_transforms=None,
                 cfile_delim='\t', afile_delim='\t', training_pairs=False):
        super().__init__()
        if len(count_files) != len(img_files) or len(count_files) != len(annot_files):
            raise ValueError('Length of count_files, img_files and annot_files must match.')
        self.select_genes = select_genes
        self.cfile_delim, self.afile_delim = cfile_delim, afile_delim
        self.countfile_mapping, self.imgpath_mapping, self.cind_mapping, self.annotations = [], [], [], []
        self.training_pairs = training_pairs
        self._cache = {}
        for cfile, imdir, afile in zip(count_files, img_files, annot_files):
            with open(cfile, 'r') as fh:
                header = next(fh).strip('\n').split(cfile_delim)
            import pandas as pd
            adat = pd.read_csv(afile, header=0, index_col=0, sep=afile_delim)
            for cstr in adat.columns:
                if cstr not in header:                                  # un-annotated / mis-annotated spot (:163-166)
                    print(afile, cstr, 'missing')
                    continue
                if not np.sum(adat[cstr]) == 1:
                    print(afile, cstr, 'improper annotation')
                    continue
                imgpath = os.path.join(imdir, cstr + '.jpg')
                if not os.path.exists(imgpath):                         # spot without image data (:173-177)
                    print(imdir, cstr, 'no image data')
                    continue
                self.annotations.append(int(np.argmax(adat[cstr].values)))
                self.countfile_mapping.append(cfile)
                self.imgpath_mapping.append(imgpath)
                self.cind_mapping.append(header.index(cstr))
        self.preprocess = to_tensor if img_transforms is None else img_transforms

    def __len__(self):
        return len(self.cind_mapping)

    def __getitem__(self, idx):
        cf = self.countfile_mapping[idx]
        if cf not in self._cache:
            self._cache[cf] = _read_counts(cf, self.cfile_delim)
        mat = self._cache[cf]
        col = mat.iloc[:, self.cind_mapping[idx] - 1]                   # header column 0 is the gene-name column
        if self.select_genes is not None:
            wanted = set(self.select_genes)
            col = col[[g in wanted for g in mat.index]]
        count_vec = torch.from_numpy(np.asarray(col.values, dtype=np.float32))
        label = torch.tensor(self.annotations[idx]).long()
        img = self.preprocess(Image.open(self.imgpath_mapping[idx])).float()
        if self.training_pairs:
            return (img, count_vec), label
        return count_vec, img, label


class MultiModalGridDataset(Dataset):
    def __init__(self, count_files, img_files, annot_files, select_genes=None, h_st=78, w_st=64, Visium=True,
                 img_transforms=None, cfile_delim='\t', afile_delim='\t', training_pairs=False):
        super().__init__()
        if len(count_files) != len(img_files) or len(count_files) != len(annot_files):
            raise ValueError('Length of count_files, img_files and annot_files must match.')
        self.count_files, self.img_files, self.annot_files = count_files, img_files, annot_files
        self.select_genes, self.h_st, self.w_st, self.Visium = select_genes, h_st, w_st, Visium
        self.cfile_delim, self.afile_delim = cfile_delim, afile_delim
        self.preprocess = to_tensor if img_transforms is None else img_transforms
        self.training_pairs = training_pairs

    def __len__(self):
        return len(self.count_files)

    def __getitem__(self, idx):
        # every spot's counts at its grid position, then the Splotch annotations column by column, a spot counting as
        # annotated when its one-hot column sums to 1 (the rule of MultiModalDataset, :168; the shared reader's row-sum
        # filter, utils.py:238, empties any realistic file)
        counts, _ = read_annotated_starray(self.count_files[idx], None, select_genes=self.select_genes, h_st=self.h_st,
                                           w_st=self.w_st, Visium=self.Visium, cfile_delim=self.cfile_delim)
        import pandas as pd
        adat = pd.read_csv(self.annot_files[idx], header=0, index_col=0, sep=self.afile_delim)
        annots = np.zeros((self.h_st, self.w_st), dtype=np.int64)
        for cstr in adat.columns:
            if not np.sum(adat[cstr]) == 1:
                continue
            if self.Visium:
                x, y = pseudo_hex_to_oddr(*map(int, cstr.split('_')))
            else:
                x, y = (int(np.rint(float(v))) for v in cstr.split('_'))
            annots[y, x] = int(np.argmax(adat[cstr].values)) + 1                # 0 is reserved for background
        counts_grid = torch.from_numpy(counts).permute(2, 0, 1).float()         # channels first (count_datasets.py:293)
        annots_grid = torch.from_numpy(annots)
        counts_grid = counts_grid * (annots_grid > 0).float().unsqueeze(0)      # only annotated spots are filled
        patch_grid = None
        rxp = re.compile(r"(\d+)_(\d+)\.jpg$")
        for fname in sorted(os.listdir(str(self.img_files[idx]))):
            hit = rxp.match(fname)
            if hit is None:
                continue
            x, y = int(hit.group(1)), int(hit.group(2))
            patch = self.preprocess(Image.open(os.path.join(self.img_files[idx], fname)))
            if patch_grid is None:
                patch_grid = torch.zeros((self.h_st, self.w_st) + tuple(patch.shape))
            if self.Visium:
                x, y = pseudo_hex_to_oddr(x, y)
            patch_grid[y, x] = patch
        if patch_grid is None:
            raise FileNotFoundError("no '<x>_<y>.jpg' spot images in %s" % self.img_files[idx])
        # foreground = image data AND annotation (:237-244), without the reference's python double loop
        has_img = patch_grid.flatten(2).max(dim=2).values != 0
        annots_grid = torch.where(has_img, annots_grid, torch.zeros_like(annots_grid))
        counts_grid = counts_grid * has_img.unsqueeze(0).float()
        patch_grid = patch_grid * (annots_grid != 0).float().view(self.h_st, self.w_st, *([1] * (patch_grid.dim() - 2)))
        if self.training_pairs:
            return (patch_grid.float(), counts_grid), annots_grid
        return counts_grid, patch_grid.float(), annots_grid
