"""Count-data datasets feeding the f∘g path (API surface of /root/reference/gridnext/count_datasets.py).

`CountDataset` (:77-213) yields (count vector (G,) float32, label int64) per annotated spot;
`CountGridDataset` (:215-303) yields (counts (G, H_ST, W_ST) float32, labels (H_ST, W_ST) int64, 0 = background)
per array - the tensors `train_spotwise` / `train_gridwise` consume.  Constructor signatures, the `.classes`
attribute, ValueErrors and the item contract follow the reference; the file handling is new code:
every count file is parsed ONCE and cached (the reference re-reads a TSV with pandas for every single spot,
count_datasets.py:185-186).

File formats (as the reference's readers define them, utils.py:88-166, :220-244):
  count file      : delimited text, genes x spots; header = spot coordinate strings "x_y" (Visium: pseudo-hex
                    array_col_array_row), first column = gene names; optionally gzip
  Loupe annotation: CSV  Barcode,<annotation>   (+ Spaceranger tissue_positions[_list].csv mapping barcodes to
                    array_row/array_col; v1 files have no header)
  Splotch annotation: delimited one-hot matrix, annotations x spot coordinate strings
Known reference defect not reproduced: without annotation files CountDataset shifts every column index by one
(count_datasets.py:175) and fails on the last spot; here un-annotated spots index their own column.
"""
import gzip
import re

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset, TensorDataset

from .utils import pseudo_hex_to_oddr

_COORD = re.compile(r'\d+_\d+')


def read_positions(position_file):
    """Spaceranger barcode -> array/pixel coordinates table (v1: headerless, v2: header line starting 'barcode')."""
    with open(position_file, 'r') as fh:
        first = fh.readline()
    if first.startswith('barcode'):
        return pd.read_csv(position_file, index_col=0, header=0)
    return pd.read_csv(position_file, index_col=0, header=None,
                       names=["in_tissue", "array_row", "array_col", "pxl_row_in_fullres", "pxl_col_in_fullres"])


def read_annotfile(afile, position_file=None, afile_delim=',', Visium=True):
    """(coordinate strings, annotations): Loupe CSV + position file -> names; Splotch one-hot matrix -> class ids."""
    table = pd.read_csv(afile, header=0, index_col=0, sep=afile_delim)
    if Visium:
        table = table[table.iloc[:, 0] != '']
        table = table.join(read_positions(position_file), how='left').dropna()
        coords = ['%d_%d' % (c, r) for c, r in zip(table['array_col'], table['array_row'])]
        return coords, table.iloc[:, 0].values
    table = table[table.sum(axis=1) == 1]          # as the reference: filter on row sums (utils.py:238)
    return table.columns, np.argmax(table.values, axis=0)


def _read_counts(count_file, delim):
    return pd.read_csv(count_file, header=0, index_col=0, sep=delim)


def _label_names(annot_files, position_files, afile_delim):
    names = np.array([])
    for afile, pfile in zip(annot_files, position_files):
        _, strs = read_annotfile(afile, position_file=pfile, Visium=True, afile_delim=afile_delim)
        names = np.union1d(names, strs)
    return names                    # sorted unique = sklearn LabelEncoder().fit(...).classes_


def _check_files(data_files, annot_files, position_files, Visium, what):
    if annot_files is not None and not len(data_files) == len(annot_files):
        raise ValueError('Length of %s and annot_files must match.' % what)
    if Visium and annot_files is not None:
        if position_files is None:
            raise ValueError('Must provide Spaceranger position files mapping barcodes to array locations.')
        if len(position_files) != len(annot_files):
            raise ValueError('Number of Spaceranger position files does not match number of annotation files.')


def read_annotated_starray(count_file, annot_file=None, select_genes=None, h_st=78, w_st=64, Visium=True,
                           position_file=None, cfile_delim='\t', afile_delim='\t'):
    """(counts_grid (h_st, w_st, genes) float64, annots_grid (h_st, w_st)) of one array (reference utils.py:88-166):
    Visium spots are placed at their odd-right position; only annotated spots are filled when annotations are given;
    annots_grid holds names ('' = background) for Loupe files, class id + 1 (0 = background) for Splotch files."""
    cmat = _read_counts(count_file, cfile_delim)
    if select_genes is not None:
        cmat = cmat.loc[select_genes, :]
    lookup, grid_is_names = None, False
    if annot_file is not None:
        if position_file is not None:
            coords, names = read_annotfile(annot_file, position_file=position_file, Visium=True)
            annots = np.empty((h_st, w_st), dtype='U%d' % max(len(a) for a in names))
            lookup, grid_is_names = dict(zip(coords, names)), True
        else:
            coords, ids = read_annotfile(annot_file, Visium=False, afile_delim=afile_delim)
            annots = np.zeros((h_st, w_st), dtype=int)
            lookup = dict(zip(coords, ids))
    else:
        annots = np.zeros((h_st, w_st), dtype=int)
    counts = np.zeros((h_st, w_st, cmat.shape[0]), dtype=float)
    values = cmat.values
    for j, cstr in enumerate(cmat.columns):
        if Visium:
            xv, yv = map(int, cstr.split('_'))
            x, y = pseudo_hex_to_oddr(xv, yv)
        else:
            xc, yc = map(float, cstr.split('_'))
            x, y = int(np.rint(xc)), int(np.rint(yc))
        if lookup is None:
            counts[y, x] = values[:, j]
        elif cstr in lookup:
            counts[y, x] = values[:, j]
            annots[y, x] = lookup[cstr] if grid_is_names else lookup[cstr] + 1
    return counts, annots


class CountDataset(Dataset):
    """Independent classification of spots from 1-D expression vectors."""

    def __init__(self, count_files, annot_files=None, position_files=None, Visium=True,
                 select_genes=None, cfile_delim='\t', afile_delim=',', verbose=False):
        super().__init__()
        _check_files(count_files, annot_files, position_files, Visium, 'count_files')
        names = None
        if Visium and annot_files is not None:
            names = _label_names(annot_files, position_files, afile_delim)
            self.classes = names
        self.cfile_delim, self.afile_delim = cfile_delim, afile_delim
        self.select_genes = select_genes
        self.countfile_mapping, self.annotations, self.cind_mapping = [], [], []
        self._cache = {}
        missing = 0
        for i, cf in enumerate(count_files):
            opener = gzip.open if str(cf).endswith('gz') else open
            with opener(cf, 'rt') as fh:
                header = next(fh).strip('\n').split(cfile_delim)
            if annot_files is not None:
                if Visium:
                    coords, strs = read_annotfile(annot_files[i], position_file=position_files[i])
                    labels = np.searchsorted(names, strs)
                else:
                    coords, labels = read_annotfile(annot_files[i], Visium=False, afile_delim=afile_delim)
                lookup = dict(zip(coords, labels))
                for col, cstr in enumerate(header):
                    if cstr not in lookup:
                        if verbose:
                            print(annot_files[i], cstr, 'missing annotation')
                        missing += 1
                        continue
                    self.annotations.append(lookup[cstr])
                    self.countfile_mapping.append(cf)
                    self.cind_mapping.append(col)
            else:
                for col, cstr in enumerate(header):
                    if _COORD.match(cstr) is not None:
                        self.countfile_mapping.append(cf)
                        self.cind_mapping.append(col)
        if annot_files is not None:
            print('%d un-annotated spots' % (missing))

    def __len__(self):
        return len(self.cind_mapping)

    def _matrix(self, cf):
        if cf not in self._cache:
            self._cache[cf] = _read_counts(cf, self.cfile_delim)
        return self._cache[cf]

    def __getitem__(self, idx):
        mat = self._matrix(self.countfile_mapping[idx])
        col = mat.iloc[:, self.cind_mapping[idx] - 1]            # header column 0 is the gene-name column
        if self.select_genes is not None:
            wanted = set(self.select_genes)
            col = col[[g in wanted for g in mat.index]]          # file order, as the reference's line scan yields
        vec = torch.from_numpy(np.asarray(col.values, dtype=np.float32))
        label = torch.tensor(self.annotations[idx] if len(self.annotations) > 0 else 0).long()
        return vec.float(), label


class CountGridDataset(Dataset):
    """Registration of entire ST arrays from 3-D expression maps."""

    def __init__(self, count_files, annot_files=None, position_files=None, Visium=True,
                 select_genes=None, h_st=78, w_st=64, cfile_delim='\t', afile_delim='\t'):
        super().__init__()
        _check_files(count_files, annot_files, position_files, Visium, 'count_files')
        if Visium and annot_files is not None:
            self.classes = _label_names(annot_files, position_files, ',')
        self.count_files, self.annot_files, self.position_files = count_files, annot_files, position_files
        self.select_genes = select_genes
        self.h_st, self.w_st, self.Visium = h_st, w_st, Visium
        self.cfile_delim, self.afile_delim = cfile_delim, afile_delim

    def __len__(self):
        return len(self.count_files)

    def __getitem__(self, idx):
        af = self.annot_files[idx] if self.annot_files is not None else None
        pf = self.position_files[idx] if self.position_files is not None else None
        counts, annots = read_annotated_starray(self.count_files[idx], af, select_genes=self.select_genes,
                                                h_st=self.h_st, w_st=self.w_st, Visium=self.Visium, position_file=pf,
                                                cfile_delim=self.cfile_delim, afile_delim=self.afile_delim)
        counts = torch.from_numpy(counts).permute(2, 0, 1)       # channels first
        if annots.dtype != int:                                   # Loupe names -> class id + 1, '' -> 0
            flat = annots.flatten()
            ids = np.zeros(flat.shape, dtype=int)
            named = flat != ''
            ids[named] = np.searchsorted(self.classes, flat[named]) + 1
            annots = ids.reshape(annots.shape)
        return counts.float(), torch.from_numpy(annots).long()


def load_count_grid_dataset(count_files, annot_files=None, select_genes=None, h_st=78, w_st=64, Visium=True):
    """All arrays in memory as one TensorDataset (reference :52-72); Splotch-format annotations."""
    xs, ys = [], []
    for i, cf in enumerate(count_files):
        af = annot_files[i] if annot_files is not None else None
        c, a = read_annotated_starray(cf, af, select_genes=select_genes, h_st=h_st, w_st=w_st, Visium=Visium)
        xs.append(c)
        ys.append(a)
    x = torch.tensor(np.array(xs)).permute(0, 3, 1, 2)
    return TensorDataset(x.float(), torch.tensor(np.array(ys)).long())
