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
