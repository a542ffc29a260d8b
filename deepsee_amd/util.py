"""Host-side helpers of the inference / demo path."""
import os

import numpy as np
import torch


def save_style_matrix(tensor, path, create_dir=False):
    """One image's style matrix [label_nc, regional_style_size] as CSV: the file format of util/util.py:150-158
    (numpy.savetxt, ',' delimiter, '%.18e'), what demo.py:72 writes next to each result image."""
    if create_dir:
        os.makedirs(os.path.dirname(path), exist_ok=True)
    assert len(tensor.shape) == 2, "Shape is incorrect: {}".format(tuple(tensor.shape))
    assert path.endswith(".csv")
    np.savetxt(path, np.array(tensor.detach().cpu()), delimiter=",")


def load_style_matrix(path, device="cuda"):
    """Inverse of save_style_matrix: [label_nc, regional_style_size] fp32, ready to be stacked into the
    `encoded_style` [N, label_nc, S] input of SRModel.forward(mode='demo')."""
    return torch.from_numpy(np.loadtxt(path, delimiter=",", dtype=np.float32, ndmin=2)).to(device)
