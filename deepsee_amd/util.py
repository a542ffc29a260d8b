"""Host-side helpers of the inference / demo path."""
import os

import numpy as np
import torch


def save_style_matrix(tensor, path, create_dir=False):
    """Write one image's style matrix (rows = semantic regions, columns = style features) as a CSV file next to the result
    image.  File format of the reference's helper (util/util.py:150-158, called from demo.py:72): one matrix row per
    line, ',' separated, every value printed as '%.18e', so files written by either side load on the other."""
    m = tensor.detach().to("cpu", torch.float64).numpy()
    if m.ndim != 2:
        raise AssertionError("a style matrix is [label_nc, style_size]; got shape %s" % (tuple(m.shape),))
    if not str(path).endswith(".csv"):
        raise AssertionError("style matrices are stored as .csv: %s" % path)
    folder = os.path.dirname(str(path))
    if create_dir and folder:
        os.makedirs(folder, exist_ok=True)
    with open(path, "w") as fh:
        for row in m:
            fh.write(",".join("%.18e" % float(v) for v in row) + "\n")


def load_style_matrix(path, device="cuda"):
    """Inverse of save_style_matrix: [label_nc, regional_style_size] fp32, ready to be stacked into the
    `encoded_style` [N, label_nc, S] input of SRModel.forward(mode='demo')."""
    return torch.from_numpy(np.loadtxt(path, delimiter=",", dtype=np.float32, ndmin=2)).to(device)
