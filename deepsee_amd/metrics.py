"""Evaluation metrics on the device: PSNR / SSIM / RMSE of generated against ground-truth images.

Mirror of the reference's ``MetricsEvaluator`` (evaluator/evaluation.py:15-158) for the three metrics that need no
pretrained network: ``collect_samples(fake, real, name)`` scores one batch, ``get_result()`` returns the same
``"psnr/mean" ... "n_samples"`` OrderedDict, ``write_details`` appends one CSV row per sample.  The reference loops over
the samples on the CPU (tensor2im -> numpy uint8 -> cv2.filter2D in float64); here one kernel pair
(``dsee_psnr_ssim``) scores the whole batch from the fp32 tensors where they are, and 3 doubles per image come back.
LPIPS, MS-SSIM and FID (pretrained AlexNet / Inception weights, downloads in the reference) are out of scope: their
columns are absent from ``columns`` and from ``get_result()``.
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import lib as L
from . import ops


def psnr_ssim_rmse(fake, real):
    """fake, real: images in [-1, 1]; native NHWC fp32 [N,H,W,>=3] device tensors, or the reference's NCHW [N,3,H,W]
    (any device).  Returns a float64 tensor [N, 3] = (psnr, ssim, rmse) per image on the CPU."""
    def native(t):
        if getattr(t, "dsee_layout", None) == "nhwc":
            return t.detach().contiguous()
        assert t.dim() == 4 and t.shape[1] == 3, "NCHW [N,3,H,W] or a tagged native NHWC tensor expected"
        return ops.to_nhwc(t.detach().float().cuda())
    f, r = native(fake), native(real)
    assert f.shape == r.shape, "fake and real differ in shape"
    n, h, w, cs = f.shape
    ws = torch.empty(L.lib().dsee_psnr_ssim_workspace(n, h, w) // 8, dtype=torch.float64, device=f.device)
    out = torch.empty(n, 3, dtype=torch.float64, device=f.device)
    L.call("psnr_ssim", f, r, n, h, w, cs, ws, ws.numel() * 8, out)
    return out.cpu()


class MetricsEvaluator:
    """Collects per-sample scores; optionally writes them to ``folder_out/metrics.csv`` (evaluation.py:15-158)."""
    columns = ["ID", "PSNR", "SSIM", "RMSE"]

    def __init__(self, write_details=False, folder_out=None, extra_columns=(), extra_columns_content=(), append=False):
        assert len(extra_columns) == len(extra_columns_content), "Extra columns and content need to be of the same size"
        self.clear()
        self.write_details = write_details
        self.writer = None
        if write_details:
            self.writer = MetricsWriter(folder_out, self.columns, extra_columns, extra_columns_content, append)

    def clear(self):
        self.psnr_buffer, self.ssim_buffer, self.rmse_buffer, self.n_samples = [], [], [], 0

    @staticmethod
    def _get_id_from_path(path):
        return os.path.basename(path)[:-4]

    def collect_samples(self, fake, real, name=None):
        assert fake.shape[0] == real.shape[0]
        scores = psnr_ssim_rmse(fake, real).numpy()
        for i in range(scores.shape[0]):
            psnr, ssim, rmse = (float(v) for v in scores[i])
            self.psnr_buffer.append(psnr)
            self.ssim_buffer.append(ssim)
            self.rmse_buffer.append(rmse)
            if self.write_details:
                self.writer.append_line([self._get_id_from_path(name[i]), psnr, ssim, rmse])
        self.n_samples += scores.shape[0]

    def get_result(self):
        return OrderedDict([("psnr/mean", np.mean(self.psnr_buffer)), ("ssim/mean", np.mean(self.ssim_buffer)),
                            ("rmse/mean", np.mean(self.rmse_buffer)), ("psnr/std", np.std(self.psnr_buffer)),
                            ("ssim/std", np.std(self.ssim_buffer)), ("rmse/std", np.std(self.rmse_buffer)),
                            ("n_samples", self.n_samples)])


class MetricsWriter:
    """``metrics.csv`` with a header row and optional constant extra columns (evaluation.py:161-200)."""

    def __init__(self, path, metrics, extra_columns=(), extra_columns_content=(), append=False):
        self.path_out = os.path.join(path, "metrics.csv")
        header = list(extra_columns) + list(metrics)
        self.extra_columns_content = list(extra_columns_content)
        new_file = not (append and os.path.exists(self.path_out))
        self.file = open(self.path_out, "a" if append else "w")
        if new_file:
            self.append_line(header, add_extra_content=False)

    def append_line(self, row, add_extra_content=True):
        content = (self.extra_columns_content if add_extra_content else []) + list(row)
        self.file.write(",".join(map(str, content)) + os.linesep)
        self.file.flush()

    def __del__(self):
        if getattr(self, "file", None):
            self.file.close()
