"""Device input pipeline: replaces the reference's data/base_dataset.py:87-116,171-201 (PIL ->
torchvision transforms -> float CPU tensors -> .cuda()) + data/preprocessor.py on the hot path.

What travels over PCIe is what the files hold: a uint8 label map [H,W] and a uint8 RGB image [H,W,3] per sample
(256 KB at 256x256 instead of 1 MB of fp32 + a 5 MB one-hot map built on the device by the reference).  ToTensor,
Normalize((.5,.5,.5),(.5,.5,.5)), the per-sample horizontal flip, the 255 -> label_nc 'unknown' remap and the bicubic
LR image are HIP kernels (dsee_image_u8_to_nhwc, dsee_label_u8_prepare, dsee_bicubic_down) that write the NHWC RGB0
fp32 / uint8-label layout the networks consume.  Host side: PIL decoding + resize + crop only (get_params /
get_transform semantics of base_dataset.py:171-201, 'resize_and_crop' mode), batches collated into pinned memory and
uploaded asynchronously one batch ahead of the compute stream.

Datasets yield dicts  {'label': uint8 [H,W], 'image': uint8 [H,W,3], 'flip': 0/1, 'path': str}  (+ 'guiding_label',
'guiding_image' for the guided variant); DeviceLoader yields the native batch dict TrainerManager.run_*_one_step
accept: {'input_semantics': ops.Labels, 'image_hr', 'image_lr'[, 'guiding_label', 'guiding_image'], 'path'}.
"""
import os
import random

import numpy as np
import torch

from . import lib as L
from . import ops

IMG_EXT = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".tiff", ".webp")


class SyntheticDataset:
    """Blocky 19-class label maps (16x16 cells, nearest-upsampled: piecewise constant like real masks) + uniform
    random images, as uint8 -- the benchmark's input distribution in the loader's wire format."""

    def __init__(self, opt, length=64, seed=1234, guided=None):
        self.opt, self.length, self.seed = opt, int(length), int(seed)
        self.guided = bool(opt.guiding_style_image) if guided is None else guided

    def __len__(self):
        return self.length

    def _pair(self, rng):
        h = self.opt.crop_size
        cells = rng.integers(0, self.opt.label_nc, size=(16, 16), dtype=np.uint8)
        label = np.repeat(np.repeat(cells, h // 16, 0), h // 16, 1)
        image = rng.integers(0, 256, size=(h, h, 3), dtype=np.uint8)
        return label, image

    def __getitem__(self, i):
        rng = np.random.default_rng(self.seed * 1000003 + i)
        label, image = self._pair(rng)
        out = {"label": label, "image": image, "flip": int(rng.integers(0, 2)) if self.opt.isTrain else 0,
               "path": "synthetic/%06d" % i}
        if self.guided:
            out["guiding_label"], out["guiding_image"] = self._pair(rng)
        return out


class FolderDataset:
    """label_dir/<id>.png (uint8 class indices, 255 = unknown) + image_dir/<id>.jpg pairs, matched by file stem like
    base_dataset.py:44-62 (paths_match).  'resize_and_crop': both are resized to load_size (NEAREST for the label,
    BICUBIC for the image, base_dataset.py:92,107), cropped to crop_size at one random position and flipped with
    p = 0.5 in training (the flip itself happens on the device)."""

    def __init__(self, opt, label_dir, image_dir, seed=0, no_flip=None):
        from PIL import Image  # noqa: F401  (fail here, not in a worker)
        mode = getattr(opt, "preprocess_mode", "resize_and_crop")
        if mode != "resize_and_crop":     # (base_dataset.py:76-104 also knows scale_width*, fixed, none: not restated here)
            raise ValueError("FolderDataset implements preprocess_mode='resize_and_crop' only, got %r" % (mode,))
        self.opt = opt
        self.no_flip = bool(getattr(opt, "no_flip", False)) if no_flip is None else bool(no_flip)
        self.rng = random.Random(seed)
        labels = {os.path.splitext(f)[0]: os.path.join(label_dir, f) for f in sorted(os.listdir(label_dir))
                  if f.lower().endswith(IMG_EXT)}
        images = {os.path.splitext(f)[0]: os.path.join(image_dir, f) for f in sorted(os.listdir(image_dir))
                  if f.lower().endswith(IMG_EXT)}
        missing = sorted(set(labels) ^ set(images))
        assert not missing, "label/image files without a partner (first: %s)" % missing[:3]
        self.items = [(labels[k], images[k]) for k in sorted(labels)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        from PIL import Image
        opt = self.opt
        lp, ip = self.items[i]
        x = self.rng.randint(0, max(0, opt.load_size - opt.crop_size))
        y = self.rng.randint(0, max(0, opt.load_size - opt.crop_size))
        flip = int(self.rng.random() > 0.5) if (opt.isTrain and not self.no_flip) else 0
        box = (x, y, x + opt.crop_size, y + opt.crop_size)
        lab = Image.open(lp)
        if lab.mode not in ("L", "P"):
            raise ValueError("%s: label maps are single-channel class-index images, got mode %r" % (lp, lab.mode))
        lab = lab.resize((opt.load_size, opt.load_size), Image.NEAREST).crop(box)
        img = Image.open(ip).convert("RGB").resize((opt.load_size, opt.load_size), Image.BICUBIC).crop(box)
        return {"label": np.asarray(lab, dtype=np.uint8), "image": np.asarray(img, dtype=np.uint8), "flip": flip,
                "path": ip}


def device_preprocess(opt, batch, stream=None):
    """uint8 host/device batch -> native batch dict on the device (all kernels on the current stream).
    batch: 'label' uint8 [N,H,W], 'image' uint8 [N,H,W,3], optional 'flip' uint8 [N], 'guiding_*', 'path'."""
    def dev(t):
        return t if t.is_cuda else t.cuda(non_blocking=True)

    flip = dev(batch["flip"]) if batch.get("flip") is not None else None

    def image(u8):
        u8 = dev(u8).contiguous()
        n, h, w, _ = u8.shape
        out = ops.new(n, h, w, 4)
        L.call("image_u8_to_nhwc", u8, flip, out, n, h, w, 4)
        out.dsee_layout = "nhwc"
        return out

    def labels(u8):
        u8 = dev(u8).contiguous()
        n, h, w = u8.shape
        out = torch.empty_like(u8)
        L.call("label_u8_prepare", u8, flip, out, n, h, w, opt.label_nc)
        return ops.Labels(out, opt.label_nc)

    hr = image(batch["image"])
    res = {"input_semantics": labels(batch["label"]), "image_hr": hr, "image_lr": ops.bicubic_down(hr, opt.start_size)}
    if "guiding_image" in batch:
        res["guiding_image"] = image(batch["guiding_image"])
        res["guiding_label"] = labels(batch["guiding_label"])
    if "path" in batch:
        res["path"] = batch["path"]
    return res


class DeviceLoader:
    """Batches a dataset into pinned uint8 tensors and runs upload + device_preprocess for batch k+1 on a side stream
    while the compute stream works on batch k.  `shard` = (rank, world) gives every data-parallel rank a disjoint,
    equally long slice of every epoch (the reference's DataLoader feeds one process and DataParallel scatters)."""

    def __init__(self, dataset, opt, batch_size=None, shuffle=True, seed=0, shard=(0, 1), drop_last=True):
        self.ds, self.opt = dataset, opt
        self.bs = int(batch_size or opt.batchSize)
        self.shuffle, self.seed, self.epoch = shuffle, seed, 0
        self.rank, self.world = shard
        self.drop_last = drop_last
        self.side = torch.cuda.Stream() if torch.cuda.is_available() else None

    def __len__(self):
        per_rank = len(self.ds) // self.world
        return per_rank // self.bs if self.drop_last else (per_rank + self.bs - 1) // self.bs

    def indices(self):
        idx = list(range(len(self.ds)))
        if self.shuffle:
            random.Random(self.seed + self.epoch).shuffle(idx)     # same permutation on every rank
        per_rank = len(idx) // self.world
        return idx[self.rank * per_rank:(self.rank + 1) * per_rank]

    def collate(self, samples):
        pin = torch.cuda.is_available()

        def stack(key):
            t = torch.from_numpy(np.stack([s[key] for s in samples]))
            return t.pin_memory() if pin else t

        out = {"label": stack("label"), "image": stack("image"),
               "flip": torch.tensor([s.get("flip", 0) for s in samples], dtype=torch.uint8),
               "path": [s.get("path", "") for s in samples]}
        if pin:
            out["flip"] = out["flip"].pin_memory()
        if "guiding_image" in samples[0]:
            out["guiding_image"], out["guiding_label"] = stack("guiding_image"), stack("guiding_label")
        return out

    def _stage(self, ids):
        host = self.collate([self.ds[i] for i in ids])
        with torch.cuda.stream(self.side):
            dev = device_preprocess(self.opt, host)
            ev = torch.cuda.Event()
            ev.record(self.side)
        return dev, ev, host       # `host` is kept alive until the copies that read it have run

    def __iter__(self):
        idx = self.indices()
        self.epoch += 1
        batches = [idx[i:i + self.bs] for i in range(0, len(idx), self.bs)]
        if self.drop_last:
            batches = [b for b in batches if len(b) == self.bs]
        nxt = self._stage(batches[0]) if batches else None
        for k in range(len(batches)):
            dev, ev, host = nxt
            nxt = self._stage(batches[k + 1]) if k + 1 < len(batches) else None
            torch.cuda.current_stream().wait_event(ev)
            for v in dev.values():                       # tensors made on the side stream, used on the compute stream
                t = v.t if isinstance(v, ops.Labels) else v
                if isinstance(t, torch.Tensor):
                    t.record_stream(torch.cuda.current_stream())
            yield dev
