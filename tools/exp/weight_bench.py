"""dsee_wino43_weights: LDS-staged kernel (16-byte-aligned weights) against the per-lane gather kernel (same weights at a 4-byte
offset), 512 x 512 x 3 x 3, the three forms, two-term fp16 and packed one-term U."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepsee_amd import lib as L
co = ci = 512
w = torch.randn(co, ci, 3, 3, device="cuda")
buf = torch.zeros(w.numel() + 4, device="cuda"); sh = buf[1:1 + w.numel()]; sh.copy_(w.view(-1))
amax = torch.zeros(2048, device="cuda"); amax[:] = float(w.abs().max())
rows, kp = L.wrows(co), L.kpad(1, 1, ci)
u = torch.zeros(36 * rows * kp * 2, dtype=torch.int16, device="cuda")
for split in (2, 4, 0):
    for flip in (0, 1, 2):
        line = []
        for name, src in (("staged", w), ("gather", sh)):
            for _ in range(3): L.call("wino43_weights", src, u, co, ci, flip, split, amax)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50): L.call("wino43_weights", src, u, co, ci, flip, split, amax)
            e.record(); torch.cuda.synchronize()
            line.append("%s %.1f us" % (name, s.elapsed_time(e) * 1e3 / 50))
        print("split %d flip %d: %s" % (split, flip, ", ".join(line)))
