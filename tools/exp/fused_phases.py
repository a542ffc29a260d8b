"""Cycle stamps of the fused SPADE kernel (DSEE_LIB=tools/exp/libfabl_32.so): per wave, stage waits | stage bodies |
main loop | epilogue."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepsee_amd import lib as L
import fused_kernel_bench as B
st = torch.zeros(64 * 8 * 8, device="cuda")
L.lib().dsee_fused_set_stamps.argtypes = [ctypes.c_void_p]
L.lib().dsee_fused_set_stamps(ctypes.c_void_p(st.data_ptr()))
for (n, h, c, ld, pi) in [(8, 256, 512, 160, True), (8, 256, 512, 128, False)]:
    ms = B.bench(n, h, c, ld, pi, reps=2)
    s = st.view(64, 8, 8).cpu()
    print("N=%d %d^2 C=%d K=%d: %.3f ms; per wave cycles (mean over 64 blocks): waits %.0f  mfma part %.0f  loads part %.0f  epilogue %.0f" % (
        n, h, c, ld, ms, s[..., 0].mean(), s[..., 1].mean(), s[..., 2].mean(), s[..., 3].mean()))
    print("   of the loads part: fragment reads %.0f  DMA requests %.0f  fold / Y update %.0f ; of the waits: own vmcnt %.0f (rest: barrier)" % (
        s[..., 4].mean(), s[..., 5].mean(), s[..., 6].mean(), s[..., 7].mean()))
    print("   block 0 waves: waits", [int(v) for v in s[0, :, 0]], "mfma", [int(v) for v in s[0, :, 1]], "loads", [int(v) for v in s[0, :, 2]], "epi", [int(v) for v in s[0, :, 3]])
