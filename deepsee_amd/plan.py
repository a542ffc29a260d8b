"""KernelPlan: which of the equivalent kernel paths a model's operators take.

Rounds 1-3 steered `deepsee_amd/ops.py` with ~18 mutable module attributes (and a process-wide `ops.HALF` for the precision),
so two models of different precision could not coexist in a process and benchmarks had to mutate a module between
measurements.  A plan is an immutable value owned by ONE model (`SRModel.plan`, built from `opt.precision` /
`opt.kernel_plan`); `SRModel.forward` makes it the active plan of the calling thread for the duration of the forward pass,
every autograd node records the plan it was built under (`ctx.plan`) and re-activates it in its backward -- which runs on the
autograd engine's thread -- so a forward and its backward always agree, whatever other models do in between.  Code that calls
operators directly (unit tests, micro-benchmarks) either runs under the default plan or inside `with plan.active():`.

Every field selects between kernel paths that compute the SAME function (the non-default side of each is exercised by
tests/test_gpu_model.py::test_kernel_path_switches); `half` alone changes the arithmetic (BASELINE configs[2]'s 16-bit mode).
"""
import contextlib
import dataclasses
import threading
from typing import Any, Optional


@dataclasses.dataclass(frozen=True)
class KernelPlan:
    # Winograd F(4x4,3x3) for the wide 3x3 / stride-1 layers (4x fewer fp32 MACs than the direct form): forward + data
    # gradient | weight gradient | the SPADE/SEAN gamma/beta convolution
    winograd: bool = True
    winograd_wgrad: bool = True
    winograd_mod: bool = True
    # fp32 GEMMs of the Winograd domain on the 16-bit matrix cores via operand splitting (gemm_bf16x3.hip); False keeps them
    # on v_mfma_f32_32x32x2_f32 (bench.py's f32_mfma_only comparison run)
    gemm_split: bool = True
    # A operand of the forward / data-gradient GEMMs kept in fp32 in HBM and split inside the GEMM kernel; False uses
    # pre-split bf16x3 A operands (round 1's form)
    gemm_af32: bool = True
    # keep the forward's V for the weight gradient (False: transform x again in the backward pass)
    keep_v: bool = True
    # two-term fp16 splits (3 instead of 6 MFMA products per fp32 multiply-add, operands scaled by exact powers of two);
    # False keeps the exact 3-term bf16 form (bench.py's bf16x3_exact comparison run)
    gemm_f16x2: bool = True
    # Half-precision compute mode (BASELINE configs[2]'s 16-bit arithmetic; opt.precision = "fp16"): the Winograd-domain
    # GEMMs take operands scaled by powers of two and rounded to ONE fp16 term (one MFMA product, fp32 accumulate), stored
    # as such in HBM, and write their products M / dV as scaled fp16; activations, statistics, master weights and the
    # optimizer stay fp32.  fp16 and not bf16 because the F(4x4,3x3) transforms amplify operand rounding ~10x (per-layer
    # error 2.6 % with bf16 operands, 0.33 % with scaled fp16; a direct bf16 convolution: 0.24 %).
    half: bool = False
    # ... and in that mode the gamma/beta convolution of the SPADE/SEAN norms (K = 128 / 160 -> 1024 rows) as well; False keeps
    # the norms on the two-term fp16x2 kernels of the fp32 path (fused forward, pre-split gradient) while the 512 -> 512
    # convolutions run one-term
    half_norms: bool = True
    # A operand of a forward Winograd GEMM written pre-split by the input transform (False: fp32 V, split inside the GEMM)
    presplit_a: bool = True
    # the pre-split NT GEMMs (forward / adjoint data gradient of the wide convolutions) on the one-wave-per-SIMD kernel of
    # csrc/gemm_w4.hip (128 x 128 wave tiles, one barrier per slab; round 6).  False: the 8-wave ping-pong kernel of rounds 2-5
    gemm_w4: bool = True
    # data gradient of the 4x4 / stride-2 convolutions by output parity: one dense 2x2 convolution with 4 Cin output columns + a
    # depth-to-space copy (False: the general strided-gather kernel, three quarters of whose MFMAs multiply zeros)
    dgrad_s2_parity: bool = True
    presplit_min_tiles: int = 512      # fewest 256 x 256 tiles for which a layer takes the pre-split NT GEMM (below: 128 x 128 tiles, fp32 A)
    conv_halo_f16: bool = True         # 3 x 3 / stride 1 direct layers on the split-operand halo kernel (patch converted once per chunk, not per tap)
    onehot_wgrad_mfma: bool = True     # SPADE-only norms: mlp_shared's weight gradient on the MFMA kernel over materialised one-hot channels
    conv_amax_out: bool = True         # direct layers write max |out| in their epilogue (operand bound of the next direct layer)
    # A dY A^T of a convolution written pre-split for its weight gradient and adjoint data gradient (False: fp32 dM)
    presplit_dm: bool = True
    # ... and the gamma/beta gradient of a SPADE/SEAN norm as well (False: fp32 dM from the norm backward's reduce pass)
    presplit_gb: bool = True
    # bias and noise-weight gradients of a Winograd layer inside its A dY A^T pass (False: separate channel_dot passes)
    dout_sums: bool = True
    # data gradient in the adjoint form from the dM = A dY A^T the weight gradient needs anyway (False: transform dy a
    # second time with B^T . B and run the rotated-kernel convolution)
    adjoint_dgrad: bool = True
    # NoiseInjection draws regenerated inside the consumer's output transform (False: stand-alone UpNoise passes)
    fuse_noise: bool = True
    # the generator's to-RGB layer (512 -> 3 channels at full resolution) as a 1x1 GEMM with 27 outputs + a 9-point gather
    # (False: the VALU / cross-lane-reduction kernels of thin.hip)
    thin_gemm: bool = True
    # one statistics pass per tensor, shared by the BatchNorms that normalise it (False: one pass per norm layer)
    share_stats: bool = True
    # BatchNorm statistics rows written by the kernel that produces the norm's input (False: a statistics pass over x)
    producer_stats: bool = True
    # gamma/beta gradient written by the norm backward directly in the Winograd domain (False: dgb + wino43_dout)
    fuse_dm: bool = True
    # the fused SPADE / SEAN forward (dsee_spade_fused_fwd; False: GEMM + wino43_output_modulate)
    fused_norm: bool = True
    # ... on the one-wave-per-SIMD kernel (csrc/spade_fused_w4.hip, round 6: 4 waves x four 16x16 blocks, every MFMA followed by its
    # share of the fragment reads / LDS-DMA requests / fold work in program order, piece-granular ring with 2 NP - 1 pieces in
    # flight).  Bit-identical to the 8-wave kernel of rounds 3-5 and measured at the SAME speed stand-alone (2.22 vs 2.23 ms at
    # N = 8, 256^2, K = 160; 1.83 vs 1.87 at K = 128) and in the step (88.1 ms either way): with a different issue structure AND a
    # three times deeper request window the time does not move, removing the requests takes 0.8 ms off both -- the kernel is bound
    # by what the L2 -> LDS path delivers to a 64 x 64 block, not by how the block issues it (profiles/r06_fused_w4*.txt).  Off by
    # default (the 8-wave kernel also serves the 16-bit storage mode); kept under the same tests.
    fused_w4: bool = False
    # the fused forward also writes the LeakyReLU branch of h as a bit mask ([pixel][C/32] words) and the two passes of the norm
    # backward read it instead of h (False: they read h, 32x the bytes, for its sign)
    sign_mask: bool = True
    # the backward of the LeakyReLU in front of the to-RGB layer inside that layer's data-gradient kernel (False: the last
    # resblock's own pass over (dy, y))
    defer_act: bool = True
    # independent branches of the step (the two discriminator scales, VGG on the generated / the real image) on side streams --
    # parallel branches of the captured hipGraph (False: everything on one stream, in program order).  Built in round 5 and
    # worth ~1-2 ms of the 88 ms step, but OFF by default: replaying a graph WITH parallel branches segfaults inside the HIP
    # runtime (hip::Graph::UpdateStreams <- hip::GraphExec::Run <- hipGraphLaunch, ROCm 7.0.2) late in a long-lived process
    # -- deterministically at the first replay of the 34th test of tests/test_gpu_model.py, never in any subset of it, nor in
    # 40 managers x capture / replay cycles (tools/exp/graph_churn.py), also with the captured hipGraph_t kept alive
    # (CUDAGraph(keep_graph=True)); rocgdb backtrace in profiles/r05_graph_branch_segv.txt.
    # Single-stream graphs (rounds 3-4) never did.
    branch_streams: bool = False
    # direct (non-Winograd) convolutions with at least this much work run their MFMAs on fp16x2-split operands; 0 disables
    conv_f16x2_min_flop: float = 1e9
    # SyncBN-over-RCCL (ops.SyncBNConfig) or None for north_star's sync-free BatchNorm
    sync_bn: Optional[Any] = None

    def replace(self, **changes):
        return dataclasses.replace(self, **changes)

    @contextlib.contextmanager
    def active(self):
        """Make this plan the one the operators of the calling thread consult."""
        prev = getattr(_tls, "plan", None)
        _tls.plan = self
        try:
            yield self
        finally:
            _tls.plan = prev

    @classmethod
    def fields(cls):
        return [f.name for f in dataclasses.fields(cls)]


def _env_overrides():
    """DSEE_PLAN="field=value,field=value": overrides of the DEFAULT plan for a whole process (A/B runs of the test suite or of a
    training script without touching its code); values are Python literals."""
    import ast
    import os
    text = os.environ.get("DSEE_PLAN", "").strip()
    if not text:
        return {}
    out = {}
    for item in text.split(","):
        k, _, v = item.partition("=")
        k = k.strip()
        if k not in KernelPlan.fields():
            raise ValueError("DSEE_PLAN: unknown KernelPlan field %r (fields: %s)" % (k, ", ".join(KernelPlan.fields())))
        out[k] = ast.literal_eval(v.strip())
    return out


DEFAULT_PLAN = KernelPlan(**_env_overrides())
_tls = threading.local()


def current():
    """The active plan of this thread (DEFAULT_PLAN outside any `with plan.active()`)."""
    return getattr(_tls, "plan", None) or DEFAULT_PLAN


def from_opt(opt):
    """The plan of a model built from `opt`: opt.precision ('fp32' | 'fp16') and the optional opt.kernel_plan (a KernelPlan,
    or a dict of field overrides such as bench.py's --arith choices)."""
    prec = getattr(opt, "precision", "fp32")
    if prec not in ("fp32", "fp16"):
        raise ValueError("opt.precision must be 'fp32' or 'fp16', got %r" % (prec,))
    over = getattr(opt, "kernel_plan", None)
    if isinstance(over, KernelPlan):
        plan = over
    else:
        plan = DEFAULT_PLAN.replace(**dict(over or {}))
    return plan.replace(half=(prec == "fp16"))
