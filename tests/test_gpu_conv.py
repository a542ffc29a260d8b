"""HIP implicit-GEMM conv family vs torch CPU fp32 (F.conv2d and autograd) on the same inputs."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(x, cs=None):
    n, c, h, w = x.shape
    cs = cs or (c + 3) // 4 * 4
    out = torch.zeros(n, h, w, cs)
    out[..., :c] = x.permute(0, 2, 3, 1)
    return out.contiguous()


def nchw(x, c):
    return x[..., :c].permute(0, 3, 1, 2).contiguous()


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


CASES = [
    # n, cin, cout, h, k, stride, pad, ups, act, bias, res
    (2, 32, 128, 16, 3, 1, 1, 0, 0, True, False),
    (2, 128, 128, 16, 3, 1, 1, 0, 1, True, True),
    (1, 64, 192, 12, 3, 1, 1, 1, 0, True, False),     # fused x2 nearest upsample, ragged Cout
    (2, 3, 64, 20, 3, 1, 1, 0, 2, True, False),       # Cin=3 (stored 4), relu
    (2, 64, 3, 16, 3, 1, 1, 0, 3, True, False),       # Cout=3 (stored 4), tanh
    (2, 22, 32, 17, 4, 2, 2, 0, 1, True, False),      # D first layer: k4 s2 p2, odd size
    (2, 32, 64, 19, 4, 2, 2, 0, 0, False, False),
    (2, 128, 256, 9, 4, 1, 2, 0, 0, False, False),
    (1, 256, 1, 10, 4, 1, 2, 0, 0, True, False),      # Cout=1
    (2, 32, 64, 16, 3, 2, 1, 0, 1, False, False),     # encoder stride-2
    (1, 512, 512, 8, 3, 1, 1, 0, 0, True, True),
    (2, 32, 64, 40, 3, 1, 1, 0, 0, True, False),      # W >= 32: wgrad fast path (buffer loads, incremental rows)
    (1, 64, 160, 64, 3, 1, 1, 0, 1, False, False),    # W = 64, ragged Cout tile
    (3, 128, 128, 32, 3, 1, 1, 0, 0, True, True),
    (6, 128, 128, 64, 3, 1, 1, 0, 0, False, False),   # regression: 24 pixel splits; the dummy prefetch past the last
                                                      # slab once read beyond the end of `in` (fault at x_end + 16 KB)
                                                      # (f16x2, round 6: the split-operand halo kernel, two column blocks, 4 chunks)
    (32, 64, 64, 32, 3, 1, 1, 0, 2, True, False),     # VGG conv1_2's shape class: split-operand halo kernel (>= 256 workgroups)
    (8, 64, 128, 64, 3, 1, 1, 0, 1, True, True),      # ... two column blocks with bias, residual and LeakyReLU
    (8, 3, 64, 50, 3, 1, 1, 0, 2, True, False),       # VGG conv1_1's shape class (f16x2: the LDS-free K = 36 kernel, ragged last tile)
]


@pytest.mark.parametrize("f16x2", [False, True], ids=["f32mfma", "f16x2"])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case, f16x2):
    """Every geometry of the path through dsee_conv2d_fwd (forward and data gradient) and dsee_conv2d_wgrad against
    F.conv2d autograd; f16x2: the same through dsee_conv2d_fwd_f16x2 (operands split into two scaled fp16 terms inside
    the kernel, maxima from dsee_absmax) -- same 2e-5 bound."""
    from deepsee_amd import lib as L
    n, cin, cout, h, k, stride, pad, ups, act, use_bias, use_res = case

    def fwd(geom_, x_, w_, b_, r_, out_, act_, slope_):
        # through the *_amax forms (round 6): the epilogue's max |out| must be exactly the maximum of what it stored
        ao = torch.zeros(2048, device="cuda")
        if not f16x2:
            L.call("conv2d_fwd_amax", C.byref(geom_), x_, w_, b_, r_, 0, out_, act_, slope_, ao)
        else:
            ax, aw = torch.zeros(2048, device="cuda"), torch.zeros(2048, device="cuda")
            L.call("absmax", x_, x_.numel(), ax)
            L.call("absmax", w_, w_.numel(), aw)
            L.call("conv2d_fwd_f16x2_amax", C.byref(geom_), x_, w_, b_, r_, 0, out_, act_, slope_, ax, aw, ao, 0)
        torch.cuda.synchronize()
        assert float(ao.max()) == float(out_.abs().max()), (float(ao.max()), float(out_.abs().max()))
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(n, cin, h, h, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) if use_bias else None
    xr = x.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if ups else xr
    y = F.conv2d(xin, wr, b, stride=stride, padding=pad)
    res = torch.randn(y.shape, generator=g) if use_res else None
    if use_res:
        y = y + res
    ypre = y
    y = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu, 3: torch.tanh}[act](y)
    gy = torch.randn(ypre.shape, generator=g)
    ypre.backward(gy)  # gradient w.r.t. the pre-activation; act backward is a separate kernel

    cin_s, cout_s = L.pad4(cin), L.pad4(cout)
    dev = "cuda"
    geom = L.geom_fwd(n, h, h, cin_s, cout_s, k, stride, pad, ups)
    x_d = nhwc(x).to(dev)
    w_d = w.to(dev)
    wp = torch.empty(L.wrows(cout_s), L.kpad(k, k, cin_s), device=dev)
    L.call("pack_weight_fwd", w_d, None, None, wp, cout, cin, k, k, cin_s, geom.korder)
    b_d = None
    if use_bias:
        b_d = torch.zeros(cout_s, device=dev)
        b_d[:cout] = b.to(dev)
    r_d = nhwc(res).to(dev) if use_res else None
    out = torch.empty(n, geom.Ho, geom.Wo, cout_s, device=dev)
    fwd(geom, x_d, wp, b_d, r_d, out, act, 0.2)
    torch.cuda.synchronize()
    assert geom.Ho == y.shape[2]
    assert rel(nchw(out.cpu(), cout), y.detach()) < 2e-5
    if cout_s != cout:
        assert float(out[..., cout:].abs().max()) == 0.0

    # data gradient (w.r.t. the logical, possibly upsampled, input)
    gd = L.geom_dgrad(geom)
    wd = torch.empty(L.wrows(cin_s), L.kpad(k, k, cout_s), device=dev)
    L.call("pack_weight_dgrad", w_d, None, None, wd, cout, cin, k, k, cout_s, gd.korder)
    gy_d = nhwc(gy).to(dev)
    dx = torch.empty(n, gd.Ho, gd.Wo, cin_s, device=dev)
    fwd(gd, gy_d, wd, None, None, dx, 0, 0.0)
    torch.cuda.synchronize()
    dx_ref = xr.grad
    dx_c = nchw(dx.cpu(), cin)
    if ups:
        dx_c = F.avg_pool2d(dx_c, 2) * 4
    assert rel(dx_c, dx_ref) < 2e-5

    # weight gradient
    ws_bytes = L.lib().dsee_conv2d_wgrad_workspace(C.byref(geom))
    ws = torch.empty(ws_bytes // 4, device=dev)
    dw = torch.empty(cout, cin, k, k, device=dev)
    if f16x2:
        ax, ad = torch.zeros(2048, device=dev), torch.zeros(2048, device=dev)
        L.call("absmax", x_d, x_d.numel(), ax)
        L.call("absmax", gy_d, gy_d.numel(), ad)
        L.call("conv2d_wgrad_f16x2", C.byref(geom), x_d, gy_d, ws, C.c_size_t(ws_bytes), dw, cout, 0, cin, ax, ad)
    else:
        L.call("conv2d_wgrad", C.byref(geom), x_d, gy_d, ws, C.c_size_t(ws_bytes), dw, cout, 0, cin)
    torch.cuda.synchronize()
    assert rel(dw.cpu(), wr.grad) < 2e-5


@pytest.mark.parametrize("n,cin,cout,h,act,use_res", [(2, 128, 256, 32, 1, True), (8, 256, 128, 16, 0, False),
                                                      (1, 512, 512, 64, 0, True)])
def test_winograd_conv_autograd_function(n, cin, cout, h, act, use_res):
    """ops.conv2d routes wide 3x3 / stride-1 layers through Winograd F(4x4,3x3) (36 grouped MFMA GEMMs + two streaming
    transforms) for the forward, the data gradient and (split-K over tiles, dw = G^T dU G) the weight gradient.  fp32 F(4x4,3x3)
    carries ~10x the rounding error of the direct form (Lavin & Gray 2016, table 4): bound 1e-4 instead of 2e-5."""
    from deepsee_amd import ops
    assert ops._wino_ok(n, h, h, cin, cout, 3, 1, 1, 0)
    g = torch.Generator().manual_seed(n * cin + h)
    x = torch.randn(n, cin, h, h, generator=g).requires_grad_()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).requires_grad_()
    b = torch.randn(cout, generator=g).requires_grad_()
    res = torch.randn(n, cout, h, h, generator=g).requires_grad_() if use_res else None
    y = F.conv2d(x, w, b, padding=1)
    if use_res:
        y = y + res
    if act == 1:
        y = F.leaky_relu(y, 0.2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = nhwc(x.detach()).cuda().requires_grad_()
    wd, bd = w.detach().cuda().requires_grad_(), b.detach().cuda().requires_grad_()
    rd = nhwc(res.detach()).cuda().requires_grad_() if use_res else None
    yd = ops.conv2d(xd, wd, bd, rd, 1, 1, 0, act)
    yd.backward(nhwc(gy).cuda())
    torch.cuda.synchronize()
    assert rel(nchw(yd.detach().cpu(), cout), y.detach()) < 1e-4
    # with a fused LeakyReLU, outputs within the Winograd rounding error (~1e-5) of zero take the other slope in the
    # backward: a ~sqrt(fraction) = 1e-3 effect on every gradient (same mechanism as in test_gpu_model.py)
    gt = 5e-3 if act else 1e-4
    assert rel(nchw(xd.grad.cpu(), cin), x.grad) < gt
    assert rel(wd.grad.cpu(), w.grad) < gt
    assert rel(bd.grad.cpu(), b.grad) < (gt if act else 2e-5)
    if use_res:
        assert rel(nchw(rd.grad.cpu(), cout), res.grad) < (gt if act else 2e-5)


def _split_rows(x):
    """fp32 [rows][K] -> slab-major bf16x3 [K/16][rows][3][16] (int16 view), the layout dsee_gemm_bf16x3 reads."""
    x0 = x.bfloat16()
    r = x - x0.float()
    x1 = r.bfloat16()
    x2 = (r - x1.float()).bfloat16()
    assert torch.equal(x0.float() + x1.float() + x2.float(), x)       # the split is exact
    rows, k = x.shape
    return torch.stack([x0, x1, x2], 0).view(3, rows, k // 16, 16).permute(2, 1, 0, 3).contiguous().view(torch.int16)


@pytest.mark.parametrize("groups,tg,n,k,tile", [(3, 256, 256, 160, 1), (2, 512, 256, 512, 2), (36, 128, 128, 32, 1),
                                                (1, 1024, 512, 1024, 0)])
def test_gemm_bf16x3_is_fp32_accurate(groups, tg, n, k, tile):
    """The split-operand GEMM on the bf16 matrix cores against float64: its error must be that of fp32 arithmetic --
    the six-product scheme drops only terms below 2^-26 |a||b| and every kept product is exact, so what remains is
    the rounding of the fp32 accumulator (one chain of 6*K/16 MFMA accumulations; a blocked CPU sgemm on the same
    data is the yardstick, bound: 2x its error and < 5e-7 relative)."""
    from deepsee_amd import lib as L
    import ctypes as C
    g = torch.Generator().manual_seed(groups * 1000 + k)
    a = torch.randn(groups * tg, k, generator=g) * torch.rand(groups * tg, 1, generator=g).exp()
    b = torch.randn(groups, n, k, generator=g)
    ref = torch.einsum("gtk,gnk->gtn", a.view(groups, tg, k).double(), b.double()).reshape(groups * tg, n)
    f32 = torch.einsum("gtk,gnk->gtn", a.view(groups, tg, k), b).reshape(groups * tg, n)
    a3 = _split_rows(a).cuda()
    b3 = torch.stack([_split_rows(b[i]) for i in range(groups)]).cuda()
    c = torch.full((groups * tg, n), float("nan"), device="cuda")
    L.call("gemm_bf16x3", a3, b3, c, C.c_long(groups * tg), n, k, C.c_long(tg), n, tile)
    torch.cuda.synchronize()
    e_split = ((c.cpu().double() - ref).norm() / ref.norm()).item()
    e_f32 = ((f32.double() - ref).norm() / ref.norm()).item()
    print("bf16x3 vs f64: %.2e | cpu sgemm vs f64: %.2e" % (e_split, e_f32))
    assert e_split < 5e-7 and e_split <= 2 * e_f32, (e_split, e_f32)
    assert ((c.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 2e-6


@pytest.mark.parametrize("n,cin,cout,h,w,act,use_bias", [(2, 256, 3, 8, 64, 3, True), (1, 512, 3, 12, 128, 0, False),
                                                          (2, 512, 1, 4, 64, 1, True), (2, 128, 3, 32, 48, 3, True),
                                                          (1, 512, 2, 16, 16, 1, False)])
def test_thin_conv3x3(n, cin, cout, h, w, act, use_bias):
    """The to-RGB layer (512 -> 3, tanh): ops.conv2d routes it to the thin path -- a 27-output 1x1 GEMM + a 9-point gather (tiled
    through LDS when H, W are multiples of 16), backward on the channel-walking kernels of thin.hip (exact fp32 FMAs, different
    summation order than ATen)."""
    from deepsee_amd import ops
    g = torch.Generator().manual_seed(cin + cout + w)
    x = torch.randn(n, cin, h, w, generator=g).requires_grad_()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).requires_grad_()
    b = torch.randn(cout, generator=g).requires_grad_() if use_bias else None
    y = F.conv2d(x, wt, b, padding=1)
    y = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 3: torch.tanh}[act](y)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = nhwc(x.detach()).cuda().requires_grad_()
    wd = wt.detach().cuda().requires_grad_()
    bd = b.detach().cuda().requires_grad_() if use_bias else None
    yd = ops.conv2d(xd, wd, bd, None, 1, 1, 0, act)
    assert yd.grad_fn is not None and yd.shape[-1] == 4
    yd.backward(nhwc(gy).cuda())
    torch.cuda.synchronize()
    assert rel(nchw(yd.detach().cpu(), cout), y.detach()) < 2e-5
    assert float(yd.detach()[..., cout:].abs().max()) == 0.0
    assert rel(nchw(xd.grad.cpu(), cin), x.grad) < 2e-5
    assert rel(wd.grad.cpu(), wt.grad) < 2e-5
    if use_bias:
        assert rel(bd.grad.cpu(), b.grad) < 2e-5


def test_bf16x3_adds_no_error_to_the_winograd_conv():
    """The same 3x3 convolution three ways against float64: direct fp32-MFMA implicit GEMM, Winograd with the 36 GEMMs
    on v_mfma_f32, Winograd with the GEMMs on the bf16 matrix cores (bf16x3 operand split).  The Winograd transforms
    set the error level (~10x the direct form); moving the GEMM to bf16x3 must not add to it."""
    from deepsee_amd import ops
    n, c, h = 2, 256, 64
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    xd, wd = nhwc(x).cuda(), w.cuda()
    errs = {}
    for name, wino, split in (("direct", False, False), ("winograd_f32", True, False), ("winograd_bf16x3", True, True)):
        with ops.KernelPlan(winograd=wino, gemm_split=split).active():
            y = ops.conv2d(xd, wd, None, None, 1, 1, 0, 0)
        torch.cuda.synchronize()
        errs[name] = float((nchw(y.cpu(), c).double() - ref).norm() / ref.norm())
    print(errs)
    assert errs["direct"] < 1e-6
    assert errs["winograd_f32"] < 2e-5
    assert errs["winograd_bf16x3"] <= 1.1 * errs["winograd_f32"] + 1e-7, errs


@pytest.mark.parametrize("groups,tg,n,k,tile", [(3, 256, 256, 160, 1), (2, 512, 256, 512, 2), (36, 128, 128, 32, 1),
                                                (1, 1024, 512, 1024, 0)])
def test_gemm_bf16x3_af32(groups, tg, n, k, tile):
    """Same contract with the A operand left in fp32 and split inside the kernel (dsee_gemm_bf16x3_af32): identical
    arithmetic, so the result must equal the pre-split kernel's bit for bit."""
    from deepsee_amd import lib as L
    import ctypes as C
    g = torch.Generator().manual_seed(groups * 77 + k)
    a = torch.randn(groups * tg, k, generator=g) * torch.rand(groups * tg, 1, generator=g).exp()
    b = torch.randn(groups, n, k, generator=g)
    a3 = _split_rows(a).cuda()
    b3 = torch.stack([_split_rows(b[i]) for i in range(groups)]).cuda()
    c0 = torch.full((groups * tg, n), float("nan"), device="cuda")
    c1 = torch.full((groups * tg, n), float("nan"), device="cuda")
    L.call("gemm_bf16x3", a3, b3, c0, C.c_long(groups * tg), n, k, C.c_long(tg), n, tile)
    L.call("gemm_bf16x3_af32", a.cuda(), b3, c1, C.c_long(groups * tg), n, k, C.c_long(tg), n, tile)
    torch.cuda.synchronize()
    assert torch.equal(c0, c1)


def _pow2_scale(amax):
    """dsee_pow2_scale on the host: the power of two that maps amax into [2^13, 2^14)."""
    import math
    return 2.0 ** (13 - math.floor(math.log2(amax))) if amax > 0 else 1.0


def _amax(value):
    """device-side maximum in the 64-line layout the kernels read (dsee_common.h: DSEE_AMAX_LINES x DSEE_AMAX_STRIDE)"""
    t = torch.zeros(64 * 32, device="cuda")
    t[0] = value
    return t


def _split2_rows(x, scale):
    """fp32 [rows][K] -> slab-major fp16x2 [K/16][rows][2][16] (int16 view) of scale * x, the layout
    dsee_gemm_f16x2_af32 reads for its B operand."""
    xs = x * scale
    h0 = xs.half()
    h1 = (xs - h0.float()).half()
    rows, k = x.shape
    return torch.stack([h0, h1], 0).view(2, rows, k // 16, 16).permute(2, 1, 0, 3).contiguous().view(torch.int16)


@pytest.mark.parametrize("groups,tg,n,k,tile,spread", [(3, 256, 256, 160, 1, 1.0), (2, 512, 256, 512, 2, 1.0),
                                                       (36, 128, 128, 32, 1, 1.0), (1, 1024, 512, 1024, 0, 1.0),
                                                       (2, 512, 256, 512, 2, 4.0), (2, 256, 128, 256, 1, 1e-3)])
def test_gemm_f16x2_is_fp32_accurate(groups, tg, n, k, tile, spread):
    """The two-term fp16 form (3 MFMA products, operands scaled by powers of two from their device-side maxima) against
    float64: same bound as bf16x3 -- at most 2x a CPU sgemm's error and < 5e-7 -- on O(1) data, on data with a
    log-normal dynamic range of e^(+-4 sigma) per row (spread 4) and on tiny operands (1e-3: the scale, not the fp16
    range, decides)."""
    from deepsee_amd import lib as L, ops
    g = torch.Generator().manual_seed(groups * 1000 + k)
    a = torch.randn(groups * tg, k, generator=g) * (torch.randn(groups * tg, 1, generator=g) * abs(spread)).exp()
    if spread < 1:
        a = torch.randn(groups * tg, k, generator=g) * spread
    b = torch.randn(groups, n, k, generator=g) * (spread if spread < 1 else 1.0)
    ref = torch.einsum("gtk,gnk->gtn", a.view(groups, tg, k).double(), b.double()).reshape(groups * tg, n)
    f32 = torch.einsum("gtk,gnk->gtn", a.view(groups, tg, k), b).reshape(groups * tg, n)
    am_a, am_b = float(a.abs().max()), float(b.abs().max())
    amax_a, amax_b = _amax(am_a), _amax(am_b)
    b2 = torch.stack([_split2_rows(b[i], _pow2_scale(am_b)) for i in range(groups)]).cuda()
    c = torch.full((groups * tg, n), float("nan"), device="cuda")
    L.call("gemm_f16x2_af32", a.cuda(), b2, c, groups * tg, n, k, tg, n, tile, amax_a, amax_b)
    torch.cuda.synchronize()
    e_split = ((c.cpu().double() - ref).norm() / ref.norm()).item()
    e_f32 = ((f32.double() - ref).norm() / ref.norm()).item()
    print("fp16x2 vs f64: %.2e | cpu sgemm vs f64: %.2e" % (e_split, e_f32))
    assert e_split < 5e-7 and e_split <= 2 * e_f32, (e_split, e_f32)
    assert ((c.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 2e-6
    # device-side maximum: the transform kernels' atomic max gives the same scalar as the host reduction
    slot = ops.amax_slot()
    L.call("absmax", a.cuda(), a.numel(), slot)
    assert float(slot.max()) == am_a


@pytest.mark.parametrize("kernel", ["w8", "w4"])
@pytest.mark.parametrize("groups,tg,n,k,bound", [(3, 256, 256, 160, 1.0), (2, 512, 512, 512, 100.0), (36, 256, 256, 32, 8.0),
                                                 (1, 8192, 512, 128, 100.0), (5, 768, 768, 64, 100.0), (300, 256, 256, 96, 2.0)])
def test_gemm_f16x2_pre_split_a(groups, tg, n, k, bound, kernel):
    """dsee_gemm_f16x2_pre: the NT GEMM with BOTH operands pre-split (A rows in the layout dsee_wino43_input_f16x2 writes,
    scaled with a bound known before the producer ran -- `bound` x the true maximum, i.e. log2(bound) bits of headroom
    given away) against float64: fp32-level accuracy (the two-term split still carries 22 - log2(bound) bits), every launch
    on NaN-poisoned LDS and bit-identical."""
    from deepsee_amd import lib as L
    # kernel = "w4": the one-wave-per-SIMD form (csrc/gemm_w4.hip, round 6; five-stage swizzled ring, requests four slabs
    # ahead, fragments of the next slab read between the MFMAs): same operands, and -- slabs in order, products smallest
    # first -- the same fp32 accumulation order, so the two kernels must agree bit for bit
    entry = "gemm_f16x2_pre_w4" if kernel == "w4" else "gemm_f16x2_pre"
    g = torch.Generator().manual_seed(groups * 100 + k)
    a = torch.randn(groups * tg, k, generator=g)
    b = torch.randn(groups, n, k, generator=g)
    ref = torch.einsum("gtk,gnk->gtn", a.view(groups, tg, k).double(), b.double()).reshape(groups * tg, n)
    am_a, am_b = float(a.abs().max()), float(b.abs().max())
    a2 = _split2_rows(a, _pow2_scale(bound * am_a)).cuda()
    b2 = torch.stack([_split2_rows(b[i], _pow2_scale(am_b)) for i in range(groups)]).cuda()
    sink = torch.zeros(1, device="cuda")
    first = None
    for it in range(3):
        L.call("selftest_lds_poison", sink)
        c = torch.full((groups * tg, n), float("nan"), device="cuda")
        L.call(entry, a2, b2, c, groups * tg, n, k, tg, n, _amax(am_a), float(bound), _amax(am_b))
        torch.cuda.synchronize()
        assert torch.isfinite(c).all()
        first = c.clone() if first is None else first
        assert torch.equal(c, first)
    err = ((first.cpu().double() - ref).norm() / ref.norm()).item()
    print("pre-split A (%s), bound %g: vs f64 %.2e" % (kernel, bound, err))
    assert err < (5e-7 if bound <= 8 else 4e-6)
    if kernel == "w4":
        c8 = torch.empty_like(first)
        L.call("gemm_f16x2_pre", a2, b2, c8, groups * tg, n, k, tg, n, _amax(am_a), float(bound), _amax(am_b))
        assert torch.equal(c8, first), "the two kernels accumulate in the same order"


def test_gemm_f16x2_tn_long_chain_accuracy():
    """The 256x256 TN tile keeps ONE fp32 accumulator chain per split (no second level: 128 accumulator registers are
    all a wave has): at the step's longest chain, 4096 tiles per split (512x512 layer at 256^2, bs = 8, split-K 8), on
    zero-mean and on positive-mean operands (sums that grow), the result must stay below 1e-6 of float64 and below the
    error of the library fp32 GEMM (torch / rocBLAS) on the same data."""
    from deepsee_amd import lib as L
    g = torch.Generator().manual_seed(3)
    groups, t, rp, rq, splits = 2, 8192, 256, 256, 2
    for off in (0.0, 0.5):
        p = (torch.randn(groups * t, rp, generator=g) + off).cuda()
        q = (torch.randn(groups * t, rq, generator=g) + off).cuda()
        c = torch.empty(groups * splits, rp, rq, device="cuda")
        L.call("gemm_f16x2_tn_f32", p, q, c, groups, t, rp, rq, rq, splits, _amax(float(p.abs().max())), _amax(float(q.abs().max())))
        ts = t // splits
        pz, qz = p.view(groups * splits, ts, rp), q.view(groups * splits, ts, rq)
        ref = torch.einsum("ztp,ztq->zpq", pz.double(), qz.double())
        f32 = torch.einsum("ztp,ztq->zpq", pz, qz)
        e = ((c.double() - ref).norm() / ref.norm()).item()
        e32 = ((f32.double() - ref).norm() / ref.norm()).item()
        print("mean %.1f: fp16x2 TN 256x256 %.2e | library sgemm %.2e" % (off, e, e32))
        assert e < 1e-6 and e <= e32, (e, e32)


@pytest.mark.parametrize("groups,t,rp,rq,splits", [(2, 1024, 256, 128, 2), (3, 512, 256, 160, 1), (36, 256, 512, 512, 1)])
def test_gemm_f16x2_tn_matches_float64(groups, t, rp, rq, splits):
    """Split-K "TN" weight-gradient form with both fp32 operands transposed, scaled and split inside the kernel
    (dsee_gemm_f16x2_tn_f32, 256x128 and 256x160 tiles): C[g*splits+s] = P[g, tiles of s]^T Q[g, tiles of s]."""
    from deepsee_amd import lib as L
    g = torch.Generator().manual_seed(t + rq)
    p = torch.randn(groups * t, rp, generator=g) * 3.0
    q = torch.randn(groups * t, rq, generator=g) * 0.02
    amax_p, amax_q = _amax(float(p.abs().max())), _amax(float(q.abs().max()))
    c = torch.full((groups * splits, rp, rq), float("nan"), device="cuda")
    L.call("gemm_f16x2_tn_f32", p.cuda(), q.cuda(), c, groups, t, rp, rq, rq, splits, amax_p, amax_q)
    c3 = torch.full((groups * splits, rp, rq), float("nan"), device="cuda")
    L.call("gemm_bf16x3_tn_f32", p.cuda(), q.cuda(), c3, groups, t, rp, rq, rq, splits)
    torch.cuda.synchronize()
    ts = t // splits
    ref = torch.einsum("ztp,ztq->zpq", p.view(groups * splits, ts, rp).double(), q.view(groups * splits, ts, rq).double())
    f32 = torch.einsum("ztp,ztq->zpq", p.view(groups * splits, ts, rp), q.view(groups * splits, ts, rq))
    e2 = ((c.cpu().double() - ref).norm() / ref.norm()).item()
    e3 = ((c3.cpu().double() - ref).norm() / ref.norm()).item()
    e32 = ((f32.double() - ref).norm() / ref.norm()).item()
    print("TN fp16x2 %.2e | bf16x3 %.2e | cpu sgemm %.2e" % (e2, e3, e32))
    assert e2 < 5e-7 and e2 <= 2 * max(e32, e3), (e2, e3, e32)


@pytest.mark.parametrize("groups,t,rp,rq,splits,bound", [(2, 1024, 256, 256, 2, 1.0), (3, 512, 256, 160, 1, 100.0),
                                                         (36, 256, 512, 512, 1, 100.0), (8, 2048, 1024, 160, 4, 8.0)])
def test_gemm_f16x2_tn_pre_split_q(groups, t, rp, rq, splits, bound):
    """dsee_gemm_f16x2_tn_qpre: the weight-gradient TN GEMM whose Q operand is the PRE-SPLIT V of the forward pass
    (dsee_wino43_input_f16x2's layout [rq/16][groups*t][2][16] fp16, tile-major) -- the MFMA fragments (8 consecutive tiles
    of one channel) come out of ds_read_b64_tr_b16 on the landed bytes.  Against float64, on NaN-poisoned LDS, bit-identical
    from launch to launch, and against the fp32-Q form on the same data."""
    from deepsee_amd import lib as L
    g = torch.Generator().manual_seed(t + rq)
    p = torch.randn(groups * t, rp, generator=g) * 3.0
    q = torch.randn(groups * t, rq, generator=g) * 0.02
    am_p, am_q = float(p.abs().max()), float(q.abs().max())
    q2 = _split2_rows(q, _pow2_scale(bound * am_q)).cuda()
    ts = t // splits
    ref = torch.einsum("ztp,ztq->zpq", p.view(groups * splits, ts, rp).double(), q.view(groups * splits, ts, rq).double())
    sink = torch.zeros(1, device="cuda")
    first = None
    for it in range(3):
        L.call("selftest_lds_poison", sink)
        c = torch.full((groups * splits, rp, rq), float("nan"), device="cuda")
        L.call("gemm_f16x2_tn_qpre", p.cuda(), q2, c, groups, t, rp, rq, rq, splits, _amax(am_p), _amax(am_q), float(bound))
        torch.cuda.synchronize()
        assert torch.isfinite(c).all()
        first = c.clone() if first is None else first
        assert torch.equal(c, first)
    err = ((first.cpu().double() - ref).norm() / ref.norm()).item()
    print("TN with pre-split Q, bound %g: vs f64 %.2e" % (bound, err))
    assert err < (5e-7 if bound <= 8 else 4e-6)


@pytest.mark.parametrize("groups,t,rp,rq,splits", [(2, 1024, 256, 256, 2), (3, 512, 256, 160, 1), (36, 256, 512, 512, 1),
                                                   (4, 512, 256, 128, 2)])
def test_gemm_f16x2_tn_both_operands_pre_split(groups, t, rp, rq, splits):
    """dsee_gemm_f16x2_tn_pqpre: P (A dY A^T, bound 225) and Q (B^T d B, bound 100) both arrive pre-split in the tile-major
    image of dsee_wino43_dout_f16x2 / dsee_wino43_input_f16x2; every MFMA fragment comes out of an LDS transpose read.
    Against float64, on NaN-poisoned LDS, bit-identical from launch to launch."""
    from deepsee_amd import lib as L
    g = torch.Generator().manual_seed(3 * t + rq)
    p = torch.randn(groups * t, rp, generator=g) * 3.0
    q = torch.randn(groups * t, rq, generator=g) * 0.02
    am_p, am_q = float(p.abs().max()), float(q.abs().max())
    # the producers scale with bound x max|input|: emulate inputs whose transform maxima sit 225 / 100 above them
    p2 = _split2_rows(p, _pow2_scale(225.0 * am_p / 40.0)).cuda()
    q2 = _split2_rows(q, _pow2_scale(100.0 * am_q / 12.0)).cuda()
    ts = t // splits
    ref = torch.einsum("ztp,ztq->zpq", p.view(groups * splits, ts, rp).double(), q.view(groups * splits, ts, rq).double())
    sink = torch.zeros(1, device="cuda")
    first = None
    for it in range(3):
        L.call("selftest_lds_poison", sink)
        c = torch.full((groups * splits, rp, rq), float("nan"), device="cuda")
        L.call("gemm_f16x2_tn_pqpre", p2, q2, c, groups, t, rp, rq, rq, splits, _amax(am_p / 40.0), 225.0,
               _amax(am_q / 12.0), 100.0)
        torch.cuda.synchronize()
        assert torch.isfinite(c).all()
        first = c.clone() if first is None else first
        assert torch.equal(c, first)
    err = ((first.cpu().double() - ref).norm() / ref.norm()).item()
    print("TN with both operands pre-split: vs f64 %.2e" % err)
    assert err < 2e-6


@pytest.mark.parametrize("n,h,c", [(2, 32, 128), (8, 16, 512), (1, 64, 256), (2, 32, 48)])
def test_dout_transform_pre_split_with_channel_sums(n, h, c):
    """dsee_wino43_dout_f16x2: A dY A^T (with the row factors every dM carries) written as the pre-split fp16x2 image -- scale
    from DM_BOUND x max|dY|, known before the kernel runs -- equals the fp32 transform of dsee_wino43_dout to 2^-20 of every
    position's own maximum, and the bias / noise-weight gradients that ride along equal the separate channel_dot / channel_dot_rng
    passes."""
    from deepsee_amd import lib as L, ops
    g = torch.Generator().manual_seed(n + h + c)
    dy = (torch.randn(n, h, h, c, generator=g) * 0.37).cuda()
    t = n * (h // 4) ** 2
    ref, ra = ops.new(36, t, c), ops.amax_slot()
    L.call("wino43_dout", dy, ref, n, h, h, c, ra)
    am = ops.tensor_amax(dy)
    dm2 = ops._i16(36 * t * c * 2)
    ws = ops.scratch(L.lib().dsee_wino43_dout_f16x2_workspace(), "doutsums2")
    db, d0, d1 = ops.new(c), ops.new(c), ops.new(c)
    L.call("wino43_dout_f16x2", dy, dm2, n, h, h, c, am, ops.DM_BOUND, ws, db, d0, 11, 4096, d1, 12, 8192)
    only = ops._i16(36 * t * c * 2)
    L.call("wino43_dout_f16x2", dy, only, n, h, h, c, am, ops.DM_BOUND, None, None, None, 0, 0, None, 0, 0)
    torch.cuda.synchronize()
    assert torch.equal(only, dm2)
    # every dM holds f_i f_j (A dY A^T)[i][j] (row factors): against the transform written out in float64 ...
    a = torch.tensor([[1, 0, 0, 0], [1, 1, 1, 1], [1, -1, 1, -1], [1, 2, 4, 8], [1, -2, 4, -8], [0, 0, 0, 1]], dtype=torch.float64)
    tiles = dy.cpu().double().reshape(n, h // 4, 4, h // 4, 4, c).permute(0, 1, 3, 5, 2, 4).reshape(t, c, 4, 4)
    exact = torch.einsum("ik,tckl,jl->ijtc", a, tiles, a).reshape(36, t, c) * _dm_row_factors("cpu").double()
    assert float((ref.cpu().double() - exact).abs().max()) <= 1e-6 * float(exact.abs().max())
    # ... bounded by max|dY| at EVERY position, so the pre-split image keeps 2^-20 of each position's OWN maximum (one scale for
    # the unfactored transform would leave the corner positions 225x = 7.8 bits short)
    assert ops.DM_BOUND == 1.0 and float(ref.abs().max()) <= float(dy.abs().max())
    sc = _pow2_scale(ops.DM_BOUND * float(dy.abs().max()))
    dec = dm2.view(torch.float16).view(c // 16, 36 * t, 2, 16).float().sum(2).permute(1, 0, 2).reshape(36, t, c) / sc
    for xi in range(36):
        assert float((dec[xi] - ref[xi]).abs().max()) <= 2.0 ** -20 * float(ref[xi].abs().max()), xi
    want_b = ops.channel_dot(dy, None, c)
    assert rel(db.cpu(), want_b.cpu()) < 1e-5
    m = n * h * h
    wsd = ops.scratch(L.lib().dsee_channel_dot_workspace(m, c), "chdot")
    for got, seed, off in ((d0, 11, 4096), (d1, 12, 8192)):
        want = ops.new(c)
        L.call("channel_dot_rng", dy, want, m, c, wsd, seed, off)
        torch.cuda.synchronize()
        assert rel(got.cpu(), want.cpu()) < 1e-5


def _dm_row_factors(device="cuda"):
    """[36, 1, 1] factors f_i f_j, f = (1, 1/4, 1/4, 1/16, 1/16, 1), the pre-split dM images carry (include/deepsee_hip.h, "ROW
    FACTORS"): position (i, j) of A dY A^T is bounded by r_i r_j max|dY| with r = the absolute row sums (1, 4, 4, 15, 15, 1) of A."""
    f = torch.tensor([1.0, 0.25, 0.25, 0.0625, 0.0625, 1.0], device=device)
    return (f[:, None] * f[None, :]).reshape(36, 1, 1)


def _sign_words(out):
    """The LeakyReLU branch of an NHWC tensor as the fused forward writes it: [C/32][pixel] words, bit 8 * (c % 4) + (c % 32) // 4
    = (out > 0)."""
    n, h, w, c = out.shape
    bits = (out > 0).reshape(n * h * w, c // 32, 8, 4).transpose(2, 3).reshape(n * h * w, c // 32, 32).to(torch.int64)
    words = (bits << torch.arange(32, device=out.device)).sum(2)
    return (words - ((words >> 31) << 32)).to(torch.int32).t().reshape(-1)    # (two's complement into int32)


@pytest.mark.parametrize("n,h,c", [(2, 32, 64), (1, 64, 128), (4, 32, 512)])
def test_norm_backward_reduce_writes_pre_split_gradient(n, h, c):
    """dsee_modulate_bwd_reduce_wino_f16x2: the gamma/beta gradient A (g*xhat | g) A^T leaves the norm backward's reduce pass
    as the pre-split fp16x2 image, scaled from the a-priori bound 225 x max|dh| x max(1, max|xhat|) (dsee_amax_product of two
    maxima written by earlier kernels).  Equal to the fp32 form to 2^-21 of the tensor maximum; identical per-channel sums; the
    bound really bounds."""
    from deepsee_amd import lib as L, ops
    g = torch.Generator().manual_seed(n * h + c)
    x = (torch.randn(n, h, h, c, generator=g) * 2 + 0.3).cuda()
    dh = (torch.randn(n, h, h, c, generator=g) * 0.01).cuda()
    out = torch.randn(n, h, h, c, generator=g).cuda()             # (only its sign is used: LeakyReLU branch)
    scale = (torch.rand(n, h, h, c, generator=g) + 0.5).half().float().cuda()      # (values fp16 holds exactly: see scale16 below)
    mean, invstd = (torch.randn(c, generator=g) * 0.3).cuda(), (torch.rand(c, generator=g) + 0.5).cuda()
    rows, t = 2 * c, n * (h // 4) ** 2
    ws = ops.scratch(L.lib().dsee_modulate_bwd_wino_workspace(n, h, h, c), "norm")
    ref, ra, sums_ref = ops.new(36, t, rows), ops.amax_slot(), ops.new(4, c)
    L.call("modulate_bwd_reduce_wino", dh, out, x, scale, mean, invstd, ref, rows, sums_ref, n, h, h, c, 0.2, ws, ra)
    a_dh, a_xh = ops.tensor_amax(dh), ops.tensor_amax(((x - mean) * invstd).contiguous())
    ga = ops.amax_slot()
    L.call("amax_product", a_dh, a_xh, 1.0, ga)
    dm2, sums = ops._i16(36 * t * rows * 2), ops.new(4, c)
    L.call("modulate_bwd_reduce_wino_f16x2", dh, out, x, scale, mean, invstd, dm2, rows, sums, n, h, h, c, 0.2, ws, ga,
           ops.DM_BOUND, None)
    torch.cuda.synchronize()
    bound = ops.DM_BOUND * float(ga.max())
    want = float(dh.abs().max()) * max(1.0, float(((x - mean) * invstd).abs().max()))
    assert abs(float(ga.max()) - want) <= 1e-6 * want
    assert float(ref.abs().max()) <= bound                     # (dM = f_i f_j (A . A^T)[i][j]: bounded at every position)
    dec = dm2.view(torch.float16).view(rows // 16, 36 * t, 2, 16).float().sum(2).permute(1, 0, 2).reshape(36, t, rows)
    dec = dec / _pow2_scale(bound)
    for xi in range(36):
        assert float((dec[xi] - ref[xi]).abs().max()) <= 2.0 ** -19 * float(ref[xi].abs().max()), xi
    assert rel(sums.cpu(), sums_ref.cpu()) < 1e-5          # (same sums, folded in a different fixed order)
    # 16-bit storage mode: the same pass writing the packed one-term image (one scaled fp16 term per element)
    dm1, sums1 = ops._i16(36 * t * rows), ops.new(4, c)
    # (... reading the fp16 modulation factor dsee_spade_fused_fwd_f16p saves)
    scale16 = scale.half()
    L.call("modulate_bwd_reduce_wino_f16p", dh, out, x, scale16, mean, invstd, dm1, rows, sums1, n, h, h, c, 0.2, ws, ga,
           ops.DM_BOUND, None)
    torch.cuda.synchronize()
    dec1 = dm1.view(torch.float16).view(rows // 32, 36 * t, 32).permute(1, 0, 2).reshape(36, t, rows).float() / _pow2_scale(bound)
    for xi in range(36):
        assert float((dec1[xi] - ref[xi]).abs().max()) <= 2.0 ** -10 * float(ref[xi].abs().max()), xi
    assert torch.equal(sums1, sums)
    # the LeakyReLU branch from the fused forward's bit mask instead of `out` (h = NULL): the same bits out of both passes
    mask = _sign_words(out)
    for name, want_dm, width in (("modulate_bwd_reduce_wino_f16x2", dm2, 2), ("modulate_bwd_reduce_wino_f16p", dm1, 1)):
        dmm, sm = ops._i16(36 * t * rows * width), ops.new(4, c)
        L.call(name, dh, None, x, scale if width == 2 else scale16, mean, invstd, dmm, rows, sm, n, h, h, c, 0.2, ws, ga,
               ops.DM_BOUND, mask)
        torch.cuda.synchronize()
        assert torch.equal(dmm, want_dm) and torch.equal(sm, sums)
    dx0, dx1, da0, da1 = torch.empty_like(x), torch.empty_like(x), ops.amax_slot(), ops.amax_slot()
    L.call("modulate_bwd_apply_amax", dh, out, x, scale, mean, invstd, sums, None, dx0, n, h * h, c, 1.0 / (n * h * h), 0.2, da0,
           0, None)
    L.call("modulate_bwd_apply_amax", dh, None, x, scale, mean, invstd, sums, None, dx1, n, h * h, c, 1.0 / (n * h * h), 0.2, da1,
           0, mask)
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1) and torch.equal(da0, da1) and float(dx0.abs().max()) > 0
    dx2, da2 = torch.empty_like(x), ops.amax_slot()        # the fp16 form of the same scale: the same bits
    L.call("modulate_bwd_apply_amax", dh, None, x, scale16, mean, invstd, sums, None, dx2, n, h * h, c, 1.0 / (n * h * h), 0.2, da2,
           1, mask)
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx2) and torch.equal(da0, da2)


def test_f16x2_special_values():
    """Edges of the operand split: zero operands (scale 1, result exactly 0), values 2^20 below the operand maximum
    (h1 subnormal: absolute error <= 2^-39 of the maximum), huge / tiny magnitudes (the power-of-two scale keeps fp16 in
    range), and inf / nan propagate to the affected outputs only."""
    from deepsee_amd import lib as L
    n = k = 128
    def run(a, b):
        c = torch.full((128, n), float("nan"), device="cuda")
        am_a, am_b = float(a.abs().nan_to_num(0, 0, 0).max()), float(b.abs().nan_to_num(0, 0, 0).max())
        if not torch.isfinite(a).all():
            am_a = float("inf")
        amax_a, amax_b = _amax(am_a), _amax(am_b)
        sb = _pow2_scale(am_b) if am_b > 0 and am_b != float("inf") else 1.0
        L.call("gemm_f16x2_af32", a.cuda(), _split2_rows(b, sb)[None].cuda(), c, 128, n, k, 128, n, 1, amax_a, amax_b)
        torch.cuda.synchronize()
        return c.cpu()
    g = torch.Generator().manual_seed(1)
    b = torch.randn(n, k, generator=g)
    assert torch.equal(run(torch.zeros(128, k), b), torch.zeros(128, n))
    a = torch.randn(128, k, generator=g)
    a[5] *= 2.0 ** -20                                           # a row far below the maximum
    ref = a.double() @ b.double().t()
    c = run(a, b)
    assert float((c[5].double() - ref[5]).abs().max()) < 2.0 ** -36 * float(a.abs().max()) * float(b.abs().max()) * k
    for s in (1e30, 1e-30):                                      # magnitudes far outside the fp16 range
        c = run(a * s, b)
        assert float(((c.double() / s) - ref).norm() / ref.norm()) < 5e-7
    a2 = a.clone()
    a2[7, 3] = float("inf")
    c = run(a2, b)
    assert not torch.isfinite(c[7]).any() or torch.isnan(c[7]).any() or torch.isinf(c[7]).any()
    assert torch.isfinite(c[8]).all()


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3", "f16"])
def test_pipelined_gemms_with_poisoned_lds(mode):
    """Hazard check of the hand-pipelined GEMM loops (LDS-DMA rings, vmcnt / barrier protocol): before every launch the
    LDS of all CUs is filled with NaN patterns (dsee_selftest_lds_poison), so a fragment or conversion read of a stage
    whose data has not landed yields NaN instead of stale data from the previous launch.  NT form (128 and 256 tiles,
    short and long K, one and several tiles per block) and the split-K TN form; every launch must be finite and
    bit-identical to the first."""
    from deepsee_amd import lib as L
    g = torch.Generator().manual_seed(5)
    sink = torch.zeros(1, device="cuda")

    def nt(groups, tg, n, k, tile):
        a = torch.randn(groups * tg, k, generator=g).cuda()
        b = torch.randn(groups, n, k, generator=g)
        am_a, am_b = _amax(float(a.abs().max())), _amax(float(b.abs().max()))
        c = torch.empty(groups * tg, n, device="cuda")
        if mode == "f16x2":
            b2 = torch.stack([_split2_rows(b[i], _pow2_scale(float(b.abs().max()))) for i in range(groups)]).cuda()
            return lambda: (L.call("gemm_f16x2_af32", a, b2, c, groups * tg, n, k, tg, n, tile, am_a, am_b), c)[1]
        if mode == "f16":
            xs = b * _pow2_scale(float(b.abs().max()))
            b1 = xs.half().view(groups, n, k // 16, 16).permute(0, 2, 1, 3).contiguous().view(torch.int16).cuda()
            cs = torch.zeros(1, device="cuda")
            return lambda: (L.call("gemm_f16_af32", a, b1, c, groups * tg, n, k, tg, n, tile, am_a, am_b, 0, cs), c)[1]
        b3 = torch.stack([_split_rows(b[i]) for i in range(groups)]).cuda()
        return lambda: (L.call("gemm_bf16x3_af32", a, b3, c, groups * tg, n, k, tg, n, tile), c)[1]

    def tn(groups, t, rp, rq, splits):
        p = torch.randn(groups * t, rp, generator=g).cuda()
        q = torch.randn(groups * t, rq, generator=g).cuda()
        am_p, am_q = _amax(float(p.abs().max())), _amax(float(q.abs().max()))
        c = torch.empty(groups * splits, rp, rq, device="cuda")
        name = {"f16x2": "gemm_f16x2_tn_f32", "f16": "gemm_f16_tn_f32", "bf16x3": "gemm_bf16x3_tn_f32"}[mode]
        extra = () if mode == "bf16x3" else (am_p, am_q)
        return lambda: (L.call(name, p, q, c, groups, t, rp, rq, rq, splits, *extra), c)[1]

    runs = [nt(2, 256, 128, 32, 1), nt(1, 256, 256, 16, 2), nt(36, 4096, 256, 160, 2), nt(40, 2048, 512, 512, 2),
            nt(600, 128, 128, 64, 1), tn(2, 256, 256, 128, 1), tn(72, 512, 256, 160, 2), tn(36, 1024, 512, 256, 4)]
    for run in runs:
        first = None
        for it in range(6):
            L.call("selftest_lds_poison", sink)
            out = run().clone()
            assert torch.isfinite(out).all(), "NaN from a poisoned LDS stage (launch %d)" % it
            if first is None:
                first = out
            assert torch.equal(out, first)


@pytest.mark.parametrize("waves", [8, 16, 4])
@pytest.mark.parametrize("packed", [False, True], ids=["f16x2", "f16p"])
@pytest.mark.parametrize("n,h,c,per_image,with_scale", [(2, 32, 64, True, True), (1, 64, 128, False, True),
                                                        (3, 32, 128, True, False), (2, 64, 64, False, False)])
def test_spade_fused_forward_vs_float64(n, h, c, per_image, with_scale, packed, waves, monkeypatch):
    """dsee_spade_fused_fwd (round 3: gamma/beta Winograd GEMM with the output transform folded in registers, normalise +
    modulate + LeakyReLU epilogue; normalization.py:107-120, 167-213) through the C ABI against a float64 restatement
    of the same layer on the CPU: direct 3x3 convolution over [embedding | one-hot] with shared weights and per-image
    table weights, packed gamma/beta row order, BN with given statistics, modulate, LeakyReLU(0.2).  Operands: the
    pre-split fp16x2 transform of dsee_wino43_input_f16x2 (scale from max|cat| x 100, known before the transform) and
    dsee_wino43_weights[_table].  Every launch runs on NaN-poisoned LDS (a fragment read of a ring slot whose LDS-DMA
    has not landed would show) and must be bit-identical to the first.
    `packed`: the 16-bit storage mode's form of the same kernel (dsee_spade_fused_fwd_f16p) on packed one-term operands
    (dsee_wino43_input_f16p, weights split = 4): K = 160 is 2.5 pieces of 64 k's (the missing half piece is fetched as zeros);
    the result carries the fp16 rounding of the operands (per-layer 0.3-0.4 %).
    `waves` = 16: round 5's form of the kernel with sixteen waves of 128 registers per workgroup (one 16 x 16 block each;
    DSEE_FUSED_W16=1, read per launch) -- measured 6 % slower than the shipped 8-wave form and therefore not the default, kept
    under the same test."""
    from deepsee_amd import lib as L, ops
    if waves == 4 and packed:
        pytest.skip("the one-wave-per-SIMD kernel takes the two-term operands only")
    monkeypatch.setenv("DSEE_FUSED_W16", "1" if waves == 16 else "0")
    g = torch.Generator().manual_seed(100 * n + h + c)
    K, rows, ca = (160 if per_image else 128), 2 * c, 128
    cat = torch.rand(n, h, h, K, generator=g)
    if per_image:
        cat[..., ca:] = 0.0
        lab = torch.randint(0, 19, (n, h, h), generator=g)
        cat[..., ca:].scatter_(3, lab[..., None], 1.0)                      # one-hot label channels
    x = torch.randn(n, h, h, c, generator=g) * 3 + 0.5
    mean, invstd = torch.randn(c, generator=g) * 0.3, torch.rand(c, generator=g) + 0.5
    wg, wb = torch.randn(c, ca, 3, 3, generator=g) * 0.05, torch.randn(c, ca, 3, 3, generator=g) * 0.05
    bg, bb = torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1
    idx, prow = ops.packed_perm(c, "cpu")
    assert prow == rows
    w2a = torch.cat([wg, wb, torch.zeros(1, ca, 3, 3)]).index_select(0, idx).contiguous()
    b2 = torch.cat([bg, bb, torch.zeros(1)]).index_select(0, idx).contiguous()
    tg = tb = None
    if per_image:
        tg, tb = torch.randn(n, 9, c, 19, generator=g) * 0.05, torch.randn(n, 9, c, 19, generator=g) * 0.05
        t2 = torch.cat([tg, tb, torch.zeros(n, 9, 1, 19)], 2).index_select(2, idx)
        table = F.pad(t2, (0, 13)).contiguous()                               # [N, 9, rows, 32]
    # ---- float64 reference
    catn = cat.double().permute(0, 3, 1, 2)
    gam = F.conv2d(catn[:, :ca], wg.double(), bg.double(), padding=1)
    bet = F.conv2d(catn[:, :ca], wb.double(), bb.double(), padding=1)
    if per_image:
        for i in range(n):
            w_g = tg[i].double().permute(1, 2, 0).reshape(c, 19, 3, 3)        # [c][r][tap]
            w_b = tb[i].double().permute(1, 2, 0).reshape(c, 19, 3, 3)
            gam[i:i + 1] += F.conv2d(catn[i:i + 1, ca:ca + 19], w_g, None, padding=1)
            bet[i:i + 1] += F.conv2d(catn[i:i + 1, ca:ca + 19], w_b, None, padding=1)
    add_one = 1.0
    xh = (x.double().permute(0, 3, 1, 2) - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    sc_ref = gam + add_one
    ref = F.leaky_relu(xh * sc_ref + bet, 0.2)
    # ---- HIP
    catd, xd = cat.cuda(), x.cuda()
    t = n * (h // 4) ** 2
    ac = ops.tensor_amax(catd)
    terms, sp = (1, 4) if packed else (2, 2)
    v2 = ops._i16(36 * t * K * terms)
    L.call("wino43_input_f16p" if packed else "wino43_input_f16x2", catd, v2, n, h, h, K, ac, 100.0)
    if per_image:
        ua = ops.weight_amax(w2a.cuda(), table.cuda())
        u = ops._i16(36 * n * rows * K * terms)
        L.call("wino43_weights_table", w2a.cuda(), table.cuda(), u, n, rows, ca, sp, ua)
    else:
        u, ua = ops._wino_u(w2a.cuda(), rows, ca, False, rows, K, sp)
    # (16-bit storage mode: the saved modulation factor is fp16 too)
    out, sc = torch.empty_like(xd), (torch.empty(xd.shape, dtype=torch.float16 if packed else torch.float32, device="cuda")
                                     if with_scale else None)
    sink = torch.zeros(1, device="cuda")
    first = None
    for it in range(4):
        L.call("selftest_lds_poison", sink)
        out.fill_(float("nan"))
        hm, xm = ops.amax_slot(), ops.amax_slot()
        mask = torch.full((n * h * h * (c // 32),), 0x5a5a5a5a, dtype=torch.int32, device="cuda") if with_scale else None
        entry = "spade_fused_fwd_f16p" if packed else ("spade_fused_fwd_w4" if waves == 4 else "spade_fused_fwd")
        L.call(entry, v2, u, ac, 100.0, ua, b2.cuda(), xd, mean.cuda(),
               invstd.cuda(), out, sc, n, h, h, c, rows, K, n if per_image else 1, add_one, 0.2, hm, xm, mask)
        torch.cuda.synchronize()
        if mask is not None:            # the LeakyReLU branch of h for the backward pass, one bit per element
            assert torch.equal(mask, _sign_words(out))
        assert float(hm.max()) == float(out.abs().max())      # the maximum the consumer's operand scale is built from
        xh_max = float((((xd - mean.cuda()) * invstd.cuda()).abs()).max())
        assert abs(float(xm.max()) - xh_max) <= 1e-5 * xh_max
        assert torch.isfinite(out).all(), "NaN from a poisoned LDS stage (launch %d)" % it
        if first is None:
            first = out.clone()
        assert torch.equal(out, first)
    if waves == 4:
        # round 6: the one-wave-per-SIMD kernel (csrc/spade_fused_w4.hip) keeps the 8-wave kernel's arithmetic order: bit-identical
        o8, s8 = torch.empty_like(out), (torch.empty_like(sc) if with_scale else None)
        L.call("spade_fused_fwd", v2, u, ac, 100.0, ua, b2.cuda(), xd, mean.cuda(), invstd.cuda(), o8, s8, n, h, h, c, rows, K,
               n if per_image else 1, add_one, 0.2, ops.amax_slot(), ops.amax_slot(), None)
        torch.cuda.synchronize()
        assert torch.equal(o8, out) and (not with_scale or torch.equal(s8, sc))
    e_h = rel(out.cpu().double().permute(0, 3, 1, 2), ref)
    print("fused SPADE forward N=%d %dx%d C=%d K=%d %s vs float64: h %.1e" % (n, h, h, c, K, "packed one-term" if packed else "", e_h))
    assert e_h < (8e-3 if packed else 2e-6)
    if with_scale:
        assert rel(sc.cpu().double().permute(0, 3, 1, 2), sc_ref) < (8e-3 if packed else 2e-6)


@pytest.mark.parametrize("shift", [0, 16, 20])
def test_small_channel_keeps_its_precision_in_the_winograd_conv(shift):
    """The fp16x2 operand split uses ONE power-of-two scale per tensor.  A channel whose values sit 2^-20 below the tensor
    maximum must still come through with fp32-level accuracy: its high term lands in fp16's normal range (2^-5 after
    scaling), its low term in the subnormals with an absolute step of 2^-24 (relative 2^-19).  Output channel 5 reads
    ONLY the small input channel 7, gradient channel 9 is 2^-20 below the others; the outputs / weight gradients that
    depend on the small operands alone are held to 1e-3 relative to THEIR OWN size against float64 (measured ~1e-5)."""
    from deepsee_amd import ops
    n, c, h, small = 2, 256, 64, 2.0 ** -shift
    assert ops._wino_ok(n, h, h, c, c, 3, 1, 1, 0) and ops.P().gemm_f16x2 and not ops.P().half
    g = torch.Generator().manual_seed(2020)
    x = torch.randn(n, c, h, h, generator=g)
    x[:, 7] *= small
    w = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    w[5] = 0.0
    w[5, 7] = torch.randn(3, 3, generator=g)
    gy = torch.randn(n, c, h, h, generator=g)
    gy[:, 9] *= small
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    y = F.conv2d(xr, wr, padding=1)
    y.backward(gy.double())
    xd, wd = nhwc(x).cuda().requires_grad_(), w.cuda().requires_grad_()
    yd = ops.conv2d(xd, wd, None, None, 1, 1, 0, 0)
    yd.backward(nhwc(gy).cuda())
    torch.cuda.synchronize()
    yh = nchw(yd.detach().cpu(), c).double()
    e_out = rel(yh[:, 5], y.detach()[:, 5])                     # output that depends on the small channel only
    e_dw_x = rel(wd.grad.cpu().double()[:, 7], wr.grad[:, 7])   # dw columns fed by the small input channel
    e_dw_g = rel(wd.grad.cpu().double()[9], wr.grad[9])         # dw rows fed by the small gradient channel
    e_all = rel(yh, y.detach())
    print("channel 2^-%d below the maximum: y[5] %.1e, dw[:,7] %.1e, dw[9] %.1e (whole tensor %.1e)"
          % (shift, e_out, e_dw_x, e_dw_g, e_all))
    assert e_all < 1e-4
    assert e_out < 1e-3 and e_dw_x < 1e-3
    # (round 4: 2.5e-3 at shift 20 before dM carried its row factors -- the corner positions of A dY A^T, which hold the outer
    # taps of dw, sat 225x below the one scale of the tensor)
    assert e_dw_g < 1e-3


def test_small_channel_keeps_its_precision_in_the_fused_spade_kernel():
    """Same question for dsee_spade_fused_fwd, whose V scale is the a-priori bound 100 x max|cat| (about 3 bits more headroom
    than the measured maximum): beta channel 5 reads only embedding channel 3, which is 2^-20 below the others, and x of that
    channel equals the batch mean, so out[..., 5] = LeakyReLU(beta_5) alone; held to 1e-3 of its own size vs float64."""
    from deepsee_amd import lib as L, ops
    n, h, c, K, ca, small = 1, 64, 64, 128, 128, 2.0 ** -20
    rows = 2 * c
    g = torch.Generator().manual_seed(77)
    cat = torch.rand(n, h, h, K, generator=g)
    cat[..., 3] *= small
    x = torch.randn(n, h, h, c, generator=g)
    mean, invstd = torch.randn(c, generator=g) * 0.3, torch.rand(c, generator=g) + 0.5
    x[..., 5] = mean[5]
    wg, wb = torch.randn(c, ca, 3, 3, generator=g) * 0.05, torch.randn(c, ca, 3, 3, generator=g) * 0.05
    wb[5] = 0.0
    wb[5, 3] = torch.randn(3, 3, generator=g)
    idx, prow = ops.packed_perm(c, "cpu")
    w2a = torch.cat([wg, wb, torch.zeros(1, ca, 3, 3)]).index_select(0, idx).contiguous()
    b2 = torch.zeros(prow)
    catn = cat.double().permute(0, 3, 1, 2)
    gam, bet = F.conv2d(catn, wg.double(), padding=1), F.conv2d(catn, wb.double(), padding=1)
    xh = (x.double().permute(0, 3, 1, 2) - mean.double()[None, :, None, None]) * invstd.double()[None, :, None, None]
    ref = F.leaky_relu(xh * (gam + 1.0) + bet, 0.2)
    catd, xd = cat.cuda(), x.cuda()
    ac = ops.tensor_amax(catd)
    v2 = ops._i16(36 * n * (h // 4) ** 2 * K * 2)
    L.call("wino43_input_f16x2", catd, v2, n, h, h, K, ac, ops.FUSED_V_BOUND)
    u, ua = ops._wino_u(w2a.cuda(), rows, ca, False, rows, K, 2)
    out = torch.empty_like(xd)
    L.call("spade_fused_fwd", v2, u, ac, ops.FUSED_V_BOUND, ua, b2.cuda(), xd, mean.cuda(), invstd.cuda(), out, None, n, h, h,
           c, rows, K, 1, 1.0, 0.2, None, None, None)
    torch.cuda.synchronize()
    got = out.cpu().double().permute(0, 3, 1, 2)
    e5, e_all = rel(got[:, 5], ref[:, 5]), rel(got, ref)
    print("fused SPADE, embedding channel 2^-20 below the maximum: out[5] %.1e (whole tensor %.1e)" % (e5, e_all))
    assert float(ref[:, 5].abs().max()) < 1e-4
    assert e5 < 1e-3 and e_all < 2e-6


@pytest.mark.parametrize("co,ci", [(512, 512), (128, 256), (256, 64), (64, 192)])
@pytest.mark.parametrize("flip", [0, 1, 2])
@pytest.mark.parametrize("split", [0, 2, 3, 4])
def test_wino43_weights_lds_staged_form_is_bit_identical(co, ci, flip, split):
    """dsee_wino43_weights stages the weights through LDS when whole 16 x 64 tiles exist (coalesced 16-byte loads instead of a 72-load
    gather per lane); a weight tensor at a 4-byte offset takes the gather kernel.  Same U, bit for bit, padding rows included --
    forward (flip 0), data-gradient (1: transposed + rotated) and adjoint (2: transposed) forms, all operand formats."""
    from deepsee_amd import lib as L
    g = torch.Generator().manual_seed(co + ci + flip)
    w = torch.randn(co, ci, 3, 3, generator=g)
    r_s, k_s = (ci, co) if flip else (co, ci)
    rows, kp = L.wrows(r_s), L.kpad(1, 1, k_s)
    amax = torch.zeros(64 * 32, device="cuda")
    amax[:] = float(w.abs().max())
    aligned = w.cuda().contiguous()
    buf = torch.zeros(w.numel() + 4, device="cuda")
    shifted = buf[1:1 + w.numel()]
    shifted.copy_(aligned.view(-1))
    assert aligned.data_ptr() % 16 == 0 and shifted.data_ptr() % 16 == 4
    outs = []
    for src in (aligned, shifted):
        u = torch.full((36 * rows * kp * 2,), 0x5a5a, dtype=torch.int16, device="cuda")
        L.call("wino43_weights", src, u, co, ci, flip, split, amax)
        torch.cuda.synchronize()
        outs.append(u)
    assert torch.equal(outs[0], outs[1])
    used = 36 * rows * kp * {0: 2, 2: 2, 3: 1, 4: 1}[split]
    assert bool((outs[0][:used] != 0x5a5a).any()) and bool((outs[0][used:] == 0x5a5a).all())


# ------------------------------------------------------------------------------------ 16-bit storage mode (packed one-term)
def _pack1_rows(x, scale):
    """fp32 [rows][K] -> packed one-term fp16 [K/32][rows][32] (int16 view) of scale * x: the image dsee_wino43_input_f16p /
    dsee_wino43_weights(split = 4) write (64-byte rows of 32 k's)."""
    rows, k = x.shape
    return (x * scale).half().view(rows, k // 32, 32).permute(1, 0, 2).contiguous().view(torch.int16)


def _unpack1_rows(img, rows, k):
    return img.view(torch.float16).view(k // 32, rows, 32).permute(1, 0, 2).reshape(rows, k).float()


@pytest.mark.parametrize("kernel", ["w8", "w4"])
@pytest.mark.parametrize("groups,tg,n,k,bound", [(3, 256, 256, 160, 1.0), (2, 512, 512, 512, 100.0), (36, 256, 256, 32, 8.0),
                                                 (1, 8192, 128, 1024, 225.0), (4, 256, 384, 96, 100.0), (40, 512, 512, 128, 100.0)])
def test_gemm_f16p_pre_packed_one_term(groups, tg, n, k, bound, kernel):
    """dsee_gemm_f16p_pre (16-bit storage mode): both operands ONE scaled fp16 term per element in the packed image, one
    MFMA product per multiply-add, product written as scaled fp16 with its inverse scale in *cscale.  Against float64 on the
    SAME rounded operands the only errors are the fp32 accumulation and the fp16 rounding of the result (2^-11); against the
    unrounded operands the fp16 operand rounding shows (~3e-4).  NaN-poisoned LDS, bit-identical launches."""
    from deepsee_amd import lib as L
    if kernel == "w4" and (n % 256 or k % 64):
        pytest.skip("gemm_w4: whole 256-column tiles and an even number of 32-k slabs")
    entry = "gemm_f16p_pre_w4" if kernel == "w4" else "gemm_f16p_pre"
    g = torch.Generator().manual_seed(groups * 100 + k)
    a = torch.randn(groups * tg, k, generator=g)
    b = torch.randn(groups, n, k, generator=g)
    am_a, am_b = float(a.abs().max()), float(b.abs().max())
    sa, sb = _pow2_scale(bound * am_a), _pow2_scale(am_b)
    a1 = _pack1_rows(a, sa).cuda()
    b1 = torch.stack([_pack1_rows(b[i], sb) for i in range(groups)]).cuda()
    ar = (a * sa).half().double() / sa
    br = (b * sb).half().double() / sb
    ref_r = torch.einsum("gtk,gnk->gtn", ar.view(groups, tg, k), br).reshape(groups * tg, n)
    ref = torch.einsum("gtk,gnk->gtn", a.view(groups, tg, k).double(), b.double()).reshape(groups * tg, n)
    sink = torch.zeros(1, device="cuda")
    first = None
    for it in range(3):
        L.call("selftest_lds_poison", sink)
        c = torch.full((groups * tg, n), float("nan"), dtype=torch.float16, device="cuda")
        cs = torch.zeros(64 * 32, device="cuda")
        L.call(entry, a1, b1, c, groups * tg, n, k, tg, n, _amax(am_a), float(bound), _amax(am_b), cs)
        torch.cuda.synchronize()
        assert torch.isfinite(c).all()
        got = c.float() * cs[0]
        first = got.clone() if first is None else first
        assert torch.equal(got, first)
    assert float(c.float().abs().max()) < 32768.0          # the a-priori output scale keeps the fp16 product in range
    e_r = ((first.cpu().double() - ref_r).norm() / ref_r.norm()).item()
    e = ((first.cpu().double() - ref).norm() / ref.norm()).item()
    print("packed one-term NT, bound %g: vs f64 on the rounded operands %.2e, on the fp32 operands %.2e" % (bound, e_r, e))
    assert e_r < 4e-4 and e < 1e-3      # 2^-11 = 4.9e-4 per rounding; rms over a GEMM is ~ 1/sqrt(3) of that


@pytest.mark.parametrize("groups,t,rp,rq,splits", [(2, 1024, 256, 256, 2), (3, 512, 256, 160, 1), (36, 256, 512, 512, 1),
                                                   (4, 512, 256, 128, 2), (8, 256, 1024, 160, 1)])
def test_gemm_f16p_tn_packed_one_term(groups, t, rp, rq, splits):
    """dsee_gemm_f16p_tn_pqpre: the weight-gradient TN product on packed one-term P (A dY A^T) and Q (B^T d B); a 32-column
    MFMA tile is ONE 64-byte-row slab and a product is one MFMA.  fp32 output: exact (to fp32 accumulation) on the rounded
    operands."""
    from deepsee_amd import lib as L
    g = torch.Generator().manual_seed(5 * t + rq)
    p = torch.randn(groups * t, rp, generator=g) * 3.0
    q = torch.randn(groups * t, rq, generator=g) * 0.02
    am_p, am_q = float(p.abs().max()), float(q.abs().max())
    sp, sq = _pow2_scale(225.0 * am_p / 40.0), _pow2_scale(100.0 * am_q / 12.0)
    p1, q1 = _pack1_rows(p, sp).cuda(), _pack1_rows(q, sq).cuda()
    pr, qr = (p * sp).half().double() / sp, (q * sq).half().double() / sq
    ts = t // splits
    ref_r = torch.einsum("ztp,ztq->zpq", pr.view(groups * splits, ts, rp), qr.view(groups * splits, ts, rq))
    ref = torch.einsum("ztp,ztq->zpq", p.view(groups * splits, ts, rp).double(), q.view(groups * splits, ts, rq).double())
    sink = torch.zeros(1, device="cuda")
    first = None
    for it in range(3):
        L.call("selftest_lds_poison", sink)
        c = torch.full((groups * splits, rp, rq), float("nan"), device="cuda")
        L.call("gemm_f16p_tn_pqpre", p1, q1, c, groups, t, rp, rq, rq, splits, _amax(am_p / 40.0), 225.0, _amax(am_q / 12.0),
               100.0)
        torch.cuda.synchronize()
        assert torch.isfinite(c).all()
        first = c.clone() if first is None else first
        assert torch.equal(c, first)
    e_r = ((first.cpu().double() - ref_r).norm() / ref_r.norm()).item()
    e = ((first.cpu().double() - ref).norm() / ref.norm()).item()
    print("packed one-term TN: vs f64 on the rounded operands %.2e, on the fp32 operands %.2e" % (e_r, e))
    assert e_r < 2e-6 and e < 1.5e-3


@pytest.mark.parametrize("n,h,c", [(2, 32, 128), (8, 16, 512), (1, 64, 256), (2, 32, 160), (1, 32, 96)])
def test_transforms_packed_one_term(n, h, c):
    """dsee_wino43_input_f16p / dsee_wino43_dout_f16p: B^T d B and A dY A^T written as ONE scaled fp16 term per element in
    the packed image equal the fp32 transforms rounded to fp16 (<= 1 ulp: the two kernels may contract different FMAs), in
    both lane mappings (C % 64 == 0: 4 tiles x 64 channels per wave; else 8 tiles x 32 channels); the channel sums that ride
    in the dout pass equal the separate passes."""
    from deepsee_amd import lib as L, ops
    g = torch.Generator().manual_seed(n + h + c)
    x = (torch.randn(n, h, h, c, generator=g) * 0.7).cuda()
    t = n * (h // 4) ** 2
    am = ops.tensor_amax(x)
    for kind, bound in (("input", 100.0), ("dout", ops.DM_BOUND)):
        ref = ops.new(36, t, c)
        L.call("wino43_" + kind, x, ref, n, h, h, c, None)
        img = ops._i16(36 * t * c)
        if kind == "input":
            L.call("wino43_input_f16p", x, img, n, h, h, c, am, bound)
        else:
            ws = ops.scratch(L.lib().dsee_wino43_dout_f16x2_workspace(), "doutsums2")
            if c // 16 <= 64:
                db, d0 = ops.new(c), ops.new(c)
                L.call("wino43_dout_f16p", x, img, n, h, h, c, am, bound, ws, db, d0, 11, 4096, None, 0, 0)
                torch.cuda.synchronize()
                assert rel(db.cpu(), ops.channel_dot(x, None, c).cpu()) < 1e-5
                want = ops.new(c)
                L.call("channel_dot_rng", x, want, n * h * h, c, ops.scratch(L.lib().dsee_channel_dot_workspace(n * h * h, c), "chdot"), 11, 4096)
                torch.cuda.synchronize()
                assert rel(d0.cpu(), want.cpu()) < 1e-5
                only = ops._i16(36 * t * c)
                L.call("wino43_dout_f16p", x, only, n, h, h, c, am, bound, None, None, None, 0, 0, None, 0, 0)
                torch.cuda.synchronize()
                assert torch.equal(only, img)
            else:
                L.call("wino43_dout_f16p", x, img, n, h, h, c, am, bound, None, None, None, 0, 0, None, 0, 0)
        torch.cuda.synchronize()
        sc = _pow2_scale(bound * float(x.abs().max()))
        got = img.view(torch.float16).view(c // 32, 36 * t, 32).permute(1, 0, 2).reshape(36, t, c).float()
        want = (ref * sc).half().float()
        ulp = torch.maximum(want.abs(), torch.tensor(2.0 ** -14, device="cuda")) * 2.0 ** -10
        assert bool(((got - want).abs() <= ulp).all()), (kind, float((got - want).abs().max()))
        assert float((got / sc - ref).abs().max()) <= 2.0 ** -10 * float(ref.abs().max())


@pytest.mark.parametrize("n,cin,cout,h", [(2, 256, 256, 64), (8, 512, 512, 32), (1, 128, 256, 128)])
def test_winograd_conv_16bit_storage_mode(n, cin, cout, h):
    """A whole convolution (forward, data gradient, weight / bias gradients) through the packed one-term chain of the 16-bit
    storage mode -- V, M, dM, dV all 2 bytes per element -- against F.conv2d in float64: per-layer error of the size measured
    for one-term fp16 Winograd operands (0.3-0.4 %), an order of magnitude above nothing else in the chain."""
    from deepsee_amd import ops
    g = torch.Generator().manual_seed(n * cin + h)
    x = torch.randn(n, cin, h, h, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    gy = torch.randn(n, cout, h, h, generator=g)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = F.conv2d(xr, wr, br, padding=1)
    yr.backward(gy.double())
    ops.PROFILE = prof = {}
    with ops.KernelPlan(half=True).active():
        xd = nhwc(x).cuda().requires_grad_()
        ops.tag_amax(xd, ops.tensor_amax(xd.detach()))          # (a producer would have written max |x|)
        wd, bd = w.cuda().requires_grad_(), b.cuda().requires_grad_()
        y = ops.conv2d(xd, wd, bd)
        gyd = nhwc(gy).cuda()
        ops.tag_amax(gyd, ops.tensor_amax(gyd))
        y.backward(gyd)
    torch.cuda.synchronize()
    ops.PROFILE = None
    # forward GEMM + adjoint data-gradient GEMM on the packed one-term kernel, the weight gradient on its TN form
    assert len(prof.get("winograd_gemm_f16_1term_packed", [])) == 2, sorted(prof)
    errs = {"y": rel(nchw(y.detach().cpu(), cout).double(), yr.detach()), "dx": rel(nchw(xd.grad.cpu(), cin).double(), xr.grad),
            "dw": rel(wd.grad.cpu().double(), wr.grad), "db": rel(bd.grad.cpu().double(), br.grad)}
    print(errs)
    assert errs["y"] < 8e-3 and errs["dx"] < 8e-3 and errs["dw"] < 8e-3 and errs["db"] < 1e-5, errs


@pytest.mark.parametrize("m,c,k,ldz", [(2 * 64 * 64, 512, 27, 28), (3 * 17 * 5, 128, 27, 28), (4096 + 3, 640, 32, 32), (77, 256, 5, 8)])
def test_thin_1x1_backward(m, c, k, ldz):
    """dsee_thin1x1_bwd: both gradients of the to-RGB layer's 27-output 1x1 GEMM, laid out along the input channels (exact fp32
    FMAs; per-block partial sums of the weight gradient folded in a fixed order) against float64; ragged pixel counts, K up to 32,
    either output optional; bit-reproducible."""
    from deepsee_amd import lib as L, ops
    g = torch.Generator().manual_seed(m + c + k)
    dz = torch.randn(m, ldz, generator=g)
    w = torch.randn(k, c, generator=g) * 0.1
    x = torch.randn(m, c, generator=g)
    want_dx = dz[:, :k].double() @ w.double()
    want_dw = dz[:, :k].double().t() @ x.double()
    dzd, wd, xd = dz.cuda(), w.cuda(), x.cuda()
    ws = ops.scratch(L.lib().dsee_thin1x1_bwd_workspace(c, k), "wgrad")
    dx, dw = torch.full((m, c), float("nan"), device="cuda"), torch.full((k, c), float("nan"), device="cuda")
    L.call("thin1x1_bwd", dzd, ldz, wd, xd, dx, dw, C.c_long(m), c, k, ws, 0, 0.2, None)
    torch.cuda.synchronize()
    assert rel(dx.cpu().double(), want_dx) < 1e-6
    assert rel(dw.cpu().double(), want_dw) < 2e-6
    dx2, dw2 = torch.empty_like(dx), torch.empty_like(dw)
    L.call("thin1x1_bwd", dzd, ldz, wd, None, dx2, None, C.c_long(m), c, k, None, 0, 0.2, None)
    L.call("thin1x1_bwd", dzd, ldz, None, xd, None, dw2, C.c_long(m), c, k, ws, 0, 0.2, None)
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx) and torch.equal(dw2, dw)
    # in_lrelu: x is the LeakyReLU output of a producer that left its activation's backward to this kernel
    dx3, am = torch.empty_like(dx), ops.amax_slot()
    L.call("thin1x1_bwd", dzd, ldz, wd, xd, dx3, None, C.c_long(m), c, k, None, 1, 0.2, am)
    torch.cuda.synchronize()
    want3 = dx * torch.where(xd > 0, 1.0, 0.2)
    assert torch.equal(dx3, want3) and float(am.max()) == float(want3.abs().max())
