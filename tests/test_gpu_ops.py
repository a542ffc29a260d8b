"""Layer-level parity of the HIP ops (through the C ABI) against torch-CPU fp32 restatements / the oracle,
forward and backward, on the same seeded inputs."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import deepsee_oracle as O

pytestmark = pytest.mark.gpu

TOL = 2e-5


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-20))


def nhwc(x):
    from deepsee_amd import ops
    return ops.to_nhwc(x.cuda())


def nchw(x, c):
    from deepsee_amd import ops
    return ops.to_nchw(x.contiguous(), c).cpu()


def gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("kind,C,R,N", [("spade", 8, 8, 2), ("sean", 64, 16, 2), ("sean", 128, 8, 3), ("puresean", 32, 8, 2)])
def test_spade_sean_norm_fwd_bwd(kind, C, R, N):
    from deepsee_amd import ops, lib as L
    g = gen(C * R + N)
    Lc, S, H = 19, 128, 32
    label = torch.randint(0, Lc, (N, 1, H, H), generator=g).float()
    seg = O.onehot_labels(label, Lc)
    style = (torch.rand(N, Lc, S, generator=g) * 2 - 1).requires_grad_()
    x = torch.randn(N, C, R, R, generator=g).requires_grad_()
    spec = {"n.param_free_norm.running_mean": (C,), "n.param_free_norm.running_var": (C,),
            "n.param_free_norm.num_batches_tracked": (), "n.mlp_shared.0.weight": (128, Lc, 3, 3),
            "n.mlp_shared.0.bias": (128,)}
    if kind in ("spade", "sean"):
        for q in ("mlp_gamma", "mlp_beta"):
            spec["n.%s.weight" % q] = (C, 128, 3, 3)
            spec["n.%s.bias" % q] = (C,)
    if kind in ("sean", "puresean"):
        for q in ("mlp_style_gamma", "mlp_style_beta"):
            spec["n.%s.weight" % q] = (C, S, 3, 3)
            spec["n.%s.bias" % q] = (C,)
    if kind == "sean":
        spec["n.alpha_beta"] = (1,)
        spec["n.alpha_gamma"] = (1,)
    st = {k: O.recipe_tensor("t_" + kind, k, s, 1.0) for k, s in spec.items()}
    orc = O.Oracle(O.make_opt(), {"SR": st})
    P = orc.S["SR"]
    y = F.leaky_relu(orc._norm(kind, P, "n", x, seg, style), 0.2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)

    # ---- HIP
    dev = "cuda"
    labels = ops.Labels(ops.label_to_u8(label.cuda()), Lc)
    shift = labels.shift_for(R)
    p = {k: v.detach().clone().to(dev).requires_grad_(v.requires_grad) for k, v in st.items() if v.is_floating_point()}
    for k in p:
        if not O.is_buffer(k):
            p[k].requires_grad_(True)
    xs = nhwc(x.detach()).requires_grad_()
    sty = style.detach().to(dev).requires_grad_()
    rm = st["n.param_free_norm.running_mean"].clone().to(dev)
    rv = st["n.param_free_norm.running_var"].clone().to(dev)
    want_actv = kind != "puresean"
    want_style = kind != "spade"
    cat = ops.SeanInput.apply(p["n.mlp_shared.0.weight"], p["n.mlp_shared.0.bias"], sty if want_style else None,
                              labels, shift, want_actv, want_style)
    if kind == "spade":
        w2, b2 = ops.pack_gamma_beta(p["n.mlp_gamma.weight"], p["n.mlp_beta.weight"], p["n.mlp_gamma.bias"],
                                     p["n.mlp_beta.bias"])
        add_one = 1.0
    elif kind == "puresean":
        w2, b2 = ops.pack_gamma_beta(p["n.mlp_style_gamma.weight"], p["n.mlp_style_beta.weight"],
                                     p["n.mlp_style_gamma.bias"], p["n.mlp_style_beta.bias"])
        add_one = 0.0
    else:
        wg, wb = torch.sigmoid(p["n.alpha_gamma"]), torch.sigmoid(p["n.alpha_beta"])
        Wg = torch.cat([(1 - wg) * p["n.mlp_gamma.weight"], wg * p["n.mlp_style_gamma.weight"]], 1)
        Wb = torch.cat([(1 - wb) * p["n.mlp_beta.weight"], wb * p["n.mlp_style_beta.weight"]], 1)
        bg = (1 - wg) * p["n.mlp_gamma.bias"] + wg * p["n.mlp_style_gamma.bias"]
        bb = (1 - wb) * p["n.mlp_beta.bias"] + wb * p["n.mlp_style_beta.bias"]
        w2, b2 = ops.pack_gamma_beta(Wg, Wb, bg, bb)
        add_one = 1.0
    h = ops.SpadeNormAct.apply(xs, cat, w2, b2, rm, rv, True, add_one, 0)
    h.backward(nhwc(gy))
    torch.cuda.synchronize()
    assert rel(nchw(h.detach(), C), y.detach()) < TOL
    assert rel(nchw(xs.grad, C), x.grad) < 5 * TOL
    assert rel(rm.cpu(), P["n.param_free_norm.running_mean"]) < TOL
    assert rel(rv.cpu(), P["n.param_free_norm.running_var"]) < TOL
    if want_style:
        assert rel(sty.grad.cpu(), style.grad) < 5 * TOL
    for k, v in P.items():
        if v.requires_grad and v.grad is not None and k in p:
            assert rel(p[k].grad.cpu(), v.grad) < 1e-4, k


@pytest.mark.parametrize("kind,C,R,N,max_fm", [
    ("spade", 64, 16, 2, 256), ("sean", 64, 16, 2, 256), ("sean", 128, 32, 3, 256), ("puresean", 64, 16, 2, 256),
    ("spade", 64, 64, 2, 256), ("sean", 64, 64, 2, 256), ("sean", 128, 64, 3, 256), ("puresean", 64, 64, 2, 256),
    # a channel count that is not a power of two (ngf = 12 / 24 / 48): the fused forward then writes no sign mask (its index
    # is a shift) and the backward pass reads `out`
    ("sean", 192, 64, 2, 256),
    # the reference's max_fm_size cap (normalization.py:188-190, 275-277): embedding at 32^2 / 16^2, upsampled, style ignored
    ("puresean", 128, 64, 2, 32), ("sean", 128, 64, 2, 16), ("puresean", 64, 16, 2, 8)])
def test_sean_norm_table_path(kind, C, R, N, max_fm):
    """The production path for R >= 16: style half as per-image one-hot tables (K = 1440 instead of 2304).  From R = 64
    the gamma/beta GEMM, its data gradient and its weight/table gradient run in the Winograd F(4x4,3x3) domain with
    (transform position, image) groups: ~10x the fp32 rounding error of the direct form, and the ReLU / LeakyReLU
    masks taken from values within that error of zero flip (a sqrt(fraction) effect on the gradients)."""
    from deepsee_amd import ops, networks as Nw
    from types import SimpleNamespace
    g = gen(7 * C + R + N)
    Lc, S, H = 19, 128, 64
    label = torch.randint(0, Lc, (N, 1, H, H), generator=g).float()
    seg = O.onehot_labels(label, Lc)
    style = (torch.rand(N, Lc, S, generator=g) * 2 - 1).requires_grad_()
    x = torch.randn(N, C, R, R, generator=g).requires_grad_()
    mod = Nw.SpadeNorm(kind, C, Lc, S, max_fm)
    st = {"n." + k: O.recipe_tensor("tt_" + kind, k, v.shape, 1.0) for k, v in mod.state_dict().items()}
    orc = O.Oracle(O.make_opt(max_fm_size=max_fm), {"SR": st})
    P = orc.S["SR"]
    y = F.leaky_relu(orc._norm(kind, P, "n", x, seg, style), 0.2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    mod.load_state_dict({k[2:]: v for k, v in st.items()})
    mod.cuda()
    labels = ops.Labels(ops.label_to_u8(label.cuda()), Lc)
    xs = nhwc(x.detach()).requires_grad_()
    sty = style.detach().cuda().requires_grad_()
    capped = kind != "spade" and max_fm < R
    wino = ops._wino_mod_chunk(N, R, R, C, 2 * C, kind != "spade" and not capped) is not None
    assert wino == (R >= 64)
    h = mod(xs, labels, sty, True)
    h.backward(nhwc(gy))
    torch.cuda.synchronize()
    ft, gt = (1e-4, 5e-3) if wino else (TOL, 2e-4)
    assert rel(nchw(h.detach(), C), y.detach()) < ft
    assert rel(nchw(xs.grad, C), x.grad) < (gt if wino else 5 * TOL)
    assert rel(mod.param_free_norm.running_var.cpu(), P["n.param_free_norm.running_var"]) < TOL
    if kind != "spade" and not capped:
        assert rel(sty.grad.cpu(), style.grad) < (gt if wino else 1e-4)
    if capped:                         # the capped path ignores the style matrix, like the reference
        assert sty.grad is None or float(sty.grad.abs().max()) == 0.0
        assert style.grad is None or float(style.grad.abs().max()) == 0.0
    for k, p in mod.named_parameters():
        ref = P["n." + k].grad
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert rel(p.grad.cpu(), ref) < gt, k


@pytest.mark.parametrize("act", [1, 3])
def test_instnorm_act(act):
    from deepsee_amd import ops
    g = gen(act)
    x = torch.randn(3, 24, 9, 11, generator=g).requires_grad_()
    f = (lambda t: F.leaky_relu(t, 0.2)) if act == 1 else torch.tanh
    y = f(O.instance_norm(x))
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xs = nhwc(x.detach()).requires_grad_()
    ys = ops.InstNormAct.apply(xs, act)
    seen = []
    xs.register_hook(seen.append)        # (the gradient tensor as the backward node hands it on, before AccumulateGrad copies it)
    ys.backward(nhwc(gy))
    assert rel(nchw(ys.detach(), 24), y.detach()) < TOL
    assert rel(nchw(xs.grad, 24), x.grad) < 5 * TOL
    # round 6: the kernels write the operand bound of the direct layer that consumes their output in the same pass
    assert ops.DEFAULT_PLAN.conv_amax_out and ops.carried_amax(ys) is not None
    assert float(ys.dsee_amax.max()) == float(ys.detach().abs().max())
    assert ops.carried_amax(seen[0]) is not None and float(seen[0].dsee_amax.max()) == float(xs.grad.abs().max())
    yp = ops.AvgPool3s2.apply(ys)
    assert ops.carried_amax(yp) is ys.dsee_amax and float(yp.detach().abs().max()) <= float(ys.dsee_amax.max())


def test_upsample_noise_and_sumpool():
    from deepsee_amd import ops
    g = gen(5)
    x = torch.randn(2, 16, 6, 6, generator=g).requires_grad_()
    w = torch.randn(16, generator=g).requires_grad_()
    eps = torch.randn(2, 16, 12, 12, generator=g)
    y = F.interpolate(x, scale_factor=2, mode="nearest") + w[None, :, None, None] * eps
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xs, ws = nhwc(x.detach()).requires_grad_(), w.detach().cuda().requires_grad_()
    ys = ops.UpNoise.apply(xs, ws, nhwc(eps), 1)
    ys.backward(nhwc(gy))
    assert rel(nchw(ys.detach(), 16), y.detach()) < TOL
    assert rel(nchw(xs.grad, 16), x.grad) < TOL
    assert rel(ws.grad.cpu(), w.grad) < TOL
    # production form: eps is the Philox stream regenerated in registers -- must equal the materialised tensor
    tok = ops.PhiloxNormal((2, 12, 12, 16), seed=77, offset=1234)
    xa, wa = xs.detach().clone().requires_grad_(), ws.detach().clone().requires_grad_()
    xb, wb = xs.detach().clone().requires_grad_(), ws.detach().clone().requires_grad_()
    ya = ops.UpNoise.apply(xa, wa, tok, 1)
    yb = ops.UpNoise.apply(xb, wb, tok.materialize(), 1)
    ya.backward(nhwc(gy))
    yb.backward(nhwc(gy))
    assert torch.equal(ya, yb) and torch.equal(xa.grad, xb.grad)
    assert rel(wa.grad.cpu(), wb.grad.cpu()) < 1e-6
    e = tok.materialize()
    assert abs(float(e.mean())) < 0.05 and abs(float(e.std()) - 1.0) < 0.05


def test_spectral_norm_fwd_bwd_and_buffers():
    from deepsee_amd import ops
    g = gen(7)
    st = {"c.weight_orig": torch.randn(48, 20, 3, 3, generator=g).requires_grad_(),
          "c.weight_u": F.normalize(torch.randn(48, generator=g), dim=0),
          "c.weight_v": F.normalize(torch.randn(180, generator=g), dim=0)}
    w_d = st["c.weight_orig"].detach().cuda().requires_grad_()
    u_d, v_d = st["c.weight_u"].clone().cuda(), st["c.weight_v"].clone().cuda()
    for power in (True, True, False):
        w = O.spectral_weight(st, "c", power)
        gw = torch.randn(w.shape, generator=g)
        st["c.weight_orig"].grad = None
        w.backward(gw)
        w_d.grad = None
        wd = ops.SpectralNorm.apply(w_d, u_d, v_d, power)
        wd.backward(gw.cuda())
        assert rel(wd.detach().cpu(), w.detach()) < TOL
        assert rel(u_d.cpu(), st["c.weight_u"]) < TOL and rel(v_d.cpu(), st["c.weight_v"]) < TOL
        assert rel(w_d.grad.cpu(), st["c.weight_orig"].grad) < 5 * TOL


def test_style_pool_fwd_bwd():
    from deepsee_amd import ops
    g = gen(9)
    N, C, Hf, H = 2, 128, 16, 32
    label = torch.randint(0, 19, (N, 1, H, H), generator=g).float()
    seg = O.onehot_labels(label, 19)
    f = torch.randn(N, C, Hf, Hf, generator=g).requires_grad_()
    s = O.style_pool(f, seg)
    gs = torch.randn(s.shape, generator=g)
    s.backward(gs)
    labels = ops.Labels(ops.label_to_u8(label.cuda()), 19)
    fs = nhwc(f.detach()).requires_grad_()
    sd = ops.StylePool.apply(fs, labels, labels.shift_for(Hf))
    sd.backward(gs.cuda())
    assert rel(sd.detach().cpu(), s.detach()) < TOL
    assert rel(nchw(fs.grad, C), f.grad) < TOL


def test_pools():
    from deepsee_amd import ops
    g = gen(11)
    x = torch.randn(2, 24, 17, 17, generator=g).requires_grad_()
    y = F.avg_pool2d(x, 3, 2, [1, 1], count_include_pad=False)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xs = nhwc(x.detach()).requires_grad_()
    ys = ops.AvgPool3s2.apply(xs)
    ys.backward(nhwc(gy))
    assert rel(nchw(ys.detach(), 24), y.detach()) < TOL
    assert rel(nchw(xs.grad, 24), x.grad) < TOL
    x = F.relu(torch.randn(2, 8, 12, 12, generator=g)).requires_grad_()  # ties at 0 like VGG after ReLU
    y = F.max_pool2d(x, 2, 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xs = nhwc(x.detach()).requires_grad_()
    ys = ops.MaxPool2.apply(xs)
    ys.backward(nhwc(gy))
    assert rel(nchw(ys.detach(), 8), y.detach()) == 0.0
    assert rel(nchw(xs.grad, 8), x.grad) == 0.0


def test_preprocess_bicubic_labels_dinput():
    from deepsee_amd import ops
    g = gen(13)
    img = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    lr = O.bicubic_down(img, 8)
    lr_d = ops.bicubic_down(nhwc(img), 8)
    assert rel(nchw(lr_d, 3), lr) < 1e-6
    label = torch.randint(0, 19, (2, 1, 64, 64), generator=g).float()
    labels = ops.Labels(ops.label_to_u8(label.cuda()), 19)
    assert torch.equal(labels.t.cpu().long(), label[:, 0].long())
    seg = O.onehot_labels(label, 19)
    fake = torch.randn(2, 3, 64, 64, generator=g)
    want = torch.cat([torch.cat([seg, fake], 1), torch.cat([seg, img], 1)], 0)
    fk = nhwc(fake).requires_grad_()
    din = ops.DInput.apply(labels, fk, nhwc(img))
    assert rel(nchw(din.detach(), 22), want) == 0.0
    assert float(din[..., 22:].abs().max()) == 0.0
    assert float(din.dsee_amax.max()) == float(din.detach().abs().max())       # (bound of D's first layer, written with the tensor)
    gd = torch.randn(4, 22, 64, 64, generator=g)
    din.backward(nhwc(gd))
    assert rel(nchw(fk.grad, 3), gd[:2, 19:22]) == 0.0


def test_losses():
    from deepsee_amd import ops
    g = gen(15)
    a = torch.randn(4, 1, 9, 9, generator=g).requires_grad_()
    ad = nhwc(a.detach()).requires_grad_()
    want = O.Oracle.hinge([[a[:2]]], True, False)
    got = ops.mean_loss(ad, None, ops.MODE_NEG, 1.0, valid_c=1, lo=0, hi=2)
    assert abs(float(got) - float(want)) < 1e-6
    for mode, real in ((ops.MODE_HINGE_REAL, True), (ops.MODE_HINGE_FAKE, False)):
        a.grad = None
        ad.grad = None
        want = O.Oracle.hinge([[a[2:]]], real, True)
        want.backward()
        got = ops.mean_loss(ad, None, mode, 1.0, valid_c=1, lo=2, hi=4)
        got.backward()
        assert abs(float(got) - float(want)) < 1e-6
        assert rel(nchw(ad.grad, 1), a.grad) < 1e-6
    x = torch.randn(2, 32, 5, 5, generator=g).requires_grad_()
    y = torch.randn(2, 32, 5, 5, generator=g)
    want = F.l1_loss(x, y) * 2.5
    want.backward()
    xd = nhwc(x.detach()).requires_grad_()
    got = ops.mean_loss(xd, nhwc(y), ops.MODE_L1, 2.5)
    got.backward()
    assert abs(float(got) - float(want)) < 1e-5
    assert rel(nchw(xd.grad, 32), x.grad) < 1e-6


def test_rng_statistics():
    from deepsee_amd import ops
    z = ops.rng_fill((1 << 20,), 1234, 0, normal=True)
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3
    u = ops.rng_fill((1 << 20,), 1234, 1 << 20, normal=False)
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 2e-3
    z2 = ops.rng_fill((1 << 20,), 1234, 0, normal=True)
    assert torch.equal(z, z2)
    # higher moments, lag correlations within and across the float4 counters, independence of neighbouring seeds / offsets on
    # 4 M draws (round 4 also tried Philox4x32-7, the smallest Crush-resistant round count: no measurable gain, kept at 10)
    z = ops.rng_fill((1 << 22,), 99, 12345, normal=True).double()
    n = z.numel()
    assert abs(float((z ** 3).mean())) < 8e-3 and abs(float((z ** 4).mean()) - 3.0) < 2e-2      # skewness 0, kurtosis 3
    for lag in (1, 2, 3, 4, 5, 8, 64, 1 << 10):
        assert abs(float((z[:-lag] * z[lag:]).mean())) < 3e-3, lag                                # sigma = 1 / sqrt(n) = 5e-4
    z4 = z.view(-1, 4)
    assert float((z4.T @ z4 / z4.shape[0] - torch.eye(4, dtype=torch.float64, device=z4.device)).abs().max()) < 4e-3   # the 4 lanes of a counter
    other = ops.rng_fill((1 << 22,), 100, 12345, normal=True).double()
    shifted = ops.rng_fill((1 << 22,), 99, 12345 + (1 << 20), normal=True).double()
    assert abs(float((z * other).mean())) < 3e-3 and abs(float((z * shifted).mean())) < 3e-3
    a = ops.rng_fill((64,), 7, 10, normal=True)
    b = ops.rng_fill((64,), 7, 11, normal=True)
    assert torch.equal(a[4:], b[:-4])                                           # stream (seed, offset + 1) = the same stream shifted
    u = ops.rng_fill((1 << 22,), 5, 0, normal=False).double()
    assert abs(float(u.var()) - 1.0 / 12) < 5e-4 and abs(float((u[:-1] * u[1:]).mean()) - 0.25) < 1e-3
    hist = torch.histc(u.float(), bins=64, min=0.0, max=1.0)
    assert float((hist - n / 64).abs().max()) < 6 * (n / 64) ** 0.5             # 64 equiprobable bins within 6 sigma


def test_loss_backward_honours_the_upstream_gradient():
    """MeanLoss.backward scales with the upstream gradient (loss scaling, 1/k accumulation, re-weighted terms): the
    gradient of 0.25 * L1 + 3 * hinge equals torch autograd's."""
    from deepsee_amd import ops
    g = gen(21)
    x = torch.randn(2, 32, 5, 5, generator=g).requires_grad_()
    y = torch.randn(2, 32, 5, 5, generator=g)
    a = torch.randn(4, 1, 7, 7, generator=g).requires_grad_()
    want = 0.25 * F.l1_loss(x, y) * 2.5 + 3.0 * O.Oracle.hinge([[a[2:]]], True, True)
    want.backward()
    xd, ad = nhwc(x.detach()).requires_grad_(), nhwc(a.detach()).requires_grad_()
    got = (0.25 * ops.mean_loss(xd, nhwc(y), ops.MODE_L1, 2.5)
           + 3.0 * ops.mean_loss(ad, None, ops.MODE_HINGE_REAL, 1.0, valid_c=1, lo=2, hi=4))
    got.backward()
    assert abs(float(got) - float(want)) < 1e-5
    assert rel(nchw(xd.grad, 32), x.grad) < 1e-6 and rel(nchw(ad.grad, 1), a.grad) < 1e-6


def test_flat_adam_matches_torch_adam_incl_skipped_tensors_and_lr_change():
    """FlatAdam (device-resident descriptors, dsee_adam_step_range) vs torch.optim.Adam over 3 steps: a tensor without
    gradient in one step is skipped (its step count does not advance), a param-group lr change reaches the device, clip
    acts like clip_grad_value_."""
    from deepsee_amd.optim import FlatAdam
    sizes = [(3000,), (5,), (32, 32), (2500,), (7,), (1100,)]
    g = gen(5)
    init = [torch.randn(s, generator=g) for s in sizes]
    params = [torch.nn.Parameter(t.clone().cuda()) for t in init]
    named = [("p%d" % i, p) for i, p in enumerate(params)]
    opt = FlatAdam([{"params": named[:4], "lr": 1e-2}, {"params": named[4:], "lr": 2.5e-3}], betas=(0.5, 0.9))
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    ropt = torch.optim.Adam([{"params": ref[:4], "lr": 1e-2}, {"params": ref[4:], "lr": 2.5e-3}], betas=(0.5, 0.9))
    for it in range(3):
        coef = [torch.randn(s, generator=gen(100 * it + i)) for i, s in enumerate(sizes)]
        skip = {4} if it == 1 else set()
        opt.zero_grad()
        sum((p * coef[i].cuda()).sum() for i, p in enumerate(params) if i not in skip).backward()
        if it == 2:
            opt.param_groups[0]["lr"] = ropt.param_groups[0]["lr"] = 5e-3
        # ranges instead of one launch: same result by construction, exercises first_block > 0
        for b0, b1, _, _ in opt.chunk_ranges(2000):
            pass
        opt.step(clip=0.8)
        ropt.zero_grad()
        for i, p in enumerate(ref):
            p.grad = None if i in skip else coef[i].clone().clamp(-0.8, 0.8)
        ropt.step()
    torch.cuda.synchronize()
    for p, r in zip(params, ref):
        assert float((p.detach().cpu() - r.detach()).abs().max()) < 2e-6
    assert opt.steps().tolist() == [3, 3, 3, 3, 2, 3]


def test_syncbn_kernels_two_shards_match_reference_dp_branch():
    """SURVEY 8 f4 on the device: dsee_norm_stats_local on two shards -> rows stacked as the all-gather would ->
    dsee_norm_stats_merge == the oracle's restatement of the reference's DataParallel branch (pinned by
    tests/golden/host_logic.json) on the concatenated batch, incl. clamp(var, eps) on a constant channel and the
    running statistics; the split backward (reduce -> summed sums -> apply) == autograd through that branch."""
    import ctypes as C
    from deepsee_amd import ops, lib as L
    g = gen(31)
    n, c, r = 4, 64, 16
    x = torch.randn(n, c, r, r, generator=g) * (torch.rand(1, c, 1, 1, generator=g) * 3) + torch.randn(1, c, 1, 1, generator=g)
    x[:, 5] = 0.25
    shards = [x[:2], x[2:]]
    # the oracle in float64: the reference's `ssum - sum * mean` (batchnorm.py:131-133) cancels badly in fp32 on
    # low-variance channels (3e-4 on inv_std here); the kernels' Chan merge does not
    xs = [s.double().requires_grad_(True) for s in shards]
    rm0, rv0 = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    omean, oinv, orm, orv, outs = O.sync_bn_master(xs, rm0.double(), rv0.double())
    rows = []
    for s in shards:
        sd = nhwc(s)
        local = ops.new(2, c)
        ws = ops.scratch(L.lib().dsee_norm_workspace(2, r * r, c, 1), "norm")
        L.call("norm_stats_local", sd, 2, r * r, c, local, ws)
        rows.append(local)
    rows = torch.stack(rows)
    mean, invstd, rm, rv = ops.new(c), ops.new(c), rm0.clone().cuda(), rv0.clone().cuda()
    L.call("norm_stats_merge", rows, 2, 2 * r * r, c, 1e-5, 0.1, 1, mean, invstd, rm, rv)
    assert rel(mean.cpu(), omean.detach()) < 1e-6 and rel(invstd.cpu(), oinv.detach()) < 1e-5
    assert abs(float(invstd[5]) - 1e-5 ** -0.5) < 1e-1                      # clamp(var, eps), not var + eps
    assert rel(rm.cpu(), orm.detach()) < 1e-6 and rel(rv.cpu(), orv.detach()) < 1e-5
    # the F.batch_norm form on the same rows
    L.call("norm_stats_merge", rows, 2, 2 * r * r, c, 1e-5, 0.1, 0, mean, invstd, None, None)
    flat = x.transpose(0, 1).reshape(c, -1)
    assert rel(invstd.cpu(), (flat.var(1, unbiased=False) + 1e-5) ** -0.5) < 1e-5
    # ---- backward of h = lrelu(xhat * scale + beta) per shard with GLOBAL statistics
    L.call("norm_stats_merge", rows, 2, 2 * r * r, c, 1e-5, 0.1, 1, mean, invstd, None, None)
    scale = [torch.randn(s.shape, generator=g) for s in shards]
    beta = [torch.randn(s.shape, generator=g) for s in shards]
    R = [torch.randn(s.shape, generator=g) for s in shards]
    hs = [F.leaky_relu(o * sc.double() + b.double(), 0.2) for o, sc, b in zip(outs, scale, beta)]
    sum((h * rr.double()).sum() for h, rr in zip(hs, R)).backward()
    sums, keep = [], []
    for s, sc, h, rr in zip(shards, scale, hs, R):
        args = [nhwc(rr), nhwc(h.detach().float()), nhwc(s), nhwc(sc)]
        dgb, sm = ops.new(2, r, r, 2 * c), ops.new(4, c)
        ws = ops.scratch(L.lib().dsee_norm_workspace(2, r * r, c, 1), "norm")
        L.call("modulate_bwd_reduce", *args, mean, invstd, dgb, 2 * c, sm, 2, r * r, c, 0.2, ws)
        sums.append(sm)
        keep.append(args)
    tot = sums[0] + sums[1]                                                  # what parallel.allreduce_sums leaves
    live = [ch for ch in range(c) if ch != 5]
    for i, args in enumerate(keep):
        dx = ops.new(2, r, r, c)
        L.call("modulate_bwd_apply", *args, mean, invstd, tot, None, dx, 2, r * r, c, 1.0 / (4 * r * r), 0.2)
        assert rel(nchw(dx, c)[:, live], xs[i].grad[:, live]) < 2e-5


def test_device_input_pipeline_kernels_bit_exact():
    """SURVEY 8 f3 on the device: uint8 image / label batches -> NHWC RGB0 fp32 + uint8 labels (ToTensor, Normalize,
    flip, 255 -> label_nc) bit-exact against the oracle's restatement, then the manager path end to end."""
    from deepsee_amd import data as D, ops
    from deepsee_amd.managers import BaseManager
    from deepsee_amd.options import make_opt
    opt = make_opt(start_size=4, crop_size=32, load_size=32, batchSize=4)
    ds = D.SyntheticDataset(opt, length=8, seed=2)
    ld = D.DeviceLoader(ds, opt, batch_size=4, shuffle=False)
    host = ld.collate([ds[i] for i in range(4)])
    host["flip"] = torch.tensor([0, 1, 1, 0], dtype=torch.uint8)
    host["label"][0, 0, :3] = 255
    want_img, want_lab = O.device_pipeline_reference(host["image"], host["label"], host["flip"], opt.label_nc)
    dev = D.device_preprocess(opt, host)
    torch.cuda.synchronize()
    assert torch.equal(nchw(dev["image_hr"], 3), want_img)
    assert torch.equal(dev["input_semantics"].t.cpu().float()[:, None], want_lab)
    assert rel(nchw(dev["image_lr"], 3), O.bicubic_down(want_img, 4)) < 1e-6
    mgr = BaseManager(opt, create_model=False)
    again = mgr.preprocess(host, from_dataloader=True)                       # uint8 wire format through the manager
    assert torch.equal(again["image_hr"], dev["image_hr"]) and mgr.preprocess(again, True) is again
    batches = list(ld)                                                       # prefetching iterator: 2 batches of 4
    assert len(batches) == 2 and tuple(batches[1]["image_hr"].shape) == (4, 32, 32, 4)
    torch.cuda.synchronize()


@pytest.mark.parametrize("n,cin,cout,h,k,stride,pad,bias", [
    (2, 22, 32, 17, 4, 2, 2, True),       # the discriminator's first layer (19 + 3 channels stored as 24), odd size
    (2, 32, 64, 19, 4, 2, 2, False),
    (3, 64, 128, 33, 4, 2, 2, False),     # 33 -> 17: the two parity classes have 17 and 16 rows
    (2, 64, 128, 64, 4, 2, 2, False),     # even size
    (2, 256, 1, 10, 4, 1, 2, True),       # the discriminator's last layer: 16-output 1x1 GEMM + 16-point gather
    (3, 256, 1, 35, 4, 1, 2, True)])
def test_discriminator_layer_paths(n, cin, cout, h, k, stride, pad, bias):
    """Round 6: the two special forms of the discriminator's convolutions (discriminator.py:78-96) through ops.conv2d against
    F.conv2d autograd in float64 -- the data gradient of the 4 x 4 / stride-2 layers by output parity (one dense 2 x 2
    convolution with 4 Cin output columns + depth-to-space; KernelPlan.dgrad_s2_parity) and the 1-channel last layer as a
    (16 Cout)-output 1x1 GEMM + gather (dsee_thin_gather_k_fwd / _bwd + dsee_thin1x1_bwd).  Output, dx, dw, db <= 2e-5."""
    from deepsee_amd import ops
    g = gen(n * 1000 + cin + h)
    x = torch.randn(n, cin, h, h, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    x6, w6 = x.double().requires_grad_(), w.double().requires_grad_()
    b6 = b.double().requires_grad_() if bias else None
    y6 = F.conv2d(x6, w6, b6, stride=stride, padding=pad)
    gy = torch.randn(y6.shape, generator=g)
    y6.backward(gy.double())
    xd = nhwc(x).requires_grad_()
    wd = w.cuda().requires_grad_()
    bd = b.cuda().requires_grad_() if bias else None
    ops.PROFILE = prof = {}
    y = ops.conv2d(xd, wd, bd, stride=stride, pad=pad)
    y.backward(nhwc(gy))
    torch.cuda.synchronize()
    ops.PROFILE = None
    assert tuple(y.shape[1:3]) == tuple(y6.shape[2:])
    errs = {"y": rel(nchw(y.detach(), cout), y6.detach()), "dx": rel(nchw(xd.grad, cin), x6.grad), "dw": rel(wd.grad.cpu(), w6.grad)}
    if bias:
        errs["db"] = rel(bd.grad.cpu()[:cout], b6.grad)
    print(errs, sorted(prof))
    assert max(errs.values()) < 2e-5, errs


def test_benchmark_shape_conv_vs_float64():
    """north_star's 1e-3 on outputs AND gradients at the benchmark's layer shape: the 512 -> 512 3x3 convolution at
    128^2 (Winograd F(4x4,3x3) + bf16x3: gemm3a 256x256 tiles forward / data gradient, gemm3t 256x128 tiles weight
    gradient) against float64.  No nonlinearity in the layer: nothing is kink-limited, every tensor must hold 1e-3
    (observed ~1e-5: the F(4x4,3x3) transform rounding)."""
    from deepsee_amd import ops
    n, c, h = 2, 512, 128
    assert ops._wino_ok(n, h, h, c, c, 3, 1, 1, 0) and ops._wgrad_mode(c, c) == 2
    g = gen(77)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    b = torch.randn(c, generator=g)
    gy = torch.randn(n, c, h, h, generator=g)
    xd = nhwc(x).requires_grad_()
    wd, bd = w.cuda().requires_grad_(), b.cuda().requires_grad_()
    yd = ops.conv2d(xd, wd, bd)
    yd.backward(nhwc(gy))
    torch.cuda.synchronize()
    got = [nchw(yd.detach(), c), nchw(xd.grad, c), wd.grad.cpu(), bd.grad.cpu()]
    x6, w6, b6 = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    y6 = F.conv2d(x6, w6, b6, padding=1)
    y6.backward(gy.double())
    want = [y6.detach(), x6.grad, w6.grad, b6.grad]
    errs = [rel(a, r) for a, r in zip(got, want)]
    print("512->512 @128^2 vs float64: y %.1e dx %.1e dw %.1e db %.1e" % tuple(errs))
    assert max(errs) < 1e-3 and errs[0] < 1e-4
    # ---- the same layer in the 16-bit storage mode (round 6, VERDICT r5 #4a): the packed one-term kernels -- forward and adjoint
    # data gradient on dsee_gemm_f16p_pre[_w4], weight gradient on dsee_gemm_f16p_tn_pqpre -- at the benchmark's shape against the
    # SAME float64 pass.  No nonlinearity: nothing is kink-limited, so this bound (not the model-level HALF_GRAD_*) is what a sign
    # error or a wrong scale in one of the 16-bit backward kernels has to get past.  One-term fp16 operands carry 2^-11 per element
    # and F(4x4,3x3) amplifies it ~10x: 3e-3 .. 4e-3 per layer observed, bound 1e-2 (VERDICT: <= 2e-2).
    ops.PROFILE = prof = {}
    with ops.KernelPlan(half=True).active():
        xh = nhwc(x).requires_grad_()
        ops.tag_amax(xh, ops.tensor_amax(xh.detach()))          # (a producer would have written max |x|)
        wh, bh = w.cuda().requires_grad_(), b.cuda().requires_grad_()
        yh = ops.conv2d(xh, wh, bh)
        gyh = nhwc(gy)
        ops.tag_amax(gyh, ops.tensor_amax(gyh))
        yh.backward(gyh)
    torch.cuda.synchronize()
    ops.PROFILE = None
    assert len(prof.get("winograd_gemm_f16_1term_packed", [])) == 2 and prof.get("winograd_wgrad_f16_1term"), sorted(prof)
    got16 = [nchw(yh.detach(), c), nchw(xh.grad, c), wh.grad.cpu(), bh.grad.cpu()]
    e16 = [rel(a, r) for a, r in zip(got16, want)]
    print("... 16-bit storage mode vs float64: y %.1e dx %.1e dw %.1e db %.1e" % tuple(e16))
    assert max(e16[:3]) < 1e-2 and e16[3] < 1e-5, e16


@pytest.mark.parametrize("kind", ["sean", "spade"])
def test_benchmark_shape_norm_vs_float64(kind):
    """The fused SPADE/SEAN normalisation at the benchmark's shape (512 channels, 128^2, N = 4 per-image style tables:
    36 x 4 GEMM groups, 256x160 weight/table-gradient tiles) against the oracle run in FLOAT64.  The only
    discontinuities of the layer are the LeakyReLU on its output and the ReLU of the 128-channel embedding; a value
    within rounding of zero may take either branch in any fp32 implementation, so the float64 reference takes ITS branch
    decisions from the HIP forward output (sign of h) -- under equal decisions every gradient must hold north_star's
    1e-3 (observed ~1e-5)."""
    from deepsee_amd import ops, networks as Nw
    N, C, R, Lc, S, H = 4, 512, 128, 19, 128, 256
    g = gen(5 + len(kind))
    label = F.interpolate(torch.randint(0, Lc, (N, 1, 32, 32), generator=g).float(), size=(H, H), mode="nearest")
    seg = O.onehot_labels(label, Lc).double()
    style = (torch.rand(N, Lc, S, generator=g) * 2 - 1)
    x = torch.randn(N, C, R, R, generator=g)
    gy = torch.randn(N, C, R, R, generator=g)
    mod = Nw.SpadeNorm(kind, C, Lc, S, 256)
    st = {"n." + k: O.recipe_tensor("bs_" + kind, k, v.shape, 1.0) for k, v in mod.state_dict().items()}
    mod.load_state_dict({k[2:]: v for k, v in st.items()})
    mod.cuda()
    labels = ops.Labels(ops.label_to_u8(label.cuda()), Lc)
    assert ops._wino_mod_chunk(N, R, R, C, 2 * C, kind != "spade") == N
    xs = nhwc(x).requires_grad_()
    sty = style.cuda().requires_grad_()
    h = mod(xs, labels, sty, True)
    h.backward(nhwc(gy))
    torch.cuda.synchronize()
    h_hip = nchw(h.detach(), C)
    # ---- float64 oracle with HIP's LeakyReLU decisions
    orc = O.Oracle(O.make_opt(), {"SR": st}, dtype=torch.float64)
    P = orc.S["SR"]
    x6, s6 = x.double().requires_grad_(), style.double().requires_grad_()
    pre = orc._norm(kind, P, "n", x6, seg, s6)
    y6 = torch.where(h_hip > 0, pre, 0.2 * pre)
    y6.backward(gy.double(), retain_graph=True)
    flips = float(((pre.detach() > 0) != (h_hip > 0)).double().mean())
    fwd = rel(h_hip, F.leaky_relu(pre.detach(), 0.2))
    errs = {"dx": rel(nchw(xs.grad, C), x6.grad)}
    if kind != "spade":
        errs["dstyle"] = rel(sty.grad.cpu(), s6.grad)
    for k, p in mod.named_parameters():
        ref = P["n." + k].grad
        if ref is not None:
            errs[k] = rel(p.grad.cpu(), ref)
    worst = max(errs, key=errs.get)
    print("%s norm 512ch @128^2 N=4 vs float64: forward %.1e (%.1e of the outputs on the other LeakyReLU branch), "
          "gradients worst %.1e (%s), dx %.1e" % (kind, fwd, flips, errs[worst], worst, errs["dx"]))
    assert fwd < 1e-4 and flips < 1e-4
    assert errs[worst] < 1e-3, errs
    # ---- the same layer in the 16-bit storage mode (round 6, VERDICT r5 #4a): packed one-term fused forward
    # (dsee_spade_fused_fwd_f16p), packed gamma/beta gradient (dsee_modulate_bwd_reduce_wino_f16p), packed TN table / embedding
    # weight gradients and adjoint GEMM, fp16 `scale` -- every gradient against the same float64 graph, the LeakyReLU decisions
    # again taken from the HIP forward under test (so no kink enters): forward ~3e-4, gradients bounded at 8e-3.
    x6.grad = s6.grad = None
    for prm in P.values():
        if getattr(prm, "grad", None) is not None:
            prm.grad = None
    mod.zero_grad(set_to_none=True)
    with ops.KernelPlan(half=True).active():
        xs16 = nhwc(x).requires_grad_()
        sty16 = style.cuda().requires_grad_()
        h16 = mod(xs16, labels, sty16, True)
        gy16 = nhwc(gy)
        ops.tag_amax(gy16, ops.tensor_amax(gy16))
        h16.backward(gy16)
    torch.cuda.synchronize()
    h16c = nchw(h16.detach(), C)
    y16 = torch.where(h16c > 0, pre, 0.2 * pre)
    y16.backward(gy.double())
    fwd16 = rel(h16c, F.leaky_relu(pre.detach(), 0.2))
    e16 = {"dx": rel(nchw(xs16.grad, C), x6.grad)}
    if kind != "spade":
        e16["dstyle"] = rel(sty16.grad.cpu(), s6.grad)
    for k, p in mod.named_parameters():
        ref = P["n." + k].grad
        if ref is not None:
            e16[k] = rel(p.grad.cpu(), ref)
    w16 = max(e16, key=e16.get)
    print("... 16-bit storage mode: forward %.1e, gradients worst %.1e (%s), dx %.1e" % (fwd16, e16[w16], w16, e16["dx"]))
    assert fwd16 < 2e-3, fwd16            # observed 2.5e-4 (SEAN) / 4.9e-4 (SPADE)
    assert e16[w16] < 8e-3, e16          # observed 2.2e-3 (mlp_shared weight), dx 2.7e-4 / 4.2e-4; VERDICT r5 asked <= 2e-2


def test_conv_noise_fused_in_output_transform():
    """noise_middle (architecture.py:111-112) fused into the Winograd output transform of conv_0: same values as the
    convolution followed by the stand-alone UpNoise pass on the same Philox stream, same gradients for the input, the
    weights and the noise weight."""
    from deepsee_amd import ops
    g = gen(41)
    n, c, h = 2, 128, 32
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    b, nw, gy = torch.randn(c, generator=g), torch.randn(c, generator=g), torch.randn(n, c, h, h, generator=g)
    eps = ops.PhiloxNormal((n, h, h, c), 4242, 1000)
    outs = []
    for fused in (True, False):
        xd = nhwc(x).requires_grad_()
        wd, bd, nd = w.cuda().requires_grad_(), b.cuda().requires_grad_(), nw.cuda().requires_grad_()
        if fused:
            y = ops.conv2d(xd, wd, bd, noise=(nd, eps))
            assert y.grad_fn.__class__.__name__.startswith("Conv2d")       # one node: no separate UpNoise pass
        else:
            y = ops.UpNoise.apply(ops.conv2d(xd, wd, bd), nd, eps, 0)
        y.backward(nhwc(gy))
        outs.append([t.detach().cpu() for t in (y, xd.grad, wd.grad, bd.grad, nd.grad)])
    torch.cuda.synchronize()
    for a, r in zip(*outs):
        assert rel(a, r) < 1e-6
    # and the noise really is N(0,1) * nw on top of the convolution
    y0 = ops.conv2d(nhwc(x), w.cuda(), b.cuda())
    z = (outs[0][0] - y0.cpu()) / nw.view(1, 1, 1, c)
    assert abs(float(z.mean())) < 2e-2 and abs(float(z.std()) - 1.0) < 2e-2


@pytest.mark.parametrize("n,c,h", [(2, 128, 32), (8, 256, 16), (8, 512, 32)])
def test_batchnorm_statistics_from_the_producers(n, c, h):
    """The BatchNorm statistics of a SPADE/SEAN norm whose input comes from UpNoise (norm_0) or from a Winograd convolution
    with fused noise (conv_0 -> norm_1) are written by those kernels (rows of (count, mean, M2) per workgroup, folded by
    dsee_norm_stats_finalize_parts; sync_batchnorm/batchnorm.py:65-68).  Against float64 statistics of the produced tensor
    and against the separate statistics pass: mean / invstd <= 2e-6, identical running-statistics update."""
    from deepsee_amd import ops
    g = gen(n * c + h)
    x = (torch.randn(n, c, h, h, generator=g) * 2 + torch.randn(1, c, 1, 1, generator=g) * 3)
    nw = torch.randn(c, generator=g).cuda()
    w = (torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5).cuda()
    b = torch.randn(c, generator=g).cuda()
    eps1 = ops.PhiloxNormal((n, 2 * h, 2 * h, c), 99, 5000)
    eps2 = ops.PhiloxNormal((n, h, h, c), 99, 777)
    producers = {"up_noise": lambda st: ops.UpNoise.apply(nhwc(x), nw, eps1, 1, st),
                 "wino_output": lambda st: ops.conv2d(nhwc(x), w, b, noise=(nw, eps2), stats=st)}
    for name, make in producers.items():
        y = make(True)
        assert getattr(y, "dsee_stats_rows", None) is not None, name
        plain = make(False)
        assert getattr(plain, "dsee_stats_rows", None) is None and torch.equal(y, plain)   # same values either way
        rm_a, rv_a = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        rm_b, rv_b = rm_a.clone(), rv_a.clone()
        mean_a, inv_a, _ = ops.bn_stats(y, rm_a, rv_a, True)
        mean_b, inv_b, _ = ops.bn_stats(plain, rm_b, rv_b, True)
        torch.cuda.synchronize()
        y64 = y.double().reshape(-1, c)
        mu, var = y64.mean(0), y64.var(0, unbiased=False)
        inv = 1.0 / torch.sqrt(var + ops.BN_EPS)
        for got in ((mean_a, inv_a), (mean_b, inv_b)):
            assert float((got[0].double() - mu).abs().max() / mu.abs().max()) < 2e-6, name
            assert float(((got[1].double() - inv) / inv).abs().max()) < 2e-6, name
        assert float((rm_a - rm_b).abs().max()) < 1e-6 and float(((rv_a - rv_b) / rv_b).abs().max()) < 1e-6, name


@pytest.mark.parametrize("kind,ups", [("sean", 0), ("spade", 1)])
def test_resblock_fused_noise_shortcut_and_gradient_sink(kind, ups):
    """A whole SPADEResnetBlock with the production noise source (Philox draws regenerated in registers: noise_middle in
    conv_0's output transform, the shortcut x + w_skip * eps in conv_1's, the shortcut's gradient folded into norm_0's
    backward through ops.GradSink) against the same block fed the SAME draws as explicit tensors (every fusion off:
    separate UpNoise passes, autograd's own fan-in addition)."""
    from types import SimpleNamespace
    from deepsee_amd import ops, networks as Nw, lib as L
    g = gen(17 + ups)
    n, c, r, Lc = 2, 128, 64, 19
    opt = SimpleNamespace(add_noise=True, semantic_nc=Lc, regional_style_size=128, max_fm_size=256)
    blk = Nw.SPADEResnetBlock(c, opt, kind)
    st = {k: O.recipe_tensor("rb_" + kind, k, v.shape, 1.0) for k, v in blk.state_dict().items()}
    blk.load_state_dict(st)
    blk.cuda()
    label = F.interpolate(torch.randint(0, Lc, (n, 1, 16, 16), generator=g).float(), size=(256, 256), mode="nearest")
    labels = ops.Labels(ops.label_to_u8(label.cuda()), Lc)
    x0 = torch.randn(n, c, r >> ups, r >> ups, generator=g)
    style = torch.rand(n, Lc, 128, generator=g) * 2 - 1
    gy = torch.randn(n, c, r, r, generator=g)

    class Tensors(Nw.DeviceNoise):          # same Philox positions, handed out as materialised tensors
        def normal_nhwc(self, shape, tag):
            return super().normal_nhwc(shape, tag).materialize()

    res = []
    for src in (Nw.DeviceNoise(seed=5), Tensors(seed=5)):
        blk.load_state_dict(st)           # (a forward advances the spectral-norm u / v and the BN running statistics)
        blk.zero_grad()
        xd = nhwc(x0).requires_grad_()
        sd = style.cuda().requires_grad_()
        y = blk(xd, labels, sd, src, "b", ups, True, L.ACT_LRELU)
        y.backward(nhwc(gy))
        res.append([y.detach().cpu(), xd.grad.cpu(), sd.grad.cpu() if sd.grad is not None else torch.zeros(1)] +
                   [p.grad.detach().cpu().clone() for _, p in sorted(blk.named_parameters()) if p.grad is not None])
    torch.cuda.synchronize()
    assert len(res[0]) == len(res[1]) and len(res[0]) > 10
    # (conv_0's bias feeds a BatchNorm: its gradient is zero up to rounding, ~1e-5 -- such tensors are held to the same
    # absolute error as the real gradients, not to a relative one)
    floor = 1e-3 * max(float(t.double().norm()) for t in res[1][3:])
    for a, b in zip(*res):
        err = float((a.double() - b.double()).norm()) / max(float(b.double().norm()), floor)
        assert err < 2e-5, (a.shape, err)
    assert float(res[0][-1].abs().max()) > 0


def test_psnr_ssim_rmse_kernel_matches_reference_numbers():
    """dsee_psnr_ssim against tests/golden/metrics.json -- the reference's own tensor2im / calculate_psnr /
    calculate_ssim outputs -- and against the oracle on a 256x256 pair: PSNR from exact integer sums (1e-12), SSIM in
    float64 (1e-9; the reference's filter2D and the kernel's separable window differ in summation order), RMSE 1e-6;
    identical images give PSNR = inf; values beyond [-1, 1] are clipped by the quantisation as in tensor2im.  Also the
    MetricsEvaluator mirror: buffers, get_result keys, per-sample CSV rows."""
    import json, math, tempfile
    from deepsee_amd import metrics as M, ops
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.json")))["cases"]
    for rec in cases:
        n, h, w = rec["shape"]
        fake, real = O.metric_case_inputs(rec["seed"], n, h, w, rec["kind"])
        got = M.psnr_ssim_rmse(fake.cuda(), real.cuda())
        for i in range(n):
            ref = float(rec["psnr"][i])
            assert (math.isinf(ref) and math.isinf(float(got[i, 0]))) or abs(float(got[i, 0]) - ref) <= 1e-12 * abs(ref), rec["kind"]
            assert abs(float(got[i, 1]) - rec["ssim"][i]) <= 1e-9, (rec["kind"], float(got[i, 1]), rec["ssim"][i])
            assert abs(float(got[i, 2]) - rec["rmse"][i]) <= 1e-6 * rec["rmse"][i] + 1e-12
    g = torch.Generator().manual_seed(9)
    real = torch.nn.functional.interpolate(torch.rand(2, 3, 32, 32, generator=g), (256, 256), mode="bicubic").clamp(0, 1) * 2 - 1
    fake = (real + 0.1 * torch.randn(2, 3, 256, 256, generator=g)).clamp(-1, 1)
    want = O.psnr_ssim_rmse(fake, real)
    got = M.psnr_ssim_rmse(ops.to_nhwc(fake.cuda()), ops.to_nhwc(real.cuda()))        # native layout in, no conversion
    assert float((got[:, 0] - want[:, 0]).abs().max()) <= 1e-11 and float((got[:, 1] - want[:, 1]).abs().max()) <= 1e-9
    assert float((got[:, 2] - want[:, 2]).abs().max()) <= 1e-9
    with tempfile.TemporaryDirectory() as d:
        ev = M.MetricsEvaluator(write_details=True, folder_out=d, extra_columns=("split",), extra_columns_content=("val",))
        ev.collect_samples(fake.cuda(), real.cuda(), name=["a/7394.jpg", "a/12.png"])
        res = ev.get_result()
        assert list(res) == ["psnr/mean", "ssim/mean", "rmse/mean", "psnr/std", "ssim/std", "rmse/std", "n_samples"]
        assert res["n_samples"] == 2 and abs(res["ssim/mean"] - float(want[:, 1].mean())) <= 1e-9
        rows = open(os.path.join(d, "metrics.csv")).read().strip().splitlines()
        assert rows[0] == "split,ID,PSNR,SSIM,RMSE" and rows[1].startswith("val,7394,") and rows[2].startswith("val,12,")


def _capi_comm_child(q):
    import ctypes as C
    from deepsee_amd import lib as L, parallel
    ident = parallel.CapiComm.unique_id()
    assert len(ident) == 128 and any(ident)
    comm = parallel.CapiComm(1, 0, ident)
    assert L.lib().dsee_comm_world(comm.handle) == 1 and L.lib().dsee_comm_rank(comm.handle) == 0
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(100003, device="cuda", generator=g)
    want = x.clone()
    comm.all_reduce_sum_(x)                                   # sum over one rank
    y = torch.arange(77, device="cuda", dtype=torch.int64)
    comm.broadcast_(y, 0)
    rows = comm.all_gather(torch.stack([want[:64], want[64:128]]))
    torch.cuda.synchronize()
    ok = (torch.equal(x, want) and torch.equal(y.cpu(), torch.arange(77)) and tuple(rows.shape) == (1, 2, 64)
          and torch.equal(rows[0, 1], want[64:128]))
    # argument checks come back as error codes with a message, not as crashes
    rc = L.lib().dsee_comm_broadcast(comm.handle, C.c_void_p(x.data_ptr()), 16, 3, None)
    msg = L.lib().dsee_last_error().decode()
    comm.close()
    q.put((ok, rc, msg))


def test_rccl_communicator_behind_the_c_abi_world1():
    """dsee_comm_* (include/deepsee_hip.h): RCCL resolved with dlopen inside libdeepsee_hip.so -- unique id, init on the current
    device, in-place sum all-reduce / broadcast / all-gather on the caller's stream, destroy.  One rank (every collective is the
    identity); the 2-rank exchange is test_two_gpu_rccl_data_parallel's business.  In a child process: its own RCCL bootstrap."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_capi_comm_child, args=(q,))
    p.start()
    try:
        ok, rc, msg = q.get(timeout=240)
    finally:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert ok
    assert rc != 0 and "root" in msg


@pytest.mark.parametrize("kind", ["sean", "spade"])
def test_coarse_sean_norm_fwd_is_the_autograd_path(kind):
    """dsee_sean_norm_fwd (include/deepsee_hip.h, coarse entry points): ONE C call = the whole SPADE / SEAN normalisation
    forward + LeakyReLU (embedding, batch statistics, operand transforms, fused gamma/beta kernel) from raw tensors in a
    caller-owned workspace -- what a host without deepsee_amd/ops.py would call.  Fed with the very tensors
    SeanNormTable.forward receives inside the module's forward, it must reproduce that forward bit for bit (h, the saved
    modulation factor, mean / invstd, the updated running statistics), training and evaluation mode."""
    import ctypes as C
    from deepsee_amd import ops, lib as L, networks as Nw
    N, Cc, R, Lc, S, H = 2, 64, 64, 19, 128, 128      # (per-image tables: whole 128-tile GEMM tiles per image)
    g = gen(77)
    label = F.interpolate(torch.randint(0, Lc, (N, 1, 8, 8), generator=g).float(), size=(H, H), mode="nearest")
    style = (torch.rand(N, Lc, S, generator=g) * 2 - 1).cuda()
    x = nhwc(torch.randn(N, Cc, R, R, generator=g) * 1.5 + 0.3)
    mod = Nw.SpadeNorm(kind, Cc, Lc, S, 256)
    mod.load_state_dict({k: O.recipe_tensor("coarse_" + kind, k, v.shape, 1.0) for k, v in mod.state_dict().items()})
    mod.cuda()
    labels = ops.Labels(ops.label_to_u8(label.cuda()), Lc)
    assert ops._fused_norm_ok(N, R, R, Cc, 2 * Cc, 160 if kind == "sean" else 128)
    for training in (True, False):
        seen = {}
        orig = ops.SeanNormTable._forward

        def spy(ctx, x_, w_sh, b_sh, w2a, table, b2, rm, rv, labels_, shift, training_, add_one, grad_sink=None, cat_ups=0):
            seen.update(w_sh=w_sh, b_sh=b_sh, w2a=w2a.contiguous(), table=None if table is None else table.contiguous(),
                        b2=b2.contiguous(), rm=rm.clone(), rv=rv.clone(), shift=shift, add_one=add_one)
            return orig(ctx, x_, w_sh, b_sh, w2a, table, b2, rm, rv, labels_, shift, training_, add_one, grad_sink, cat_ups)

        call = L.call

        def call_spy(name, *a):
            if name == "spade_fused_fwd":     # (V2, U2, amax_cat, v_bound, amax_u, bias, x, mean, invstd, h, scale, ...)
                seen.update(mean=a[7], invstd=a[8], scale=a[10])
            return call(name, *a)

        ops.SeanNormTable._forward = staticmethod(spy)
        L.call = call_spy
        try:
            h = mod(x.clone().requires_grad_(), labels, style if kind == "sean" else None, training)
        finally:
            ops.SeanNormTable._forward = staticmethod(orig)
            L.call = call
        assert "scale" in seen, "the module's forward did not take the fused kernel"
        st = mod.param_free_norm
        has_t = seen["table"] is not None
        assert has_t == (kind == "sean")
        nbytes = L.lib().dsee_sean_norm_fwd_workspace(N, R, R, Cc, Lc, int(has_t))
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        out, scale = torch.empty_like(x), torch.empty_like(x)
        mean, invstd, hm = ops.new(Cc), ops.new(Cc), torch.zeros(2048, device="cuda")
        rm, rv = seen["rm"].clone(), seen["rv"].clone()
        L.call("sean_norm_fwd", labels.t, labels.h, labels.w, seen["shift"], Lc, seen["w_sh"].contiguous(),
               seen["b_sh"].contiguous(), seen["w2a"], seen["table"], seen["b2"], x, rm, rv, int(training), 1e-5, 0.1,
               float(seen["add_one"]), 0.2, out, scale, None, mean, invstd, hm, N, R, R, Cc, ws, nbytes)
        torch.cuda.synchronize()
        assert torch.equal(out, h.detach()), (kind, training, rel(out.cpu(), h.detach().cpu()))
        assert torch.equal(scale, seen["scale"]) and torch.equal(mean, seen["mean"]) and torch.equal(invstd, seen["invstd"])
        assert torch.equal(rm, st.running_mean) and torch.equal(rv, st.running_var)
        assert float(hm.max()) == float(out.abs().max())
    # the argument checks answer with DSEE_EINVAL and a message instead of launching anything
    with pytest.raises(L.DseeError, match="workspace"):
        L.call("sean_norm_fwd", labels.t, labels.h, labels.w, seen["shift"], Lc, seen["w_sh"].contiguous(),
               seen["b_sh"].contiguous(), seen["w2a"], seen["table"], seen["b2"], x, rm, rv, 1, 1e-5, 0.1, 1.0, 0.2, out, None,
               None, mean, invstd, None, N, R, R, Cc, ws, 1024)


def test_coarse_spade_resblock_fwd_matches_the_module():
    """dsee_spade_resblock_fwd: ONE C call = a whole SPADEResnetBlock forward (two SEAN norms on the fused kernel, two Winograd
    convolutions on pre-split operands, identity shortcut) from raw tensors and a caller-owned workspace -- against the Python
    module's forward (training mode, add_noise off) fed with the same effective weights.  Not bit-identical by construction:
    the module takes norm_1's batch statistics from the rows conv_0's output transform wrote and splits this small layer's V
    inside the GEMM, the C path runs a statistics pass and the pre-split GEMM -- same arithmetic, different summation order:
    <= 1e-5."""
    import ctypes as C
    from deepsee_amd import ops, lib as L, networks as Nw
    from deepsee_amd.options import make_opt
    N, Cc, R, Lc, S, H = 2, 256, 64, 19, 128, 128
    g = gen(91)
    opt = make_opt(ngf=Cc // 16, add_noise=False, start_size=8, crop_size=H, load_size=H, batchSize=N)
    blk = Nw.SPADEResnetBlock(Cc, opt, "sean")
    blk.load_state_dict({k: O.recipe_tensor("coarse_blk", k, v.shape, 1.0) for k, v in blk.state_dict().items()})
    blk.cuda()
    label = F.interpolate(torch.randint(0, Lc, (N, 1, 8, 8), generator=g).float(), size=(H, H), mode="nearest")
    labels = ops.Labels(ops.label_to_u8(label.cuda()), Lc)
    style = (torch.rand(N, Lc, S, generator=g) * 2 - 1).cuda()
    x = nhwc(torch.randn(N, Cc, R, R, generator=g))
    norms, convs = [], []
    fwd, conv2d = ops.SeanNormTable._forward, ops.conv2d

    def norm_spy(ctx, x_, w_sh, b_sh, w2a, table, b2, rm, rv, labels_, shift, training_, add_one, grad_sink=None, cat_ups=0):
        norms.append(dict(w_sh=w_sh.contiguous(), b_sh=b_sh.contiguous(), w2a=w2a.contiguous(), table=table.contiguous(),
                          b2=b2.contiguous(), rm=rm.clone(), rv=rv.clone(), shift=shift, add_one=add_one))
        return fwd(ctx, x_, w_sh, b_sh, w2a, table, b2, rm, rv, labels_, shift, training_, add_one, grad_sink, cat_ups)

    def conv_spy(x_, w, bias=None, **kw):
        convs.append((w.detach().contiguous().clone(), None if bias is None else bias.detach().contiguous().clone()))
        return conv2d(x_, w, bias, **kw)

    ops.SeanNormTable._forward, ops.conv2d = staticmethod(norm_spy), conv_spy
    try:
        with torch.no_grad():
            want = blk(x, labels, style, None, "blk", 0, True, L.ACT_LRELU)
    finally:
        ops.SeanNormTable._forward, ops.conv2d = staticmethod(fwd), conv2d
    assert len(norms) == 2 and len(convs) == 2

    class NormLayer(C.Structure):
        _fields_ = [(k, C.c_void_p) for k in ("w_shared", "b_shared", "w2a", "table", "bias_packed", "running_mean",
                                              "running_var")] + [("add_one", C.c_float)]

    def layer(d):
        return NormLayer(d["w_sh"].data_ptr(), d["b_sh"].data_ptr(), d["w2a"].data_ptr(), d["table"].data_ptr(),
                         d["b2"].data_ptr(), d["rm"].data_ptr(), d["rv"].data_ptr(), float(d["add_one"]))

    nbytes = L.lib().dsee_spade_resblock_fwd_workspace(N, R, R, Cc, Lc, 1)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    out = torch.empty_like(x)
    n0, n1 = layer(norms[0]), layer(norms[1])
    args = (C.byref(n0), convs[0][0], convs[0][1], C.byref(n1), convs[1][0], convs[1][1], labels.t, labels.h, labels.w,
            norms[0]["shift"], Lc, x, out, L.ACT_LRELU, 1, 1e-5, 0.1, 0.2, N, R, R, Cc, ws)
    L.call("spade_resblock_fwd", *args, nbytes)
    torch.cuda.synchronize()
    dev = rel(out.cpu(), want.cpu())
    print("coarse resblock forward vs the module: %.2e" % dev)
    assert dev < 1e-5, dev
    # the running statistics advanced as the module's did
    for d, nm in zip(norms, (blk.norm_0, blk.norm_1)):
        assert rel(d["rm"].cpu(), nm.param_free_norm.running_mean.cpu()) < 1e-5
        assert rel(d["rv"].cpu(), nm.param_free_norm.running_var.cpu()) < 1e-5
    with pytest.raises(L.DseeError, match="256 x 128 tiles"):       # shapes the pre-split GEMM does not tile are refused
        L.call("spade_resblock_fwd", *args[:18], N, 32, 32, Cc, ws, nbytes)


@pytest.mark.parametrize("kind", ["sean", "spade"])
def test_coarse_resblock_training_pair(kind):
    """Round 6 (SURVEY 7 "whole resblock fwd/bwd"): dsee_spade_resblock_train_fwd + dsee_spade_resblock_bwd -- two C calls run the
    hot block of configs[1] as it trains (architecture.py:75-147: two SPADE / SEAN norms on the fused kernel, two Winograd
    convolutions on pre-split operands, noise_middle in conv_0's output transform, the shortcut x + noise_skip(x) in conv_1's)
    forward AND backward, from raw tensors, a caller-owned `saved` area and workspaces -- against the Python module's own
    launches at the benchmark's channel count (512 channels, 64 x 64, N = 8: the shapes at which ops.py takes the pre-split
    path): output, dx and every parameter gradient BIT-IDENTICAL (same kernels, same operands, same order), running statistics
    advanced alike."""
    import ctypes as C
    from deepsee_amd import ops, lib as L, networks as Nw
    from deepsee_amd.options import make_opt
    N, Cc, R, Lc, S, H = 8, 512, 64, 19, 128, 128
    g = gen(17 + len(kind))
    opt = make_opt(ngf=Cc // 16, add_noise=True, start_size=8, crop_size=H, load_size=H, batchSize=N)
    blk = Nw.SPADEResnetBlock(Cc, opt, kind)
    blk.load_state_dict({k: O.recipe_tensor("coarse_train_" + kind, k, v.shape, 1.0) for k, v in blk.state_dict().items()})
    blk.cuda()
    label = F.interpolate(torch.randint(0, Lc, (N, 1, 8, 8), generator=g).float(), size=(H, H), mode="nearest")
    labels = ops.Labels(ops.label_to_u8(label.cuda()), Lc)
    style = (torch.rand(N, Lc, S, generator=g) * 2 - 1).cuda()
    x = nhwc(torch.randn(N, Cc, R, R, generator=g)).requires_grad_()
    gout = nhwc(torch.randn(N, Cc, R, R, generator=g))
    noise = Nw.DeviceNoise(5)
    noise.begin_step()
    shp = (N, R, R, Cc)
    eps_mid, eps_skip = noise.normal_nhwc(shp, "blk.noise_middle"), noise.normal_nhwc(shp, "blk.noise_skip")
    w_mid = torch.randn(Cc, generator=g).cuda().requires_grad_()
    w_skip = torch.randn(Cc, generator=g).cuda().requires_grad_()
    w0 = blk.conv_0.weight(True).detach().clone().requires_grad_()
    w1 = blk.conv_1.weight(True).detach().clone().requires_grad_()
    b0, b1 = blk.conv_0.bias, blk.conv_1.bias
    norms = []
    fwd = ops.SeanNormTable._forward

    def norm_spy(ctx, x_, w_sh, b_sh, w2a, table, b2, rm, rv, labels_, shift, training_, add_one, grad_sink=None, cat_ups=0):
        for t in (w2a, table, b2):
            if t is not None and t.requires_grad:
                t.retain_grad()
        norms.append(dict(w_sh=w_sh, b_sh=b_sh, w2a=w2a, table=table, b2=b2, rm=rm.clone(), rv=rv.clone(), shift=shift,
                          add_one=add_one, mod_rm=rm, mod_rv=rv))
        return fwd(ctx, x_, w_sh, b_sh, w2a, table, b2, rm, rv, labels_, shift, training_, add_one, grad_sink, cat_ups)

    ops.SeanNormTable._forward = staticmethod(norm_spy)
    try:
        # SPADEResnetBlock.forward from the block's (already noised / upsampled) input on, with leaf effective weights
        sink = ops.GradSink()
        h = blk.norm_0(x, labels, style, True, sink)
        dxm = ops.conv2d(h, w0, b0, noise=(w_mid, eps_mid), stats=True)
        h = blk.norm_1(dxm, labels, style, True)
        want = ops.conv2d(h, w1, b1, res=x, act=L.ACT_NONE, res_noise=(w_skip, eps_skip), res_sink=sink)
    finally:
        ops.SeanNormTable._forward = staticmethod(fwd)
    assert len(norms) == 2
    ops.tag_amax(gout, ops.tensor_amax(gout))
    want.backward(gout)
    torch.cuda.synchronize()

    class NormLayer(C.Structure):
        _fields_ = [(k, C.c_void_p) for k in ("w_shared", "b_shared", "w2a", "table", "bias_packed", "running_mean",
                                              "running_var")] + [("add_one", C.c_float)]

    class BlockNoise(C.Structure):
        _fields_ = [("w_middle", C.c_void_p), ("seed_middle", C.c_uint64), ("offset_middle", C.c_uint64),
                    ("w_skip", C.c_void_p), ("seed_skip", C.c_uint64), ("offset_skip", C.c_uint64)]

    class NormGrads(C.Structure):
        _fields_ = [(k, C.c_void_p) for k in ("dw_shared", "db_shared", "dw2a", "dtable", "dgamma_beta_sums")]

    class BlockGrads(C.Structure):
        _fields_ = [("norm_0", NormGrads), ("norm_1", NormGrads)] + [(k, C.c_void_p) for k in
                    ("dw_conv_0", "db_conv_0", "dw_conv_1", "db_conv_1", "dw_noise_middle", "dw_noise_skip")]

    has_t = norms[0]["table"] is not None
    keep = []

    def dev(t):
        t = t.detach().contiguous()
        keep.append(t)
        return t.data_ptr()

    def layer(d):
        return NormLayer(dev(d["w_sh"]), dev(d["b_sh"]), dev(d["w2a"]), dev(d["table"]) if has_t else None, dev(d["b2"]),
                         d["rm"].data_ptr(), d["rv"].data_ptr(), float(d["add_one"]))

    n0, n1 = layer(norms[0]), layer(norms[1])
    bn = BlockNoise(w_mid.data_ptr(), eps_mid.seed, eps_mid.offset, w_skip.data_ptr(), eps_skip.seed, eps_skip.offset)
    lib = L.lib()
    sbytes = lib.dsee_spade_resblock_saved_bytes(N, R, R, Cc, Lc, int(has_t))
    fbytes = lib.dsee_spade_resblock_train_fwd_workspace(N, R, R, Cc, Lc, int(has_t))
    shift = norms[0]["shift"]
    bbytes = lib.dsee_spade_resblock_bwd_workspace(N, R, R, Cc, Lc, int(has_t), labels.h, labels.w, shift)
    saved = torch.empty(sbytes, dtype=torch.uint8, device="cuda")
    ws = torch.empty(max(fbytes, bbytes), dtype=torch.uint8, device="cuda")
    out = torch.empty_like(x)
    xd = x.detach()
    w0d, w1d = w0.detach(), w1.detach()
    eps_mid.bind()
    L.call("spade_resblock_train_fwd", C.byref(n0), w0d, b0.detach(), C.byref(n1), w1d, b1.detach(), C.byref(bn), labels.t, labels.h,
           labels.w, shift, Lc, xd, out, 1e-5, 0.1, 0.2, N, R, R, Cc, saved, C.c_size_t(sbytes), ws, C.c_size_t(ws.numel()))
    torch.cuda.synchronize()
    assert torch.equal(out, want.detach()), rel(out.cpu(), want.detach().cpu())
    for d in norms:
        assert torch.equal(d["rm"], d["mod_rm"]) and torch.equal(d["rv"], d["mod_rv"])
    # ---- backward
    rows = norms[0]["w2a"].shape[0]

    def ngrads(d):
        z = dict(dw_sh=torch.full_like(d["w_sh"].detach(), float("nan")), db_sh=torch.full_like(d["b_sh"].detach(), float("nan")),
                 dw2a=torch.full((rows, 128, 3, 3), float("nan"), device="cuda"),
                 dtable=torch.full((N, 9, rows, 32), float("nan"), device="cuda") if has_t else None,
                 dsum=torch.full((2, Cc), float("nan"), device="cuda"))
        keep.append(z)
        return z, NormGrads(z["dw_sh"].data_ptr(), z["db_sh"].data_ptr(), z["dw2a"].data_ptr(),
                            z["dtable"].data_ptr() if has_t else None, z["dsum"].data_ptr())

    g0, ng0 = ngrads(norms[0])
    g1, ng1 = ngrads(norms[1])
    cg = {k: torch.full_like(v, float("nan")) for k, v in dict(dw0=w0d, db0=b0.detach(), dw1=w1d, db1=b1.detach(), dwm=w_mid.detach(),
                                                              dws=w_skip.detach()).items()}
    bg = BlockGrads(ng0, ng1, cg["dw0"].data_ptr(), cg["db0"].data_ptr(), cg["dw1"].data_ptr(), cg["db1"].data_ptr(),
                    cg["dwm"].data_ptr(), cg["dws"].data_ptr())
    dx = torch.full_like(xd, float("nan"))
    amax_dx = torch.zeros(2048, device="cuda")
    L.call("spade_resblock_bwd", C.byref(n0), w0d, C.byref(n1), w1d, C.byref(bn), labels.t, labels.h, labels.w, shift, Lc, xd, gout,
           gout.dsee_amax, C.byref(bg), dx, amax_dx, 0.2, N, R, R, Cc, saved, C.c_size_t(sbytes), ws, C.c_size_t(ws.numel()))
    torch.cuda.synchronize()
    checks = {"dx": (dx, x.grad), "dw_conv_0": (cg["dw0"], w0.grad), "dw_conv_1": (cg["dw1"], w1.grad),
              "db_conv_0": (cg["db0"], b0.grad), "db_conv_1": (cg["db1"], b1.grad),
              "dw_noise_middle": (cg["dwm"], w_mid.grad), "dw_noise_skip": (cg["dws"], w_skip.grad)}
    idx, _ = ops.packed_perm(Cc, "cuda")
    for i, (d, z) in enumerate(zip(norms, (g0, g1))):
        checks["norm_%d.dw_shared" % i] = (z["dw_sh"], d["w_sh"].grad)
        checks["norm_%d.db_shared" % i] = (z["db_sh"], d["b_sh"].grad)
        checks["norm_%d.dw2a" % i] = (z["dw2a"], d["w2a"].grad)
        if has_t:
            checks["norm_%d.dtable" % i] = (z["dtable"], d["table"].grad)
        packed = torch.cat([z["dsum"].reshape(-1), torch.zeros(1, device="cuda")]).index_select(0, idx)
        checks["norm_%d.dbias_packed" % i] = (packed, d["b2"].grad)
    bad = {k: rel(a.cpu(), b.cpu()) for k, (a, b) in checks.items() if b is None or not torch.equal(a, b)}
    print("coarse training pair (%s): %d tensors compared, not bit-identical: %s" % (kind, len(checks), bad))
    assert not bad, bad
    assert float(amax_dx.max()) == float(x.grad.abs().max())
    blk.zero_grad()
    # shapes the pre-split kernels do not tile are refused, not mis-computed
    with pytest.raises(L.DseeError, match="pre-split training path"):
        L.call("spade_resblock_train_fwd", C.byref(n0), w0d, b0.detach(), C.byref(n1), w1d, b1.detach(), C.byref(bn), labels.t,
               labels.h, labels.w, shift + 1, Lc, xd, out, 1e-5, 0.1, 0.2, N, 32, 32, Cc, saved, C.c_size_t(sbytes), ws,
               C.c_size_t(ws.numel()))
