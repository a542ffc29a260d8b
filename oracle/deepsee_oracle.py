"""CPU oracle for the DeepSEE train-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional, plain-PyTorch-CPU restatement of the
reference algorithm.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product (``deepsee_amd``) never does.

Parity status: PINNED.  ``oracle/gen_golden.py`` (run in the build container,
where ``/root/reference`` is importable) loads identical recipe weights into
the real reference modules and into this oracle, runs G and D steps on the
same inputs/noise, and asserts agreement; the numbers it wrote to
``tests/golden/*.json`` are re-checked against this file by
``tests/test_oracle_golden.py`` on every run.

All tensors are NCHW fp32 on CPU.  Parameters/buffers live in flat
``OrderedDict``s whose keys and shapes equal the reference ``state_dict()``
(SURVEY.md Appendix A), so checkpoints interchange.

Reference citations are relative to /root/reference/.
"""
import math
import random
import zlib
from collections import OrderedDict
from types import SimpleNamespace

import torch
import torch.nn.functional as F

LRELU = 0.2
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
SN_EPS = 1e-12
NHIDDEN = 128  # normalization.py:95 "Yes, hardcoded."


# --------------------------------------------------------------------------- opt
def make_opt(**over):
    """Flat option namespace with the reference defaults (options/base_options.py,
    options/train_options.py, options/configurations.py)."""
    d = dict(
        name="oracle", gpu_ids=[], model="sr",
        norm_G="spectrallateseansyncbatch3x3", norm_D="spectralinstance", norm_E="spectralinstance",
        add_noise=True, noisy_style_scale=0.2, noisy_style_dist="uniform",
        batchSize=8, load_size=256, crop_size=256, start_size=32, aspect_ratio=1.0,
        label_nc=19, contain_dontcare_label=False, semantic_nc=19, output_nc=3,
        max_fm_size=256, downsampling_method="bicubic",
        netG="deepsee", netE="combinedstyle", netD="multiscale", netD_subarch="n_layer",
        ngf=32, nef=32, ndf=32, num_D=2, n_layers_D=4,
        init_type="xavier", init_variance=0.02, regional_style_size=128,
        full_style_image=False, guiding_style_image=False, random_style_matrix=False,
        model_parallel_mode=0, isTrain=True, continue_train=False, which_epoch="latest",
        beta1=0.0, beta2=0.9, no_TTUR=False, efficient=False, lr=2e-4,
        lambda_feat=10.0, lambda_vgg=10.0, no_ganFeat_loss=False, no_vgg_loss=False,
        gan_mode="hinge", gradient_clip=-1.0, num_upsampling_layers="normal",
        niter=50, niter_decay=25, gpu_info=False, checkpoints_dir="./checkpoints",
    )
    d.update(over)
    return SimpleNamespace(**d)


def variant_of(opt):
    # sr_model.py:29
    return "guided" if "full" in opt.netE else "independent"


def n_blocks_of(opt):
    # sr.py:27
    return int(round(math.log2(opt.crop_size) - math.log2(opt.start_size)))


def block_plan(opt):
    """[(state-dict prefix, kind)] in forward order; kind in spade|sean|puresean.
    sr.py:33-51: head_0 is SPADE because 'late' in norm_G; tail blocks turn
    PureSEAN when load_size >= 512."""
    nb = n_blocks_of(opt)
    early_style = "late" not in opt.norm_G
    has_sean = "sean" in opt.norm_G.replace("spectral", "")
    sk = "sean" if has_sean else "spade"
    plan = [("head_0", sk if early_style else "spade"), ("G_middle_0", sk), ("G_middle_1", sk)]
    max_nb = 4 if opt.load_size >= 512 else 99
    n_sean = max(0, min(nb, max_nb) - 1)
    kinds = [sk] * n_sean
    if max_nb != 99:
        kinds += ["puresean"] * max(0, nb - max_nb)
    for i, k in enumerate(kinds):
        plan.append(("up_list.%d" % i, k))
    return plan


# --------------------------------------------------------------------------- state layout
def _add_sn_conv(spec, p, cout, cin, k, bias):
    if bias:
        spec[p + ".bias"] = (cout,)
    spec[p + ".weight_orig"] = (cout, cin, k, k)
    spec[p + ".weight_u"] = (cout,)
    spec[p + ".weight_v"] = (cin * k * k,)


def _add_conv(spec, p, cout, cin, k):
    spec[p + ".weight"] = (cout, cin, k, k)
    spec[p + ".bias"] = (cout,)


def sr_spec(opt):
    C, L, S = 16 * opt.ngf, opt.semantic_nc, opt.regional_style_size
    spec = OrderedDict()
    _add_conv(spec, "initial", C, 3, 3)
    for prefix, kind in block_plan(opt):
        for cv in ("conv_0", "conv_1"):
            _add_sn_conv(spec, "%s.%s" % (prefix, cv), C, C, 3, True)
        for nm in ("norm_0", "norm_1"):
            p = "%s.%s" % (prefix, nm)
            spec[p + ".param_free_norm.running_mean"] = (C,)
            spec[p + ".param_free_norm.running_var"] = (C,)
            spec[p + ".param_free_norm.num_batches_tracked"] = ()
            _add_conv(spec, p + ".mlp_shared.0", NHIDDEN, L, 3)
            if kind in ("spade", "sean"):
                _add_conv(spec, p + ".mlp_gamma", C, NHIDDEN, 3)
                _add_conv(spec, p + ".mlp_beta", C, NHIDDEN, 3)
            if kind in ("sean", "puresean"):
                spec[p + ".style_conv.weight"] = (19, 19, 1)  # defined, never used (normalization.py:156)
                spec[p + ".style_conv.bias"] = (19,)
                _add_conv(spec, p + ".mlp_style_gamma", C, S, 3)
                _add_conv(spec, p + ".mlp_style_beta", C, S, 3)
            if kind == "sean":
                spec[p + ".alpha_beta"] = (1,)
                spec[p + ".alpha_gamma"] = (1,)
        if opt.add_noise:
            for nz in ("noise_in", "noise_skip", "noise_middle"):
                spec["%s.%s.weight" % (prefix, nz)] = (C,)
    _add_conv(spec, "conv_img", 3, C, 3)
    return spec


def d_spec(opt):
    spec = OrderedDict()
    cin0 = opt.label_nc + opt.output_nc + (1 if opt.contain_dontcare_label else 0)
    for i in range(opt.num_D):
        p = "discriminator_%d" % i
        nf = opt.ndf
        _add_conv(spec, p + ".model0.0", nf, cin0, 4)
        for n in range(1, opt.n_layers_D):
            prev, nf = nf, min(nf * 2, 512)
            _add_sn_conv(spec, "%s.model%d.0.0" % (p, n), nf, prev, 4, False)
        _add_conv(spec, "%s.model%d.0" % (p, opt.n_layers_D), 1, nf, 4)
    return spec


def _enc_branch(spec, p, nf, names, cin0, out):
    chans = [(nf, cin0), (2 * nf, nf), (4 * nf, 2 * nf), (8 * nf, 4 * nf)]
    for (nm, (co, ci)) in zip(names, chans):
        _add_sn_conv(spec, p + nm, co, ci, 3, False)


def e_spec(opt):
    nf, S = opt.nef, opt.regional_style_size
    spec = OrderedDict()
    full_names = ["initial.0.0", "down0.0.0", "down1.0.0", "up_conv.1.0"]
    mini_names = ["initial.0.0", "conv0.0.0", "conv1.0.0", "conv2.1.0"]
    cin_full = opt.label_nc if opt.random_style_matrix else 3
    if opt.netE == "combinedstyle":
        if opt.noisy_style_scale > 0:
            spec["noise_weights"] = (opt.label_nc,)
        _add_sn_conv(spec, "final.0.0", S, 8 * nf, 3, False)
        _add_sn_conv(spec, "encoder_full.final.0.0", S, 8 * nf, 3, False)   # constructed, unused
        _enc_branch(spec, "encoder_full.", nf, full_names, cin_full, S)
        _add_sn_conv(spec, "encoder_mini.final.0.0", S, 8 * nf, 3, False)   # constructed, unused
        _enc_branch(spec, "encoder_mini.", nf, mini_names, 3, S)
    elif opt.netE == "fullstyle":
        if opt.noisy_style_scale > 0:
            spec["noise_weights"] = (opt.label_nc,)
        _add_sn_conv(spec, "final.0.0", S, 8 * nf, 3, False)
        _enc_branch(spec, "", nf, full_names, cin_full, S)
    else:
        raise NotImplementedError(opt.netE)  # ministyle crashes in the reference too (SURVEY B-13)
    return spec


VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512]
VGG_TAPS = (1, 6, 11, 20, 29)  # indices of the relu whose output is tapped (architecture.py:161-170)


def vgg_spec():
    spec, cin, idx = OrderedDict(), 3, 0
    for v in VGG_CFG:
        if v == "M":
            idx += 1
            continue
        spec["features.%d.weight" % idx] = (v, cin, 3, 3)
        spec["features.%d.bias" % idx] = (v,)
        cin = v
        idx += 2
    return spec


def net_specs(opt):
    return {"SR": sr_spec(opt), "D": d_spec(opt), "E": e_spec(opt), "VGG": vgg_spec()}


def is_buffer(key):
    return key.endswith(("weight_u", "weight_v", "running_mean", "running_var", "num_batches_tracked"))


def recipe_tensor(net, key, shape, gain=1.0):
    """Deterministic 'recipe' value for one state-dict entry; depends only on
    (net, key, shape, gain) so the reference, the oracle and the HIP build all
    get the same numbers without shipping weights."""
    g = torch.Generator().manual_seed(zlib.crc32(("%s/%s" % (net, key)).encode()) & 0x7FFFFFFF)
    shape = tuple(shape)
    if key.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.long)
    if key.endswith("running_mean"):
        return torch.randn(shape, generator=g) * 0.1
    if key.endswith("running_var"):
        return torch.rand(shape, generator=g) * 0.5 + 0.75
    if key.endswith(("weight_u", "weight_v")):
        t = torch.randn(shape, generator=g)
        return t / t.norm().clamp_min(1e-12)
    if key.endswith(("alpha_beta", "alpha_gamma")):
        return torch.rand(shape, generator=g)
    if key == "noise_weights":
        return torch.randn(shape, generator=g) * 0.5
    if key.endswith(("noise_in.weight", "noise_skip.weight", "noise_middle.weight")):
        return torch.randn(shape, generator=g) * 0.1
    if key.endswith("bias"):
        return torch.randn(shape, generator=g) * 0.05
    # conv weights: kaiming-like scale so activations are O(1) (SURVEY B-12)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) * (gain * math.sqrt(2.0 / max(fan_in, 1)))


def recipe_state(opt, gain=1.0):
    out = {}
    for net, spec in net_specs(opt).items():
        out[net] = OrderedDict((k, recipe_tensor(net, k, s, gain)) for k, s in spec.items())
    return out


def init_state(opt, seed=0):
    """Training init with the reference's semantics (base_network.py:28-59,
    networks/__init__.py:37-43): xavier_normal(gain=init_variance) on conv
    weights, zero biases, SN u/v ~ normalised N(0,1), alpha ~ U(0,1), noise 0."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for net, spec in net_specs(opt).items():
        st = OrderedDict()
        for k, shape in spec.items():
            if k.endswith("num_batches_tracked"):
                t = torch.zeros((), dtype=torch.long)
            elif k.endswith("running_mean"):
                t = torch.zeros(shape)
            elif k.endswith("running_var"):
                t = torch.ones(shape)
            elif k.endswith(("weight_u", "weight_v")):
                t = torch.randn(shape, generator=g)
                t = t / t.norm().clamp_min(1e-12)
            elif k.endswith(("alpha_beta", "alpha_gamma")):
                t = torch.rand(shape, generator=g)
            elif k.endswith("bias") or k == "noise_weights" or ".noise_" in k:
                t = torch.zeros(shape)
            else:
                rf = 1
                for s in shape[2:]:
                    rf *= s
                fan_in, fan_out = shape[1] * rf, shape[0] * rf
                if net == "VGG":
                    std = math.sqrt(2.0 / fan_in)  # no pretrained download available: He init
                elif opt.init_type == "xavier":
                    std = opt.init_variance * math.sqrt(2.0 / (fan_in + fan_out))
                elif opt.init_type == "kaiming":
                    std = math.sqrt(2.0 / fan_in)
                elif opt.init_type == "normal":
                    std = opt.init_variance
                else:
                    raise NotImplementedError(opt.init_type)
                t = torch.randn(shape, generator=g) * std
            st[k] = t
        out[net] = st
    return out


# --------------------------------------------------------------------------- randomness control
class Ctl:
    """Every random draw on the path, in the reference's call order.  The default
    draws from torch's global CPU generator / python ``random`` exactly like the
    reference (normalization.py:299-304, encoder.py:59-61, sr_model.py:616,643)."""

    def coin(self, tag):
        return random.random()

    def normal(self, shape, tag):
        return torch.empty(shape).normal_()

    def uniform(self, shape, tag):
        return torch.rand(shape)


class RecordingCtl(Ctl):
    def __init__(self):
        self.tape = []

    def coin(self, tag):
        v = random.random()
        self.tape.append(("coin", tag, v))
        return v

    def normal(self, shape, tag):
        v = torch.empty(shape).normal_()
        self.tape.append(("normal", tag, v))
        return v

    def uniform(self, shape, tag):
        v = torch.rand(shape)
        self.tape.append(("uniform", tag, v))
        return v


class _CastCtl(Ctl):
    def __init__(self, inner, dtype):
        self.inner, self.dtype = inner, dtype

    def coin(self, tag):
        return self.inner.coin(tag)

    def normal(self, shape, tag):
        return self.inner.normal(shape, tag).to(self.dtype)

    def uniform(self, shape, tag):
        return self.inner.uniform(shape, tag).to(self.dtype)


class ReplayCtl(Ctl):
    def __init__(self, tape):
        self.tape, self.pos = list(tape), 0

    def _next(self, kind, tag):
        k, t, v = self.tape[self.pos]
        assert k == kind and t == tag, "replay mismatch: want %s/%s got %s/%s" % (kind, tag, k, t)
        self.pos += 1
        return v

    def coin(self, tag):
        return self._next("coin", tag)

    def normal(self, shape, tag):
        v = self._next("normal", tag)
        assert tuple(v.shape) == tuple(shape)
        return v

    def uniform(self, shape, tag):
        v = self._next("uniform", tag)
        assert tuple(v.shape) == tuple(shape)
        return v


# --------------------------------------------------------------------------- primitive ops
def lrelu(x):
    return F.leaky_relu(x, LRELU)


def onehot_labels(label, nc, dtype=torch.float32):
    """data/preprocessor.py:35-41 — float label map [N,1,H,W] (0..nc-1) -> one-hot."""
    lab = label.long()
    n, _, h, w = lab.shape
    return torch.zeros(n, nc, h, w, dtype=dtype).scatter_(1, lab, 1.0)


def bicubic_down(img, size):
    """data/preprocessor.py:17-33 — bicubic (A=-0.75, align_corners=False, no antialias) + clamp."""
    return F.interpolate(img, (size, size), mode="bicubic").clamp(min=-1, max=1)


def device_pipeline_reference(image_u8, label_u8, flip, label_nc):
    """What the reference's loader hands the model for one uint8 sample pair (data/base_dataset.py:87-116,171-201 after
    the PIL resize/crop): __flip(img, params['flip']) on both, transforms.ToTensor() (v / 255 as float32, HWC -> CHW),
    transforms.Normalize((.5,.5,.5),(.5,.5,.5)) on the image; label = ToTensor(label) * 255.0 with 255 -> label_nc.
    torchvision is absent from this image, so these three transforms are restated from their published definitions:
    parity unpinned for this piece (the arithmetic is (v/255 - 0.5)/0.5 and a horizontal mirror).
    image_u8 [N,H,W,3], label_u8 [N,H,W], flip [N] -> (image [N,3,H,W] f32, label [N,1,H,W] f32)."""
    img = image_u8.float().div(255.0)
    img = (img - 0.5) / 0.5
    lab = label_u8.float().div(255.0) * 255.0
    lab = torch.where(lab == 255, torch.full_like(lab, float(label_nc)), lab)
    f = flip.bool()
    img = torch.where(f[:, None, None, None], img.flip(2), img)
    lab = torch.where(f[:, None, None], lab.flip(2), lab)
    return img.permute(0, 3, 1, 2).contiguous(), lab[:, None].contiguous()


def metric_case_inputs(seed, n, h, w, kind):
    """Seeded (fake, real) image pairs [n,3,h,w] of tests/golden/metrics.json (shared by the generator and the tests)."""
    g = torch.Generator().manual_seed(seed)
    real = torch.rand(n, 3, h, w, generator=g) * 2 - 1
    if kind == "smooth":
        real = F.interpolate(torch.rand(n, 3, 8, 8, generator=g), (h, w), mode="bilinear") * 2 - 1
    if kind == "noise":
        fake = torch.rand(n, 3, h, w, generator=g) * 2 - 1
    elif kind == "saturated":
        fake = real * 1.7 + 0.2 * torch.randn(n, 3, h, w, generator=g)     # values beyond [-1, 1]: clipped by tensor2im
    elif kind == "identical":
        fake = real.clone()
    else:
        fake = real + 0.05 * torch.randn(n, 3, h, w, generator=g)
    return fake, real


def psnr_ssim_rmse(fake, real):
    """MetricsEvaluator.collect_samples' PSNR / SSIM / RMSE for a batch (evaluator/evaluation.py:88-137): images
    [N,3,H,W] in [-1,1] -> float64 [N,3].  tensor2im (util/util.py:93-103): uint8(clip((x+1)/2*255, 0, 255)) in float32;
    calculate_psnr (evaluator/calculate_PSNR_SSIM.py:71-79) in float64 over the whole image; calculate_ssim (:81-122):
    Gaussian 11x11 window (cv2.getGaussianKernel(11, 1.5) = normalised exp(-(i-5)^2/(2*1.5^2)), outer product), windowed
    moments on the valid region, C1 = (0.01*255)^2, C2 = (0.03*255)^2, mean of the map over positions and channels (the
    channel loop at :113-116 passes the full 3-channel image each time); RMSE on the [-1,1] values (evaluation.py:107-110).
    Pinned by tests/golden/metrics.json, which oracle/gen_golden.py writes by running the reference's own functions (cv2 is
    absent from this image: its two calls, getGaussianKernel and filter2D, are provided there from their documented
    definitions through scipy)."""
    import numpy as np
    from scipy.signal import correlate2d
    k = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    k /= k.sum()
    window = np.outer(k, k)
    out = []
    for f, r in zip(fake, real):
        f32, r32 = f.detach().cpu().float().numpy(), r.detach().cpu().float().numpy()
        q = [np.clip((np.transpose(a, (1, 2, 0)) + 1) / 2.0 * 255.0, 0, 255).astype(np.uint8) for a in (f32, r32)]
        a, b = q[0].astype(np.float64), q[1].astype(np.float64)
        mse = np.mean((a - b) ** 2)
        psnr = float("inf") if mse == 0 else 20 * np.log10(255.0 / np.sqrt(mse))
        c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
        vals = []
        for c in range(3):
            filt = lambda z: correlate2d(z, window, mode="valid")
            mu1, mu2 = filt(a[..., c]), filt(b[..., c])
            s11 = filt(a[..., c] ** 2) - mu1 ** 2
            s22 = filt(b[..., c] ** 2) - mu2 ** 2
            s12 = filt(a[..., c] * b[..., c]) - mu1 * mu2
            vals.append(((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s11 + s22 + c2)))
        rmse = float(np.sqrt(np.mean((f32.astype(np.float64) - r32.astype(np.float64)) ** 2)))
        out.append([psnr, float(np.mean(vals)), rmse])
    return torch.tensor(out, dtype=torch.float64)


def nearest_resize(x, size):
    """F.interpolate(mode='nearest'): src = floor(dst * in / out) (SURVEY B-3)."""
    return F.interpolate(x, size=size, mode="nearest")


def batch_norm_train(x, st, prefix, training):
    """sync_batchnorm/batchnorm.py:65-68 single-device branch == F.batch_norm:
    biased batch var + eps; running stats momentum .1 with unbiased var."""
    rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
    if not training:
        return (x - rm[None, :, None, None]) / torch.sqrt(rv[None, :, None, None] + BN_EPS)
    m = x.numel() // x.shape[1]
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    with torch.no_grad():
        rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
        rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * (m / max(m - 1, 1)))
    return (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)


def sync_bn_master(shards, running_mean, running_var, eps=1e-5, momentum=0.1):
    """The DataParallel branch of SynchronizedBatchNorm2d (sync_batchnorm/batchnorm.py:70-145) on a list of replica
    inputs [n_r, C, H, W]: every replica sends (sum, ssum) to the master (:77-85), the master adds them
    (ReduceAddCoalesced, :117) and computes mean = sum/size, sumvar = ssum - sum*mean, biased / unbiased variance,
    running-stat momentum update with the UNBIASED variance, inv_std = clamp(bias_var, eps)^-1/2 (:128-145) --
    clamp, not `+ eps` as F.batch_norm.  Returns (mean, inv_std, new_running_mean, new_running_var, outputs)."""
    c = shards[0].shape[1]
    size = sum(s.shape[0] * s.shape[2] * s.shape[3] for s in shards)
    sum_ = sum(s.transpose(0, 1).reshape(c, -1).sum(1) for s in shards)
    ssum = sum((s ** 2).transpose(0, 1).reshape(c, -1).sum(1) for s in shards)
    mean = sum_ / size
    sumvar = ssum - sum_ * mean
    unbias_var, bias_var = sumvar / (size - 1), sumvar / size
    rm = (1 - momentum) * running_mean + momentum * mean
    rv = (1 - momentum) * running_var + momentum * unbias_var
    inv_std = bias_var.clamp(eps) ** -0.5
    outs = [(s - mean[None, :, None, None]) * inv_std[None, :, None, None] for s in shards]
    return mean, inv_std, rm, rv, outs


def instance_norm(x):
    """nn.InstanceNorm2d(affine=False): per-(n,c) biased var, eps 1e-5 (normalization.py:47-48)."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + BN_EPS)


def spectral_weight(st, prefix, training):
    """torch.nn.utils.spectral_norm (hook form; call sites architecture.py:40-44,
    normalization.py:29-30): one power iteration per train-mode forward, in place
    on the u/v buffers (also under no_grad); sigma = u^T W v; W = W_orig / sigma."""
    w = st[prefix + ".weight_orig"]
    u, v = st[prefix + ".weight_u"], st[prefix + ".weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=SN_EPS))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=SN_EPS))
    sigma = torch.dot(u.clone(), torch.mv(wm, v.clone()))
    return w / sigma


def noise_inject(x, weight, ctl, tag):
    """normalization.py:289-304 — x + w_c * eps, eps ~ N(0,1) fresh per element."""
    eps = ctl.normal(tuple(x.shape), tag)
    return x + weight[None, :, None, None] * eps


def style_gather(style, seg):
    """normalization.py:179-185 — style_map[b,:,h,w] = sum_r style[b,r,:] * seg[b,r,h,w]."""
    return torch.einsum("brs,brhw->bshw", style, seg)


def style_pool(feat, seg):
    """encoder.py:36-49 — S[b,r,c] = mean_hw(feat[b,c] * seg[b,r]); divides by H*W,
    not by region area (SURVEY B-4)."""
    if seg.shape[2:] != feat.shape[2:]:
        seg = nearest_resize(seg, feat.shape[2:])
    hw = feat.shape[2] * feat.shape[3]
    return torch.einsum("bchw,brhw->brc", feat, seg) / hw


# --------------------------------------------------------------------------- the oracle model
class Oracle:
    """Holds the four state dicts and evaluates the path functionally."""

    def __init__(self, opt, states, ctl=None, dtype=torch.float32):
        """dtype=torch.float64 turns the oracle into an exact-arithmetic yardstick: tests compare
        err(HIP, f64) with err(this oracle in f32, f64)."""
        self.opt = opt
        self.dtype = dtype
        self.ctl = _CastCtl(ctl or Ctl(), dtype) if dtype != torch.float32 else (ctl or Ctl())
        self.S = {}
        for net, st in states.items():
            d = OrderedDict()
            for k, v in st.items():
                t = v.detach().clone()
                if t.is_floating_point():
                    t = t.to(dtype)
                if not is_buffer(k) and net != "VGG" and t.is_floating_point():
                    t.requires_grad_(True)
                d[k] = t
            self.S[net] = d
        self.training = True
        self.last_encoded_style_is_full = True
        self.last_encoded_style_is_noisy = False
        self.opt_G = self.opt_D = None

    # ---- normalisation blocks (normalization.py:71-286)
    def _mlp_shared(self, st, p, seg):
        return F.relu(F.conv2d(seg, st[p + ".mlp_shared.0.weight"], st[p + ".mlp_shared.0.bias"], padding=1))

    def spade(self, st, p, x, segmap):
        xn = batch_norm_train(x, st, p + ".param_free_norm", self.training)
        seg = nearest_resize(segmap, x.shape[2:])
        actv = self._mlp_shared(st, p, seg)
        gamma = F.conv2d(actv, st[p + ".mlp_gamma.weight"], st[p + ".mlp_gamma.bias"], padding=1)
        beta = F.conv2d(actv, st[p + ".mlp_beta.weight"], st[p + ".mlp_beta.bias"], padding=1)
        return xn * (1 + gamma) + beta

    def _sean_maps(self, st, p, x, segmap, style):
        out_size = tuple(x.shape[2:])
        fm = tuple(min(s, self.opt.max_fm_size) for s in out_size)
        seg = nearest_resize(segmap, fm)
        actv = self._mlp_shared(st, p, seg)
        smap = style_gather(style, seg)
        if fm != out_size:
            # normalization.py:188-190 / 275-277: BOTH become the upsampled SPADE activation
            actv = nearest_resize(actv, out_size)
            smap = nearest_resize(actv, out_size)
        return actv, smap

    def sean(self, st, p, x, segmap, style):
        xn = batch_norm_train(x, st, p + ".param_free_norm", self.training)
        actv, smap = self._sean_maps(st, p, x, segmap, style)
        gamma = F.conv2d(actv, st[p + ".mlp_gamma.weight"], st[p + ".mlp_gamma.bias"], padding=1)
        beta = F.conv2d(actv, st[p + ".mlp_beta.weight"], st[p + ".mlp_beta.bias"], padding=1)
        beta_s = F.conv2d(smap, st[p + ".mlp_style_beta.weight"], st[p + ".mlp_style_beta.bias"], padding=1)
        gamma_s = F.conv2d(smap, st[p + ".mlp_style_gamma.weight"], st[p + ".mlp_style_gamma.bias"], padding=1)
        wb = torch.sigmoid(st[p + ".alpha_beta"])
        wg = torch.sigmoid(st[p + ".alpha_gamma"])
        offset = wb * beta_s + (1.0 - wb) * beta
        scale = wg * gamma_s + (1.0 - wg) * gamma + 1
        return xn * scale + offset

    def puresean(self, st, p, x, segmap, style):
        xn = batch_norm_train(x, st, p + ".param_free_norm", self.training)
        _, smap = self._sean_maps(st, p, x, segmap, style)
        beta_s = F.conv2d(smap, st[p + ".mlp_style_beta.weight"], st[p + ".mlp_style_beta.bias"], padding=1)
        gamma_s = F.conv2d(smap, st[p + ".mlp_style_gamma.weight"], st[p + ".mlp_style_gamma.bias"], padding=1)
        return xn * gamma_s + beta_s

    def _norm(self, kind, st, p, x, seg, style):
        if kind == "spade":
            return self.spade(st, p, x, seg)
        if kind == "sean":
            return self.sean(st, p, x, seg, style)
        return self.puresean(st, p, x, seg, style)

    # ---- resblock (architecture.py:75-147), fin == fout so the shortcut is the identity
    def resblock(self, st, p, kind, x, seg, style):
        noisy = self.opt.add_noise and self.training and (p + ".noise_in.weight") in st
        if noisy:
            x = noise_inject(x, st[p + ".noise_in.weight"], self.ctl, p + ".noise_in")
        x_s = x
        if noisy:
            x_s = noise_inject(x, st[p + ".noise_skip.weight"], self.ctl, p + ".noise_skip")
        h = lrelu(self._norm(kind, st, p + ".norm_0", x, seg, style))
        dx = F.conv2d(h, spectral_weight(st, p + ".conv_0", self.training), st[p + ".conv_0.bias"], padding=1)
        if noisy:
            dx = noise_inject(dx, st[p + ".noise_middle.weight"], self.ctl, p + ".noise_middle")
        h = lrelu(self._norm(kind, st, p + ".norm_1", dx, seg, style))
        dx = F.conv2d(h, spectral_weight(st, p + ".conv_1", self.training), st[p + ".conv_1.bias"], padding=1)
        return x_s + dx

    # ---- generator (sr.py:62-98)
    def sr_forward(self, image_lr, seg, style):
        st = self.S["SR"]
        x = F.conv2d(image_lr, st["initial.weight"], st["initial.bias"], padding=1)
        plan = block_plan(self.opt)
        x = self.resblock(st, plan[0][0], plan[0][1], x, seg, style)
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = self.resblock(st, plan[1][0], plan[1][1], x, seg, style)
        x = self.resblock(st, plan[2][0], plan[2][1], x, seg, style)
        for prefix, kind in plan[3:]:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = self.resblock(st, prefix, kind, x, seg, style)
        x = F.conv2d(lrelu(x), st["conv_img.weight"], st["conv_img.bias"], padding=1)
        return torch.tanh(x)

    # ---- encoders (encoder.py)
    def _enc_layer(self, st, p, x, stride):
        w = spectral_weight(st, p, self.training)
        return lrelu(instance_norm(F.conv2d(x, w, None, stride=stride, padding=1)))

    def _enc_main(self, st, prefix, mode, x):
        if mode == "full":       # encoder.py:83-99: s1, s2, s2, up x2 + conv
            x = self._enc_layer(st, prefix + "initial.0.0", x, 1)
            x = self._enc_layer(st, prefix + "down0.0.0", x, 2)
            x = self._enc_layer(st, prefix + "down1.0.0", x, 2)
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = self._enc_layer(st, prefix + "up_conv.1.0", x, 1)
        else:                    # encoder.py:142-158: all stride 1 on the LR image
            x = self._enc_layer(st, prefix + "initial.0.0", x, 1)
            x = self._enc_layer(st, prefix + "conv0.0.0", x, 1)
            x = self._enc_layer(st, prefix + "conv1.0.0", x, 1)
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = self._enc_layer(st, prefix + "conv2.1.0", x, 1)
        return x

    def encoder_forward(self, x, seg, mode, no_noise):
        """CombinedstyleEncoder.forward (encoder.py:195-210) / FullStyleEncoder.forward (:116-132)."""
        st = self.S["E"]
        combined = self.opt.netE == "combinedstyle"
        prefix = ("encoder_full." if mode == "full" else "encoder_mini.") if combined else ""
        x = self._enc_main(st, prefix, mode, x)
        w = spectral_weight(st, "final.0.0", self.training)
        x = torch.tanh(instance_norm(F.conv2d(x, w, None, padding=1)))
        sm = style_pool(x, seg)
        if self.opt.noisy_style_scale > 0 and not no_noise:
            # encoder.py:51-70 corrupt_style_matrix (uniform branch), then clamp
            nw = torch.sigmoid(st["noise_weights"])[None, :, None]
            if self.opt.noisy_style_dist == "uniform":
                noise = (self.ctl.uniform(tuple(sm.shape), "style_noise") * 2 - 1) * self.opt.noisy_style_scale
            else:
                noise = (self.ctl.normal(tuple(sm.shape), "style_noise") * 2 - 1) * self.opt.noisy_style_scale
            sm = (sm + noise * nw).clamp(-1, 1)
        return sm

    def encode_style(self, data, no_noise):
        """sr_model.py:582-650."""
        opt = self.opt
        seg, img = data["input_semantics"], data["image_lr"]
        if variant_of(opt) == "guided":
            mode = "full"
            if opt.guiding_style_image:
                seg, img = data["guiding_label"], data["guiding_image"]
            else:
                img = data["image_hr"]
            return self.encoder_forward(img, seg, mode, no_noise)
        if opt.full_style_image or (self.training and self.ctl.coin("enc_full") < 0.5):
            mode = "full"
            self.last_encoded_style_is_full = True
            if opt.guiding_style_image:
                seg, img = data["guiding_label"], data["guiding_image"]
            else:
                img = data["image_hr"]
        else:
            mode = "mini"
            self.last_encoded_style_is_full = False
        if not no_noise:
            no_noise = self.ctl.coin("enc_noise") < 0.5
            self.last_encoded_style_is_noisy = not no_noise
        return self.encoder_forward(img, seg, mode, no_noise)

    def generate_fake(self, data, no_noise=False):
        style = self.encode_style(data, no_noise)
        return self.sr_forward(data["image_lr"], data["input_semantics"], style), style

    # ---- discriminator (discriminator.py)
    def _nlayer_d(self, st, p, x):
        outs = []
        x = lrelu(F.conv2d(x, st[p + ".model0.0.weight"], st[p + ".model0.0.bias"], stride=2, padding=2))
        outs.append(x)
        nl = self.opt.n_layers_D
        for n in range(1, nl):
            stride = 1 if n == nl - 1 else 2
            w = spectral_weight(st, "%s.model%d.0.0" % (p, n), self.training)
            x = lrelu(instance_norm(F.conv2d(x, w, None, stride=stride, padding=2)))
            outs.append(x)
        q = "%s.model%d.0" % (p, nl)
        x = F.conv2d(x, st[q + ".weight"], st[q + ".bias"], stride=1, padding=2)
        outs.append(x)
        return outs

    def d_forward(self, x):
        res = []
        for i in range(self.opt.num_D):
            res.append(self._nlayer_d(self.S["D"], "discriminator_%d" % i, x))
            x = F.avg_pool2d(x, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)
        return res

    def discriminate(self, seg, fake, real):
        """sr_model.py:655-683 — one D call on cat([fake;real]) then split on N."""
        both = torch.cat([torch.cat([seg, fake], 1), torch.cat([seg, real], 1)], 0)
        out = self.d_forward(both)
        h = both.shape[0] // 2
        return [[t[:h] for t in o] for o in out], [[t[h:] for t in o] for o in out]

    # ---- VGG19 taps (architecture.py:151-181; loss.py:105-119)
    def vgg_features(self, x):
        st, feats, idx = self.S["VGG"], [], 0
        for v in VGG_CFG:
            if v == "M":
                x = F.max_pool2d(x, 2, 2)
                idx += 1
                continue
            x = F.relu(F.conv2d(x, st["features.%d.weight" % idx], st["features.%d.bias" % idx], padding=1))
            if idx + 1 in VGG_TAPS:
                feats.append(x)
            idx += 2
        return feats

    def vgg_loss(self, x, y):
        fx, fy = self.vgg_features(x), self.vgg_features(y)
        ws = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]
        loss = 0
        for w, a, b in zip(ws, fx, fy):
            loss = loss + w * F.l1_loss(a, b.detach())
        return loss

    # ---- losses (loss.py:68-99, sr_model.py:518-564)
    @staticmethod
    def hinge(preds, target_is_real, for_d):
        total = 0
        for p in preds:
            x = p[-1]
            if for_d:
                mv = torch.min((x - 1) if target_is_real else (-x - 1), torch.zeros(1, dtype=x.dtype))
                l = -mv.mean()
            else:
                l = -x.mean()
            total = total + l.view(1)
        return total / len(preds)

    def generator_losses(self, data):
        opt = self.opt
        fake, _ = self.generate_fake(data)
        pf, pr = self.discriminate(data["input_semantics"], fake, data["image_hr"])
        losses = OrderedDict()
        losses["GAN"] = self.hinge(pf, True, False)
        if not opt.no_ganFeat_loss:
            fm = torch.zeros(1, dtype=fake.dtype)
            for i in range(len(pf)):
                for j in range(len(pf[i]) - 1):
                    fm = fm + F.l1_loss(pf[i][j], pr[i][j].detach()) * opt.lambda_feat / len(pf)
            losses["GAN_Feat"] = fm
        if not opt.no_vgg_loss:
            losses["VGG"] = self.vgg_loss(fake, data["image_hr"]) * opt.lambda_vgg
        return losses, fake

    def discriminator_losses(self, data):
        with torch.no_grad():
            fake, _ = self.generate_fake(data)
        fake = fake.detach().requires_grad_()
        pf, pr = self.discriminate(data["input_semantics"], fake, data["image_hr"])
        losses = OrderedDict()
        losses["D_Fake"] = self.hinge(pf, False, True)
        losses["D_Real"] = self.hinge(pr, True, True)
        return losses

    # ---- manager level (base_manager.py:28-66, trainer_manager.py:32-96)
    def preprocess(self, batch):
        opt, dt = self.opt, self.dtype
        img = batch["image"].to(dt)
        out = {
            "input_semantics": onehot_labels(batch["label"], opt.label_nc, dt),
            "image_lr": bicubic_down(img, opt.start_size),
            "image_hr": img,
        }
        if opt.guiding_style_image:
            out["guiding_image"] = batch["guiding_image"].to(dt)
            out["guiding_label"] = onehot_labels(batch["guiding_label"], opt.label_nc, dt)
        return out

    def params(self, net):
        return [(k, v) for k, v in self.S[net].items() if v.requires_grad]

    def create_optimizers(self):
        """sr_model.py:469-495."""
        opt = self.opt
        g_main = [v for _, v in self.params("SR")] + [v for k, v in self.params("E") if "mini" not in k]
        g_low = [v for k, v in self.params("E") if "mini" in k]
        lr_g, lr_d = (opt.lr, opt.lr) if opt.no_TTUR else (opt.lr / 2, opt.lr * 2)
        self.opt_G = torch.optim.Adam([{"params": g_main}, {"params": g_low, "lr": lr_g / 4}],
                                      lr=lr_g, betas=(opt.beta1, opt.beta2))
        self.opt_D = torch.optim.Adam([v for _, v in self.params("D")], lr=lr_d, betas=(opt.beta1, opt.beta2))
        return self.opt_G, self.opt_D

    def all_params(self):
        return [v for n in ("SR", "E", "D") for _, v in self.params(n)]

    def run_generator_one_step(self, batch):
        if self.opt_G is None:
            self.create_optimizers()
        self.opt_G.zero_grad()
        losses, fake = self.generator_losses(self.preprocess(batch))
        sum(losses.values()).mean().backward()
        if self.opt.gradient_clip > 0:
            torch.nn.utils.clip_grad_value_(self.all_params(), self.opt.gradient_clip)
        self.opt_G.step()
        self.g_losses, self.generated = losses, fake
        return losses, fake

    def run_discriminator_one_step(self, batch):
        if self.opt_D is None:
            self.create_optimizers()
        self.opt_D.zero_grad()
        losses = self.discriminator_losses(self.preprocess(batch))
        sum(losses.values()).mean().backward()
        if self.opt.gradient_clip > 0:
            torch.nn.utils.clip_grad_value_(self.all_params(), self.opt.gradient_clip)
        self.opt_D.step()
        self.d_losses = losses
        return losses

    def update_learning_rate(self, epoch):
        """trainer_manager.py:76-96: linear decay of lr after opt.niter epochs (one step of lr/niter_decay per call),
        TTUR split lr_G = lr/2, lr_D = 2*lr unless no_TTUR; EVERY param group of both optimizers gets the new value
        (the 'mini' encoder group loses its /4 at the first decay, as in the reference)."""
        opt = self.opt
        if self.opt_G is None:
            self.create_optimizers()
        if not hasattr(self, "old_lr"):
            self.old_lr = opt.lr
        new_lr = self.old_lr - opt.lr / opt.niter_decay if epoch > opt.niter else self.old_lr
        if new_lr != self.old_lr:
            lr_g, lr_d = (new_lr, new_lr) if opt.no_TTUR else (new_lr / 2, new_lr * 2)
            for g in self.opt_D.param_groups:
                g["lr"] = lr_d
            for g in self.opt_G.param_groups:
                g["lr"] = lr_g
            self.old_lr = new_lr

    def inference(self, batch):
        """sr_model.py:85-91 — eval mode: running-stat BN, no SN iteration, no noise."""
        was = self.training
        self.training = False
        try:
            with torch.no_grad():
                fake, _ = self.generate_fake(self.preprocess(batch), no_noise=True)
        finally:
            self.training = was
        return fake


    def encode_only(self, batch):
        """sr_model.py:92-99 (`encode_only`, as the inference / demo managers call it: eval mode): the style matrix
        [N, label_nc, regional_style_size] of the style image without corruption noise."""
        was = self.training
        self.training = False
        try:
            with torch.no_grad():
                style = self.encode_style(self.preprocess(batch), no_noise=True)
        finally:
            self.training = was
        return style

    def demo(self, batch, encoded_style):
        """sr_model.py:100-108 (`demo`): the generator alone on an explicit style matrix (eval mode, no noise)."""
        was = self.training
        self.training = False
        try:
            with torch.no_grad():
                data = self.preprocess(batch)
                fake = self.sr_forward(data["image_lr"], data["input_semantics"], encoded_style)
        finally:
            self.training = was
        return fake


# --------------------------------------------------------------------------- synthetic batches
def synthetic_batch(opt, n, seed=1234, guided=None):
    """blocky 19-class label map (16x16 cells, nearest-upsampled) and
    uniform [-1,1] image, all seeded."""
    g = torch.Generator().manual_seed(seed)
    h = opt.crop_size
    cells = min(16, h)

    def lab():
        c = torch.randint(0, opt.label_nc, (n, 1, cells, cells), generator=g).float()
        return F.interpolate(c, size=(h, h), mode="nearest")

    batch = {"label": lab(), "image": torch.rand(n, 3, h, h, generator=g) * 2 - 1}
    if guided if guided is not None else opt.guiding_style_image:
        batch["guiding_label"] = lab()
        batch["guiding_image"] = torch.rand(n, 3, h, h, generator=g) * 2 - 1
    return batch
