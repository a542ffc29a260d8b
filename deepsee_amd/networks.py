"""Networks of the DeepSEE train-step hot path on the HIP ops.

Module/parameter names reproduce the reference ``state_dict()`` keys exactly so
``{epoch}_net_{SR,D,E}.pth`` checkpoints interchange; the computation is re-designed for MI355X:
NHWC fp32, uint8 label maps instead of one-hot tensors, gamma/beta convolutions fused with the
sync-free BatchNorm + modulation + LeakyReLU in one MFMA kernel, upsample/noise/residual folded
into producers/consumers.

Reference: deepsee_models/networks/{sr,architecture,normalization,encoder,discriminator,loss}.py.
"""
import math

import torch
import torch.nn as nn

from . import lib as L
from . import ops

NHIDDEN = ops.NHIDDEN


# ------------------------------------------------------------------------------------ parameter holders
class Holder(nn.Module):
    """Pure container used to reproduce nn.Sequential-style key paths ('mlp_shared.0.weight')."""


def attach(root, dotted, module):
    parts = dotted.split(".")
    cur = root
    for p in parts[:-1]:
        if not hasattr(cur, p):
            cur.add_module(p, Holder())
        cur = getattr(cur, p)
    cur.add_module(parts[-1], module)
    return module


class ConvP(nn.Module):
    def __init__(self, cout, cin, k, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None


class SNConvP(nn.Module):
    """Parameters of spectral_norm(nn.Conv2d): weight_orig (+bias), buffers weight_u / weight_v."""

    def __init__(self, cout, cin, k, bias):
        super().__init__()
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout))
        else:
            self.bias = None
        self.weight_orig = nn.Parameter(torch.zeros(cout, cin, k, k))
        self.register_buffer("weight_u", torch.zeros(cout))
        self.register_buffer("weight_v", torch.zeros(cin * k * k))

    _pre = None   # this forward's normalised weight when the network ran its layers as one group (ops.SNGroup)

    def weight(self, power_iter):
        if self._pre is not None:
            w, self._pre = self._pre, None
            return w
        return ops.SpectralNorm.apply(self.weight_orig, self.weight_u, self.weight_v, bool(power_iter))


class BNStats(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long))  # never incremented


class VecP(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(n))


class Conv1dP(nn.Module):
    """style_conv = nn.Conv1d(19, 19, 1): defined by the reference, never used (normalization.py:156)."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(19, 19, 1))
        self.bias = nn.Parameter(torch.zeros(19))


# ------------------------------------------------------------------------------------ noise / randomness
EPOCH_STRIDE = 1 << 36          # stream positions reserved per forward (a forward draws < 2^31 float4s)
_COIN_TAGS = {"enc_full": 0, "enc_noise": 1}
_REGISTERED_EPOCH = None


class DeviceNoise:
    """Production source of the path's random draws: Philox counter RNG on the device for N(0,1)/U(0,1)
    tensors; the two per-forward branch coins (sr_model.py:616,643: python ``random`` in the reference) are a pure
    function of (coin_seed, forward index, tag).  Under data-parallel training every rank must take the same
    encoder branch (the unused branch gets no gradient): `coin_seed` is therefore the same on all ranks
    (it never depends on the rank, unlike `seed`, which parallel.attach offsets per rank for the noise tensors), and
    nothing else in the process -- a dataset drawing from the global ``random``, say -- can desynchronise it.  Being
    stateless, the coins of the NEXT forward can be looked at before it runs (the hipGraph variant to replay is chosen
    by them, managers.TrainerManager)."""

    def __init__(self, seed=0, coin_seed=None):
        self.seed, self.offset, self.step = int(seed), 0, 0
        self.coin_seed = int(seed) if coin_seed is None else int(coin_seed)
        # The device-side epoch every Philox stream drawn through this object is offset by (dsee_rng_set_epoch): one int64
        # in HBM, advanced by a device-side add at the start of every training forward -- so that a captured hipGraph,
        # whose kernel arguments (seed, offset) are frozen, still draws fresh noise on every replay.
        self.epoch = torch.zeros(1, dtype=torch.int64, device="cuda")
        self._register()

    def _register(self):
        """Tell the library which epoch the kernels launched from now on read (a captured graph keeps the pointer it was
        captured with).  The registered tensor is pinned by a module-level reference: the library must never be left
        with the address of freed memory."""
        import ctypes
        from . import lib as L
        global _REGISTERED_EPOCH
        _REGISTERED_EPOCH = self.epoch
        L.lib().dsee_rng_set_epoch(ctypes.c_void_p(self.epoch.data_ptr()))

    def ensure_registered(self):
        """Re-register this object's epoch if another DeviceNoise (a second model in the process) registered its own since:
        called by every consumer of a Philox stream drawn here (ops.PhiloxNormal.bind), forward and backward."""
        if _REGISTERED_EPOCH is not self.epoch:
            self._register()

    def begin_step(self):
        """Start of a training forward: stream offsets restart at 0 and the device epoch advances."""
        self.step += 1
        self.offset = 0
        self.epoch.add_(EPOCH_STRIDE)
        self._register()

    def state_dict(self):
        """What a resumed run needs to CONTINUE the branch-coin and noise sequences instead of repeating them from the start
        of training (saved next to the checkpoints, SRModel.save)."""
        return {"step": int(self.step), "epoch": int(self.epoch.item())}

    def load_state_dict(self, state):
        self.step = int(state["step"])
        self.epoch.fill_(int(state["epoch"]))

    def coin(self, tag, step=None):
        import random as _r
        s = self.step if step is None else int(step)
        return _r.Random((self.coin_seed * 2654435761 + s) * 4 + _COIN_TAGS[tag]).random()

    def _fill(self, shape_nhwc, normal):
        n = 1
        for s in shape_nhwc:
            n *= s
        t = ops.rng_fill(tuple(shape_nhwc), self.seed, self.offset, normal)
        self.offset += (n + 3) // 4
        return t

    def normal_nhwc(self, shape_nhwc, tag):
        """The noise-injection draws are consumed lazily (ops.PhiloxNormal): same stream positions as _fill."""
        n = 1
        for s in shape_nhwc:
            n *= s
        assert n % 4 == 0
        t = ops.PhiloxNormal(shape_nhwc, self.seed, self.offset, self)
        self.offset += n // 4
        return t

    def uniform(self, shape, tag):
        return self._fill(shape, False)

    def normal(self, shape, tag):
        return self._fill(shape, True)


class ReplayNoise:
    """Test-time source: replays a tape recorded from the oracle (oracle.RecordingCtl) so both sides see the
    same noise tensors and branch decisions."""

    def __init__(self, tape):
        self.tape, self.pos = list(tape), 0

    def begin_step(self):
        pass

    def _next(self, kind, tag):
        k, t, v = self.tape[self.pos]
        assert k == kind and t == tag, "replay mismatch: want %s/%s, tape has %s/%s" % (kind, tag, k, t)
        self.pos += 1
        return v

    def coin(self, tag):
        return self._next("coin", tag)

    def normal_nhwc(self, shape_nhwc, tag):
        v = self._next("normal", tag)  # NCHW cpu
        t = ops.to_nhwc(v.cuda(), shape_nhwc[3])
        assert tuple(t.shape) == tuple(shape_nhwc)
        return t

    def uniform(self, shape, tag):
        v = self._next("uniform", tag)
        assert tuple(v.shape) == tuple(shape)
        return v.cuda()

    def normal(self, shape, tag):
        v = self._next("normal", tag)
        assert tuple(v.shape) == tuple(shape)
        return v.cuda()


# ------------------------------------------------------------------------------------ SPADE / SEAN / PureSEAN
class SpadeNorm(nn.Module):
    """normalization.py: SPADE (:71-120), SEAN_Block (:123-213), PureSEAN_Block (:216-286) followed by the
    resblock's LeakyReLU (architecture.py:92,114)."""

    def __init__(self, kind, c, label_nc, style_size, max_fm_size):
        super().__init__()
        self.kind, self.c, self.max_fm = kind, c, max_fm_size
        self.param_free_norm = BNStats(c)
        attach(self, "mlp_shared.0", ConvP(NHIDDEN, label_nc, 3))
        if kind in ("spade", "sean"):
            self.mlp_gamma = ConvP(c, NHIDDEN, 3)
            self.mlp_beta = ConvP(c, NHIDDEN, 3)
        if kind in ("sean", "puresean"):
            self.style_conv = Conv1dP()
            self.mlp_style_gamma = ConvP(c, style_size, 3)
            self.mlp_style_beta = ConvP(c, style_size, 3)
        if kind == "sean":
            self.alpha_beta = nn.Parameter(torch.rand(1))
            self.alpha_gamma = nn.Parameter(torch.rand(1))

    def forward(self, x, labels, style, training, grad_sink=None):
        """`grad_sink` (ops.GradSink): the gradient of the other consumer of x (the resblock shortcut) is added to dx
        inside this norm's backward pass."""
        n, h, w, c = x.shape
        fm = h if self.kind == "spade" else min(h, self.max_fm)  # SPADE.forward has no fm cap
        sh = self.mlp_shared._modules["0"]
        st = self.param_free_norm
        capped = fm != h
        if capped and (h * w) % 128 == 0 and c % 64 == 0:
            return self._forward_capped(x, labels, training, fm, grad_sink)
        if capped or (self.kind != "spade" and (h * w) % 128 != 0):
            return self._forward_dense(x, labels, style, training, fm, grad_sink)
        shift = labels.shift_for(h)
        # ---- table path: one fused autograd node per norm; its packed / blended weight set comes from ONE kernel
        P = ops.SeanPack.apply
        am = ops.amax_slot()     # max |packed weights| (and |style table|), written by the producers themselves
        if self.kind == "spade":
            w2a, _, b2 = P(0, self.mlp_gamma.weight, self.mlp_beta.weight, None, None, self.mlp_gamma.bias,
                           self.mlp_beta.bias, None, None, None, None, am)
            w2a.dsee_amax = am
            return ops.SeanNormTable.apply(x, sh.weight, sh.bias, w2a, None, b2, st.running_mean, st.running_var,
                                           labels, shift, training, 1.0, grad_sink)
        if self.kind == "sean":
            w2a, wst, b2 = P(1, self.mlp_gamma.weight, self.mlp_beta.weight, self.mlp_style_gamma.weight,
                             self.mlp_style_beta.weight, self.mlp_gamma.bias, self.mlp_beta.bias,
                             self.mlp_style_gamma.bias, self.mlp_style_beta.bias, self.alpha_gamma, self.alpha_beta, am)
            w2a.dsee_amax = am
            table = ops.style_table_packed(style, wst, b2.shape[0], am)
            return ops.SeanNormTable.apply(x, sh.weight, sh.bias, w2a, table, b2, st.running_mean, st.running_var,
                                           labels, shift, training, 1.0, grad_sink)
        # puresean: out = xhat * gamma_s + beta_s
        _, wst, b2 = P(2, None, None, self.mlp_style_gamma.weight, self.mlp_style_beta.weight, None, None,
                       self.mlp_style_gamma.bias, self.mlp_style_beta.bias, None, None, None)
        table = ops.style_table_packed(style, wst, b2.shape[0], am)
        return ops.SeanNormTable.apply(x, None, None, None, table, b2, st.running_mean, st.running_var, labels, shift,
                                       training, 0.0, grad_sink)

    def _forward_capped(self, x, labels, training, fm, grad_sink=None):
        """The reference's max_fm_size cap (normalization.py:171-190, 258-277) on the fused / Winograd path: above the
        cap BOTH the SPADE embedding and the "style map" are the embedding computed at the capped resolution and
        nearest-upsampled (the style matrix is ignored; it only type-checks because nhidden == style size), so the two
        weight sets act on the same 128 channels and fold into one: gamma/beta = conv(up(actv), W_folded)."""
        n, h, w, c = x.shape
        sh, st = self.mlp_shared._modules["0"], self.param_free_norm
        ups = int(round(math.log2(h // fm)))
        am = ops.amax_slot()
        if self.kind == "sean":
            w2a, _, b2 = ops.SeanPack.apply(3, self.mlp_gamma.weight, self.mlp_beta.weight, self.mlp_style_gamma.weight,
                                            self.mlp_style_beta.weight, self.mlp_gamma.bias, self.mlp_beta.bias,
                                            self.mlp_style_gamma.bias, self.mlp_style_beta.bias, self.alpha_gamma,
                                            self.alpha_beta, am)
            add_one = 1.0
        else:  # puresean: out = xhat * gamma_s + beta_s
            w2a, _, b2 = ops.SeanPack.apply(0, self.mlp_style_gamma.weight, self.mlp_style_beta.weight, None, None,
                                            self.mlp_style_gamma.bias, self.mlp_style_beta.bias, None, None, None, None, am)
            add_one = 0.0
        w2a.dsee_amax = am
        return ops.SeanNormTable.apply(x, sh.weight, sh.bias, w2a, None, b2, st.running_mean, st.running_var, labels,
                                       labels.shift_for(fm), training, add_one, grad_sink, ups)

    def _forward_dense(self, x, labels, style, training, fm, grad_sink=None):
        """General path (style map materialised as 128 gathered channels): resolutions below 16x16, and the
        reference's max_fm_size cap where the upsampled embedding replaces the style map."""
        n, h, w, c = x.shape
        shift = labels.shift_for(fm)
        cat_ups = 0
        sh = self.mlp_shared._modules["0"]
        capped = fm != h
        if capped:
            # normalization.py:188-190 / 275-277: actv AND style_map both become the nearest-upsampled SPADE
            # activation (style is ignored; only type-checks because nhidden == style size).
            cat_ups = int(round(math.log2(h // fm)))
            actv = ops.SeanInput.apply(sh.weight, sh.bias, None, labels, shift, True, False)
        if self.kind == "spade":
            cat = ops.SeanInput.apply(sh.weight, sh.bias, None, labels, shift, True, False)
            w2, b2 = ops.pack_gamma_beta(self.mlp_gamma.weight, self.mlp_beta.weight, self.mlp_gamma.bias,
                                         self.mlp_beta.bias)
            add_one = 1.0
        elif self.kind == "sean":
            wg, wb = torch.sigmoid(self.alpha_gamma), torch.sigmoid(self.alpha_beta)
            if capped:
                # both halves of the concatenated input are `actv`: fold the two weight sets together
                wgam = (1.0 - wg) * self.mlp_gamma.weight + wg * self.mlp_style_gamma.weight
                wbet = (1.0 - wb) * self.mlp_beta.weight + wb * self.mlp_style_beta.weight
                cat = actv
            else:
                cat = ops.SeanInput.apply(sh.weight, sh.bias, style, labels, shift, True, True)
                wgam = torch.cat([(1.0 - wg) * self.mlp_gamma.weight, wg * self.mlp_style_gamma.weight], 1)
                wbet = torch.cat([(1.0 - wb) * self.mlp_beta.weight, wb * self.mlp_style_beta.weight], 1)
            bg = (1.0 - wg) * self.mlp_gamma.bias + wg * self.mlp_style_gamma.bias
            bb = (1.0 - wb) * self.mlp_beta.bias + wb * self.mlp_style_beta.bias
            w2, b2 = ops.pack_gamma_beta(wgam, wbet, bg, bb)
            add_one = 1.0
        else:  # puresean: out = xhat * gamma_s + beta_s
            cat = actv if capped else ops.SeanInput.apply(sh.weight, sh.bias, style, labels, shift, False, True)
            w2, b2 = ops.pack_gamma_beta(self.mlp_style_gamma.weight, self.mlp_style_beta.weight,
                                         self.mlp_style_gamma.bias, self.mlp_style_beta.bias)
            add_one = 0.0
        st = self.param_free_norm
        return ops.SpadeNormAct.apply(x, cat, w2, b2, st.running_mean, st.running_var, training, add_one, cat_ups,
                                      grad_sink)


class SPADEResnetBlock(nn.Module):
    """architecture.py:24-147 with fin == fout (identity shortcut)."""

    def __init__(self, c, opt, kind):
        super().__init__()
        self.kind, self.add_noise_cfg = kind, bool(opt.add_noise)
        self.conv_0 = SNConvP(c, c, 3, True)
        self.conv_1 = SNConvP(c, c, 3, True)
        self.norm_0 = SpadeNorm(kind, c, opt.semantic_nc, opt.regional_style_size, opt.max_fm_size)
        self.norm_1 = SpadeNorm(kind, c, opt.semantic_nc, opt.regional_style_size, opt.max_fm_size)
        if self.add_noise_cfg:
            self.noise_in, self.noise_skip, self.noise_middle = VecP(c), VecP(c), VecP(c)

    def forward(self, x, labels, style, noise, tag, ups, training, out_act=L.ACT_NONE, defer_act_bwd=False):
        """`defer_act_bwd`: the block's only consumer applies the backward of `out_act` (ops.Conv2d.forward)."""
        noisy = self.add_noise_cfg and training
        n, h0, w0, c = x.shape
        shp = (n, h0 << ups, w0 << ups, c)
        res_noise = None
        if noisy:
            x = ops.UpNoise.apply(x, self.noise_in.weight, noise.normal_nhwc(shp, tag + ".noise_in"), ups, training)
            # the shortcut x_s = noise_skip(x) (architecture.py:133-134) is not materialised: conv_1's output transform
            # adds x + w_skip * eps_skip (a replayed noise TENSOR, as in the parity tests, takes the explicit pass)
            res_noise = (self.noise_skip.weight, noise.normal_nhwc(shp, tag + ".noise_skip"))
        elif ups:
            x = ops.UpNoise.apply(x, None, None, ups)
        # x feeds norm_0 and the shortcut: the shortcut's gradient reaches x through norm_0's backward (ops.GradSink)
        sink = ops.GradSink() if (torch.is_grad_enabled() and x.requires_grad) else None
        h = self.norm_0(x, labels, style, training, sink)
        # noise_middle (architecture.py:111-112) rides in conv_0's output transform
        dx = ops.conv2d(h, self.conv_0.weight(training), self.conv_0.bias,
                        noise=(self.noise_middle.weight, noise.normal_nhwc(shp, tag + ".noise_middle")) if noisy else None,
                        stats=training)
        h = self.norm_1(dx, labels, style, training)
        return ops.conv2d(h, self.conv_1.weight(training), self.conv_1.bias, res=x, act=out_act, res_noise=res_noise,
                          res_sink=sink, defer_act_bwd=defer_act_bwd)


class DeepSEESR(nn.Module):
    """sr.py:10-98."""

    def __init__(self, opt, plan):
        super().__init__()
        c = 16 * opt.ngf
        self.c = c
        self.initial = ConvP(c, 3, 3)
        self.head_0 = SPADEResnetBlock(c, opt, plan[0][1])
        self.G_middle_0 = SPADEResnetBlock(c, opt, plan[1][1])
        self.G_middle_1 = SPADEResnetBlock(c, opt, plan[2][1])
        self.up_list = nn.ModuleList([SPADEResnetBlock(c, opt, k) for _, k in plan[3:]])
        self.conv_img = ConvP(3, c, 3)
        self._sn = None

    def forward(self, image_lr, labels, style, noise, training):
        blocks = [("head_0", self.head_0, 0), ("G_middle_0", self.G_middle_0, 1), ("G_middle_1", self.G_middle_1, 0)]
        blocks += [("up_list.%d" % i, b, 1) for i, b in enumerate(self.up_list)]
        if self._sn is None:      # all spectral-normalised convolutions of the generator: one group launch per forward
            self._sn = ops.SNGroup([c for _, b, _ in blocks for c in (b.conv_0, b.conv_1)])
        self._sn.run(training)
        self._sn.wino_weights(torch.is_grad_enabled())
        x = ops.conv2d(image_lr, self.initial.weight, self.initial.bias)
        for i, (tag, blk, ups) in enumerate(blocks):
            # the LeakyReLU in front of conv_img (sr.py:94) rides in the last block's epilogue
            # ... and its BACKWARD in conv_img's data-gradient kernel (x has no other consumer): ops.Conv2d.forward
            last = i == len(blocks) - 1
            defer = last and torch.is_grad_enabled() and x.requires_grad and ops.P().defer_act
            x = blk(x, labels, style, noise, tag, ups, training, L.ACT_LRELU if last else L.ACT_NONE, defer)
        return ops.conv2d(x, self.conv_img.weight, self.conv_img.bias, act=L.ACT_TANH, in_act=L.ACT_LRELU if defer else 0)


# ------------------------------------------------------------------------------------ style encoders
class EncBranch(nn.Module):
    """conv(SN, no bias) + InstanceNorm + LeakyReLU stacks of encoder.py:83-99 (full) / :142-158 (mini)."""

    def __init__(self, names, strides, ups, nf, cin):
        super().__init__()
        chans = [(nf, cin), (2 * nf, nf), (4 * nf, 2 * nf), (8 * nf, 4 * nf)]
        self.names, self.strides, self.ups = names, strides, ups
        for nm, (co, ci) in zip(names, chans):
            attach(self, nm, SNConvP(co, ci, 3, False))

    def layer(self, nm):
        m = self
        for p in nm.split("."):
            m = m._modules[p]
        return m

    def forward_main(self, x, training):
        for nm, s, u in zip(self.names, self.strides, self.ups):
            x = ops.conv2d(x, self.layer(nm).weight(training), None, stride=s, pad=1, ups=u)
            x = ops.InstNormAct.apply(x, L.ACT_LRELU)
        return x


FULL_NAMES = ["initial.0.0", "down0.0.0", "down1.0.0", "up_conv.1.0"]
MINI_NAMES = ["initial.0.0", "conv0.0.0", "conv1.0.0", "conv2.1.0"]


class StyleEncoder(nn.Module):
    """CombinedstyleEncoder (encoder.py:178-210) or FullStyleEncoder (:73-132)."""

    def __init__(self, opt):
        super().__init__()
        nf, s = opt.nef, opt.regional_style_size
        self._sn = {}
        self.combined = opt.netE == "combinedstyle"
        self.scale = opt.noisy_style_scale
        self.dist = opt.noisy_style_dist
        if self.scale > 0:
            self.noise_weights = nn.Parameter(torch.zeros(opt.label_nc))
        attach(self, "final.0.0", SNConvP(s, 8 * nf, 3, False))
        if self.combined:
            self.encoder_full = EncBranch(FULL_NAMES, [1, 2, 2, 1], [0, 0, 0, 1], nf, 3)
            attach(self.encoder_full, "final.0.0", SNConvP(s, 8 * nf, 3, False))  # constructed, unused
            self.encoder_mini = EncBranch(MINI_NAMES, [1, 1, 1, 1], [0, 0, 0, 1], nf, 3)
            attach(self.encoder_mini, "final.0.0", SNConvP(s, 8 * nf, 3, False))  # constructed, unused
        elif opt.netE == "fullstyle":
            # FullStyleEncoder keeps its layers at the root of the state dict
            chans = [(nf, 3), (2 * nf, nf), (4 * nf, 2 * nf), (8 * nf, 4 * nf)]
            for nm, (co, ci) in zip(FULL_NAMES, chans):
                attach(self, nm, SNConvP(co, ci, 3, False))
        else:
            raise NotImplementedError("netE=%s (ministyle crashes in the reference too)" % opt.netE)

    def _root_layer(self, nm):
        m = self
        for p in nm.split("."):
            m = m._modules[p]
        return m

    def forward(self, x, labels, mode, no_noise, noise, training):
        # the spectral-normalised layers THIS forward uses (the other branch's u / v must not advance: the reference's
        # hook only runs for modules that are called), as one group launch
        if mode not in self._sn:
            if self.combined:
                br = self.encoder_full if mode == "full" else self.encoder_mini
                layers = [br.layer(nm) for nm in br.names]
            else:
                layers = [self._root_layer(nm) for nm in FULL_NAMES]
            self._sn[mode] = ops.SNGroup(layers + [self._root_layer("final.0.0")])
        self._sn[mode].run(training)
        if self.combined:
            x = (self.encoder_full if mode == "full" else self.encoder_mini).forward_main(x, training)
        else:
            for nm, s, u in zip(FULL_NAMES, [1, 2, 2, 1], [0, 0, 0, 1]):
                x = ops.conv2d(x, self._root_layer(nm).weight(training), None, stride=s, pad=1, ups=u)
                x = ops.InstNormAct.apply(x, L.ACT_LRELU)
        fin = self._root_layer("final.0.0")
        x = ops.conv2d(x, fin.weight(training), None)
        x = ops.InstNormAct.apply(x, L.ACT_TANH)
        sm = ops.StylePool.apply(x, labels, labels.shift_for(x.shape[1]))
        if self.scale > 0 and not no_noise:
            # encoder.py:51-70 on the [N,19,S] style matrix (KB-sized parameter-space glue)
            nw = torch.sigmoid(self.noise_weights)[None, :, None]
            if self.dist == "uniform":
                z = noise.uniform(tuple(sm.shape), "style_noise")
            elif self.dist == "normal":
                z = noise.normal(tuple(sm.shape), "style_noise")
            else:
                raise ValueError("Does not exist: {}".format(self.dist))      # encoder.py:66
            sm = (sm + ((z * 2 - 1) * self.scale) * nw).clamp(-1, 1)
        return sm


# ------------------------------------------------------------------------------------ discriminator
class NLayerD(nn.Module):
    """discriminator.py:67-120."""

    def __init__(self, opt):
        super().__init__()
        nf = opt.ndf
        cin = opt.label_nc + opt.output_nc + (1 if opt.contain_dontcare_label else 0)
        self.nl = opt.n_layers_D
        attach(self, "model0.0", ConvP(nf, cin, 4))
        for n in range(1, self.nl):
            prev, nf = nf, min(nf * 2, 512)
            attach(self, "model%d.0.0" % n, SNConvP(nf, prev, 4, False))
        attach(self, "model%d.0" % self.nl, ConvP(1, nf, 4))

    def forward(self, x, training, x_detached=None):
        """`x_detached` (round 6): a second batch segment that needs no gradient -- the real images of the GENERATOR step, whose
        features enter the feature-matching loss detached (sr_model.py:547-564).  The reference stacks it on N behind the
        generated images (sr_model.py:655-668); the discriminator has no batch statistics (InstanceNorm is per sample, SURVEY
        B-8), so running it as its own no-grad pass on the SAME weights of this forward (one spectral-norm power iteration) gives
        the same numbers and spares the backward pass the data gradients of a batch half whose upstream gradient is zero."""
        m0 = self.model0._modules["0"]
        mids = [getattr(self, "model%d" % n)._modules["0"]._modules["0"].weight(training) for n in range(1, self.nl)]
        ml = getattr(self, "model%d" % self.nl)._modules["0"]

        def run(x):
            outs = []
            x = ops.conv2d(x, m0.weight, m0.bias, stride=2, pad=2, act=L.ACT_LRELU)
            outs.append(x)
            for n in range(1, self.nl):
                x = ops.conv2d(x, mids[n - 1], None, stride=1 if n == self.nl - 1 else 2, pad=2)
                x = ops.InstNormAct.apply(x, L.ACT_LRELU)
                outs.append(x)
            outs.append(ops.conv2d(x, ml.weight, ml.bias, stride=1, pad=2))
            return outs
        if x_detached is None:
            return run(x)
        with torch.no_grad():
            det = run(x_detached)
        return run(x), det


class MultiscaleDiscriminator(nn.Module):
    """discriminator.py:14-63."""

    def __init__(self, opt):
        super().__init__()
        self.num_d = opt.num_D
        self._sn = None
        for i in range(opt.num_D):
            self.add_module("discriminator_%d" % i, NLayerD(opt))

    def forward(self, x, training, x_detached=None):
        """Returns the per-scale feature lists; with `x_detached` a pair (features of x, features of x_detached): see NLayerD."""
        if self._sn is None:
            self._sn = ops.SNGroup([getattr(getattr(self, "discriminator_%d" % i), "model%d" % n)._modules["0"]._modules["0"]
                                    for i in range(self.num_d)
                                    for n in range(1, getattr(self, "discriminator_%d" % i).nl)])
        self._sn.run(training)
        # the scales are independent given x (and its pooled copies): one branch per scale (ops.branches)
        xs, ds = [x], [x_detached]
        for i in range(1, self.num_d):
            xs.append(ops.AvgPool3s2.apply(xs[-1]))
            if x_detached is not None:
                with torch.no_grad():
                    ds.append(ops.AvgPool3s2.apply(ds[-1]))
            else:
                ds.append(None)
        # (tensors made on this stream that the side branches read, forward and backward: the pooled inputs and the
        # spectral-normalised weights W / sigma of this forward)
        shared = xs[1:] + [t for t in ds[1:] if t is not None] + list(self._sn.last)   # (SNGroup.last: the buffer all W / sigma are slices of + their maxima)
        res = ops.branches(*[(lambda i=i: getattr(self, "discriminator_%d" % i)(xs[i], training, ds[i])) for i in range(self.num_d)],
                           inputs=shared)
        if x_detached is None:
            return res
        return [r[0] for r in res], [r[1] for r in res]


# ------------------------------------------------------------------------------------ VGG19 perceptual taps
VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512]
VGG_TAPS = (1, 6, 11, 20, 29)
VGG_WEIGHTS = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]


class VGG19Taps(nn.Module):
    """architecture.py:151-181 (frozen).  Keys follow torchvision's vgg19().features indices."""

    def __init__(self):
        super().__init__()
        self.features = Holder()
        cin, idx = 3, 0
        for v in VGG_CFG:
            if v == "M":
                idx += 1
                continue
            self.features.add_module(str(idx), ConvP(v, cin, 3))
            cin = v
            idx += 2
        for p in self.parameters():
            p.requires_grad = False
            p.dsee_frozen = True     # ops._frozen_cache: packed weights and their maxima are built once, not per forward

    def forward(self, x):
        feats, idx = [], 0
        for v in VGG_CFG:
            if v == "M":
                x = ops.MaxPool2.apply(x)
                idx += 1
                continue
            m = self.features._modules[str(idx)]
            x = ops.conv2d(x, m.weight, m.bias, act=L.ACT_RELU)
            if idx + 1 in VGG_TAPS:
                feats.append(x)
            idx += 2
        return feats
