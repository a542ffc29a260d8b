"""SRModel — the model facade of the hot path, same surface as the reference's
deepsee_models/sr_model.py::SRModel: attributes netSR / netD / netE / opt / model_variant / logs /
last_encoded_style_is_full / last_encoded_style_is_noisy; forward(data, mode) with modes 'generator', 'encode_only', 'demo',
'discriminator', 'inference' (anything else raises ValueError like sr_model.py:445-446); create_optimizers(opt);
save(epoch) / load_weights().  All activation-space compute runs in libdeepsee_hip.so.
"""
import math
import os
import sys
import warnings
from collections import OrderedDict

import torch
import torch.nn as nn

from . import lib as L
from . import networks as N
from . import ops
from .optim import FlatAdam
from .plan import from_opt as plan_from_opt


def block_plan(opt):
    """sr.py:27-51: head_0 SPADE ('late' in norm_G), SEAN blocks, PureSEAN tail iff
    load_size >= 512."""
    nb = int(round(math.log2(opt.crop_size) - math.log2(opt.start_size)))
    cfg = opt.norm_G.replace("spectral", "")
    sk = "sean" if "sean" in cfg else "spade"
    early = "late" not in opt.norm_G
    plan = [("head_0", sk if early else "spade"), ("G_middle_0", sk), ("G_middle_1", sk)]
    max_nb = 4 if opt.load_size >= 512 else 99
    kinds = [sk] * max(0, min(nb, max_nb) - 1)
    if max_nb != 99:
        kinds += ["puresean"] * max(0, nb - max_nb)
    return plan + [("up_list.%d" % i, k) for i, k in enumerate(kinds)]


def init_weights(net, init_type, gain, gen):
    """base_network.py:28-59 semantics on our parameter holders: xavier_normal(gain) (or kaiming / normal) on
    every conv weight incl. spectral-norm weight_orig and the unused Conv1d; biases 0; noise weights 0;
    alpha ~ U(0,1); SN u/v ~ normalised N(0,1)."""
    for name, p in list(net.named_parameters()) + list(net.named_buffers()):
        leaf = name.split(".")[-1]
        if leaf in ("running_mean", "num_batches_tracked"):
            p.data.zero_()
        elif leaf == "running_var":
            p.data.fill_(1.0)
        elif leaf in ("weight_u", "weight_v"):
            t = torch.randn(p.shape, generator=gen)
            p.data.copy_(t / t.norm().clamp_min(1e-12))
        elif leaf in ("alpha_beta", "alpha_gamma"):
            p.data.copy_(torch.rand(p.shape, generator=gen))
        elif leaf == "bias" or leaf == "noise_weights" or ".noise_" in name:
            p.data.zero_()
        elif p.dim() >= 3:
            rf = 1
            for s in p.shape[2:]:
                rf *= s
            fan_in, fan_out = p.shape[1] * rf, p.shape[0] * rf
            if init_type == "xavier":
                std = gain * math.sqrt(2.0 / (fan_in + fan_out))
            elif init_type == "kaiming":
                std = math.sqrt(2.0 / fan_in)
            elif init_type == "normal":
                std = gain
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            p.data.copy_(torch.randn(p.shape, generator=gen) * std)


def _sum_terms(terms):
    """Sum of the [1]-shaped loss terms in two launches (cat + reduce) instead of one ATen add per term."""
    if not terms:        # a discriminator without intermediate features (the reference starts its sums from 0, sr_model.py:535)
        return torch.zeros(1, device="cuda")
    return terms[0] if len(terms) == 1 else torch.cat(terms).sum(0, keepdim=True)


class SRModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        if not torch.cuda.is_available():
            raise RuntimeError("deepsee_amd.SRModel needs an MI355X (HIP) device: there is no CPU fallback")
        L.lib()  # fail loudly if libdeepsee_hip.so is missing
        self.opt = opt
        # which of the equivalent kernel paths this model's operators take, and its precision (opt.precision 'fp32' | 'fp16':
        # 16-bit matrix-core GEMMs, fp32 everything else) -- an immutable value owned by THIS model (deepsee_amd/plan.py)
        self.plan = plan_from_opt(opt)
        self.use_E = opt.netE is not None and len(opt.netE) > 0
        self.model_variant = "guided" if (self.use_E and "full" in opt.netE) else "independent"   # sr_model.py:26-30
        gen = torch.Generator().manual_seed(int(getattr(opt, "seed", 0)))
        self.netSR = N.DeepSEESR(opt, block_plan(opt))
        init_weights(self.netSR, opt.init_type, opt.init_variance, gen)
        self.netD = None
        if opt.isTrain:
            self.netD = N.MultiscaleDiscriminator(opt)
            init_weights(self.netD, opt.init_type, opt.init_variance, gen)
        self.netE = None
        if self.use_E:
            self.netE = N.StyleEncoder(opt)
            init_weights(self.netE, opt.init_type, opt.init_variance, gen)
        self.vgg = None
        if opt.isTrain and not opt.no_vgg_loss:
            # the reference downloads torchvision's pretrained VGG19 (architecture.py:154).  opt.vgg_weights names a
            # local copy of that state dict (torchvision keys 'features.N.weight/bias', or the bare 'N.weight/bias' of
            # vgg19().features); without it the taps are He-initialised random features, NOT a perceptual loss.
            self.vgg = N.VGG19Taps()
            init_weights(self.vgg, "kaiming", 1.0, gen)
            self.vgg_pretrained = False
            path = getattr(opt, "vgg_weights", None)
            if path:
                self.load_vgg_state(torch.load(path, map_location="cpu"))
            else:
                warnings.warn("deepsee_amd: no pretrained VGG19 weights (opt.vgg_weights is unset): the VGG loss term "
                              "(lambda_vgg=%g) runs on randomly initialised features.  Fine for benchmarks and parity "
                              "tests; for real training pass opt.vgg_weights=<torchvision vgg19 state dict> or call "
                              "SRModel.load_vgg_state()." % opt.lambda_vgg, RuntimeWarning, stacklevel=2)
        self.cuda()
        self.noise = N.DeviceNoise(seed=int(getattr(opt, "seed", 0)) * 7919 + 17)
        self.logs = OrderedDict()
        self.last_encoded_style_is_full = True
        self.last_encoded_style_is_noisy = False
        if not opt.isTrain or opt.continue_train:
            self.load_weights()

    # ---- reference surface
    def get_logs(self):
        return self.logs

    def use_gpu(self):
        return True

    def forward(self, data, mode, **kwargs):
        with self.plan.active():       # the backward pass re-activates it per autograd node (ops._under_plan)
            return self._forward(data, mode, **kwargs)

    def _forward(self, data, mode, **kwargs):
        d = self._native(data)
        if mode in ("generator", "discriminator"):
            self.noise.begin_step()     # fresh Philox positions / branch coins for this forward
        if mode == "generator":
            g_loss, generated = self.compute_generator_loss(d)
            self.logs["image/downsized"] = d["image_lr"]
            return g_loss, generated
        elif mode == "discriminator":
            return self.compute_discriminator_loss(d)
        elif mode == "inference":
            with torch.no_grad():
                fake, _ = self.generate_fake(d, no_noise=True)
            data["fake_image"] = ops.to_nchw(fake, 3)
            return {k: v for k, v in data.items() if v is not None}
        elif mode == "encode_only":
            # sr_model.py:92-99: the (uncorrupted) style matrix [N, label_nc, regional_style_size]
            return self.encode_style(d, no_noise=True)
        elif mode == "demo":
            # sr_model.py:100-108: the generator alone on an explicit style matrix `encoded_style`
            with torch.no_grad():
                style = data["encoded_style"].to("cuda", torch.float32).contiguous()
                fake = self.netSR(d["image_lr"], d["labels"], style, self.noise, self.training)
            data["fake_image"] = ops.to_nchw(fake, 3)
            return {k: v for k, v in data.items() if v is not None}
        else:
            raise ValueError("|mode| is invalid")

    def load_vgg_state(self, state):
        """Load torchvision's vgg19 weights into the frozen perceptual taps (architecture.py:151-181).  Accepts the
        full model's state dict ('features.N.*', classifier keys ignored) or that of vgg19().features ('N.*')."""
        own = self.vgg.state_dict()
        picked = {}
        for k in own:
            bare = k[len("features."):]
            if k in state:
                picked[k] = state[k]
            elif bare in state:
                picked[k] = state[bare]
            else:
                raise RuntimeError("VGG19 state dict lacks %s" % k)
        self.load_net_state(self.vgg, picked)
        self.vgg_pretrained = True

    def create_optimizers(self, opt):
        """sr_model.py:469-495: G = SR params + non-'mini' E params @ lr/2 (group 0), 'mini' E params @ lr/8
        (group 1); D @ 2*lr; betas (beta1, beta2).  Without a discriminator (isTrain False) opt_D is None."""
        g_main = [("SR." + k, p) for k, p in self.netSR.named_parameters()]
        g_low = []
        if self.use_E:
            for k, p in self.netE.named_parameters():
                (g_low if "mini" in k else g_main).append(("E." + k, p))
        lr_g, lr_d = (opt.lr, opt.lr) if opt.no_TTUR else (opt.lr / 2, opt.lr * 2)
        print("lr G: {}, lr D: {}".format(lr_g, lr_d), file=sys.stderr)   # (sr_model.py:112 prints this; keep stdout clean)
        groups = [{"params": g_main, "lr": lr_g}]
        if g_low:
            groups.append({"params": g_low, "lr": lr_g / 4})
        opt_g = FlatAdam(groups, betas=(opt.beta1, opt.beta2))
        opt_d = None
        if self.netD is not None:
            opt_d = FlatAdam([{"params": [("D." + k, p) for k, p in self.netD.named_parameters()], "lr": lr_d}],
                             betas=(opt.beta1, opt.beta2))
        return opt_g, opt_d

    def _ckpt(self, label, epoch):
        return os.path.join(self.opt.checkpoints_dir, self.opt.name, "%s_net_%s.pth" % (epoch, label))

    def save(self, epoch):
        """util/util.py:217-226: {epoch}_net_{SR,D,E}.pth = {"model": state_dict} with the reference's keys.
        Data-parallel: only rank 0 writes (parameters and spectral-norm buffers are identical on every rank; the
        sync-free BatchNorm running statistics saved are rank 0's shard's, as in the reference DP where only the master
        replica keeps them, sync_batchnorm/batchnorm.py:128-145).  Files are written to a temporary name and renamed,
        so a reader never sees a torn checkpoint."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return
        os.makedirs(os.path.join(self.opt.checkpoints_dir, self.opt.name), exist_ok=True)
        for label, net in (("SR", self.netSR), ("D", self.netD), ("E", self.netE)):
            if net is not None:
                path = self._ckpt(label, epoch)
                tmp = "%s.tmp.%d" % (path, os.getpid())
                torch.save({"model": OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())}, tmp)
                os.replace(tmp, path)
        if hasattr(self.noise, "state_dict"):
            # not part of the reference's layout (its noise is torch's global generator, never saved): the position of the
            # Philox / branch-coin sequences, so that --continue_train continues them instead of replaying them from step 0
            path = self._ckpt("noise_state", epoch)
            tmp = "%s.tmp.%d" % (path, os.getpid())
            torch.save(self.noise.state_dict(), tmp)
            os.replace(tmp, path)

    def load_weights(self):
        opt = self.opt
        for label, net in (("SR", self.netSR), ("D", self.netD if opt.isTrain else None), ("E", self.netE)):
            if net is None:
                continue
            ck = torch.load(self._ckpt(label, opt.which_epoch), map_location="cpu")
            self.load_net_state(net, ck["model"] if "model" in ck else ck)
        path = self._ckpt("noise_state", opt.which_epoch)
        if opt.isTrain and os.path.exists(path) and hasattr(self.noise, "load_state_dict"):
            self.noise.load_state_dict(torch.load(path, map_location="cpu"))

    @staticmethod
    def load_net_state(net, state):
        """load_state_dict that keeps parameter storage (flat optimizer buffers stay valid)."""
        own = net.state_dict()
        missing = set(own) - set(state)
        extra = set(state) - set(own)
        if missing or extra:
            raise RuntimeError("state dict mismatch: missing %s, unexpected %s" % (sorted(missing), sorted(extra)))
        with torch.no_grad():
            for k, v in own.items():
                v.copy_(state[k].to(v.device, v.dtype))

    def load_states(self, states):
        for label, net in (("SR", self.netSR), ("D", self.netD), ("E", self.netE), ("VGG", self.vgg)):
            if net is not None and label in states:
                self.load_net_state(net, states[label])

    # ---- data plumbing
    def _native(self, data):
        """Accepts the manager's native dict (NHWC tensors + ops.Labels) or the reference's dict (one-hot NCHW
        `input_semantics`, NCHW images) and returns the native form."""
        d = {}
        sem = data.get("input_semantics")
        if isinstance(sem, torch.Tensor):
            lab = sem.argmax(1, keepdim=True).float()
            sem = ops.Labels(ops.label_to_u8(lab.cuda()), self.opt.label_nc)
        d["labels"] = sem
        gl = data.get("guiding_label")
        if isinstance(gl, torch.Tensor):
            gl = ops.Labels(ops.label_to_u8(gl.argmax(1, keepdim=True).float().cuda()), self.opt.label_nc)
        d["guiding_labels"] = gl
        for k in ("image_lr", "image_hr", "guiding_image"):
            v = data.get(k)
            # the layout is tagged by whoever produced the tensor (ops.to_nhwc / bicubic_down / the device input
            # pipeline set .dsee_layout), never guessed from the shape: [N,3,H,4] is a legal NCHW image at start_size 4
            if isinstance(v, torch.Tensor) and getattr(v, "dsee_layout", None) != "nhwc":
                assert v.dim() == 4 and v.shape[1] == 3, "%s: expected an NCHW RGB tensor, got %s" % (k, tuple(v.shape))
                v = ops.to_nhwc(v.cuda())
            d[k] = v
        return d

    # ---- losses (sr_model.py:518-564)
    def compute_generator_loss(self, d):
        opt = self.opt
        losses = OrderedDict()
        fake, _ = self.generate_fake(d)
        # D(fake | real), VGG(fake) and VGG(real) are independent given `fake`: three branches (ops.branches; the two
        # discriminator scales fork again inside netD)
        use_vgg = not opt.no_vgg_loss

        def vgg_real():
            with torch.no_grad():
                return self.vgg(d["image_hr"])
        parts = [lambda: self.discriminate(d["labels"], fake, d["image_hr"], train_d=False)]
        if use_vgg:
            parts += [lambda: self.vgg(fake), vgg_real]
        packed = getattr(self, "_vgg_packed", None)
        if packed is None:
            packed = self._vgg_packed = set()
        if use_vgg and self.plan not in packed:
            # the first forward under a plan builds the frozen VGG weights' packed / transformed images once (ops._frozen_cache
            # entries depend on the plan: precision, storage mode) and both VGG passes read them: that one time they run in
            # order, on one stream -- also when that first forward is a capture (a plan swapped in after construction)
            pred, fx, fy = [f() for f in parts]
            packed.add(self.plan)
        else:
            pred, fx, fy = (ops.branches(*parts, inputs=[fake, d["image_hr"]]) + [None, None])[:3]
        n = fake.shape[0]
        pred, pred_real = pred      # (the generated half with its graph, the real half without: discriminate(train_d=False))
        gan = 0
        for p in pred:
            gan = gan + ops.mean_loss(p[-1], None, ops.MODE_NEG, 1.0 / len(pred), valid_c=1, lo=0, hi=n)
        losses["GAN"] = gan
        if not opt.no_ganFeat_loss:
            terms = []
            for p, pr in zip(pred, pred_real):
                for f, real in zip(p[:-1], pr[:-1]):
                    terms.append(ops.mean_loss(f, real, ops.MODE_L1, opt.lambda_feat / len(pred), lo=0, hi=n))
            losses["GAN_Feat"] = _sum_terms(terms)
        if use_vgg:
            losses["VGG"] = _sum_terms([ops.mean_loss(a, b, ops.MODE_L1, w * opt.lambda_vgg)
                                        for w, a, b in zip(N.VGG_WEIGHTS, fx, fy)])
        return losses, ops.ToNCHW.apply(fake, 3)

    def compute_discriminator_loss(self, d):
        with torch.no_grad():
            fake, _ = self.generate_fake(d)
        fake = fake.detach()
        pred = self.discriminate(d["labels"], fake, d["image_hr"], train_d=True)
        n = fake.shape[0]
        losses = OrderedDict()
        df = dr = 0
        for p in pred:
            df = df + ops.mean_loss(p[-1], None, ops.MODE_HINGE_FAKE, 1.0 / len(pred), valid_c=1, lo=0, hi=n)
            dr = dr + ops.mean_loss(p[-1], None, ops.MODE_HINGE_REAL, 1.0 / len(pred), valid_c=1, lo=n, hi=2 * n)
        losses["D_Fake"], losses["D_Real"] = df, dr
        return losses

    def discriminate(self, labels, fake, real, train_d):
        """sr_model.py:655-683: one D pass over cat([fake; real]) on N.  In the generator step the D weights only
        need data gradients (their .grad is zeroed before use), so they enter detached."""
        if train_d:
            return self.netD(ops.DInput.apply(labels, fake, real), self.training)
        # generator step (round 6): the real half enters the losses detached (sr_model.py:547-564) -- it runs as a no-grad pass of
        # its own on the same weights (networks.NLayerD.forward), and the data gradients of the backward pass cover the generated
        # half only.  Returns (features of the generated images, features of the real images).
        x = ops.DInput.apply(labels, fake)
        with torch.no_grad():
            xr = ops.DInput.apply(labels, real)
        req = [p.requires_grad for p in self.netD.parameters()]
        for p in self.netD.parameters():
            p.requires_grad_(False)
        try:
            return self.netD(x, self.training, xr)
        finally:
            for p, r in zip(self.netD.parameters(), req):
                p.requires_grad_(r)

    # ---- generator path (sr_model.py:566-650)
    def generate_fake(self, d, no_noise=False):
        style = self.encode_style(d, no_noise)
        fake = self.netSR(d["image_lr"], d["labels"], style, self.noise, self.training)
        return fake, style

    def encoder_branch(self, no_noise=False, step=None):
        """('full' | 'mini', no_noise) of a forward (sr_model.py:601-650): the guided variant always encodes the full
        image; the independent variant flips two coins per training forward.  `step`: look at the coins of that forward
        index instead of the current one (DeviceNoise only)."""
        opt = self.opt
        kw = {} if step is None else {"step": step}
        if self.model_variant == "guided":
            return "full", no_noise
        full = opt.full_style_image or (self.training and self.noise.coin("enc_full", **kw) < 0.5)
        if not no_noise:
            no_noise = self.noise.coin("enc_noise", **kw) < 0.5
        return ("full" if full else "mini"), no_noise

    def encode_style(self, d, no_noise):
        opt = self.opt
        labels, img = d["labels"], d["image_lr"]
        mode, nn_ = self.encoder_branch(no_noise)
        if self.model_variant == "guided":
            if opt.guiding_style_image:
                labels, img = d["guiding_labels"], d["guiding_image"]
            else:
                img = d["image_hr"]
            return self.netE(img, labels, mode, no_noise, self.noise, self.training)
        self.last_encoded_style_is_full = mode == "full"
        if mode == "full":
            if opt.guiding_style_image:
                labels, img = d["guiding_labels"], d["guiding_image"]
            else:
                img = d["image_hr"]
        if not no_noise:
            self.last_encoded_style_is_noisy = not nn_
        return self.netE(img, labels, mode, nn_, self.noise, self.training)
