"""End-to-end parity of the HIP train step (TrainerManager on MI355X) against the CPU oracle on identical recipe
weights, inputs, noise tensors and branch decisions.  Outputs/losses are held to 1e-3 rel (north_star) — in
practice ~1e-5; gradients to the reference's own noise floor (see tests/test_oracle_golden.py)."""
import math
import random
import time

import pytest
import torch

from oracle import deepsee_oracle as O

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-20))


CASES = {
    "indep_4to32_ngf8": dict(start_size=4, crop_size=32, load_size=32, batchSize=2, ngf=8),
    "indep_8to64_ngf8": dict(start_size=8, crop_size=64, load_size=64, batchSize=2, ngf=8),
    "guided_4to32_ngf8": dict(start_size=4, crop_size=32, load_size=32, batchSize=2, ngf=8, netE="fullstyle",
                              noisy_style_scale=0.05, guiding_style_image=True),
    "guided_normal_4to32_ngf8": dict(start_size=4, crop_size=32, load_size=32, batchSize=2, ngf=8, netE="fullstyle",
                                     noisy_style_scale=0.05, noisy_style_dist="normal", guiding_style_image=True),
    "config1_4to32_full": dict(start_size=4, crop_size=32, load_size=32, batchSize=2),
    "puresean_4to128_ngf4": dict(start_size=4, crop_size=128, load_size=512, batchSize=2, ngf=4, add_noise=False,
                                 max_fm_size=64),
    "clip_nottur_4to32": dict(start_size=4, crop_size=32, load_size=32, batchSize=2, ngf=8, add_noise=False,
                              no_TTUR=True, gradient_clip=0.01),
    # (the benchmark's geometry with a batch > 1 -- BatchNorm over several images, per-image style-table groups, the Winograd
    # chunking of a batch -- is held against the oracle at the benchmark's own bs = 8 x 512 channels by
    # test_benchmark_path_matches_oracle, whose first occurrence of every graph is this eager path; the 128-channel bs = 4 case of
    # round 4 left the suite with round 5: 45 - 60 s of CPU oracle time, profiles/r05_gpu_tests.log)
}


def load_oracle_state(tm, orc):
    """Copy the oracle's current parameters and buffers into the HIP model (same keys: checkpoint layout)."""
    tm.sr_model.load_states({net: {k: v.detach().clone() for k, v in orc.S[net].items()} for net in ("SR", "D", "E")})


def _grad_or_zero(p):
    """FlatAdam.zero_grad() sets .grad to None (torch's set_to_none): a parameter the backward pass did not reach has no
    gradient tensor -- the reference's zero-filled .grad."""
    return (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().clone()


def run_case(over, seed, iters=1, sync_before_d=True):
    """One or more G+D iterations on both sides.  `sync_before_d`: after the G step the oracle's post-step state is
    loaded into the HIP model, so the D step starts from IDENTICAL weights/buffers on both sides and its gradients can
    be held tight (without it they inherit the +-lr sign-noise of the beta1 = 0 Adam step in between: 2e-2 .. 2e-1)."""
    from deepsee_amd import networks as N
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    n = over["batchSize"]
    oopt = O.make_opt(**over)
    states = O.recipe_state(oopt, gain=1.0)
    batch = O.synthetic_batch(oopt, n, seed=seed)
    ctl = O.RecordingCtl()
    orc = O.Oracle(oopt, states, ctl)
    orc.create_optimizers()
    tm = TrainerManager(make_opt(**over))
    tm.sr_model.load_states(states)
    out = []
    random.seed(seed)
    torch.manual_seed(seed)
    for it in range(iters):
        start = len(ctl.tape)
        gl, fake = orc.run_generator_one_step({k: v.clone() for k, v in batch.items()})
        ggrads = {"%s.%s" % (net, k): p.grad.clone() for net in ("SR", "E") for k, p in orc.params(net)
                  if p.grad is not None}

        class _Snap:   # the oracle's state right after its G step (what load_oracle_state copies)
            S = {net: {k: v.detach().clone() for k, v in orc.S[net].items()} for net in ("SR", "D", "E")}
        orc_state_after_g = _Snap
        dl = orc.run_discriminator_one_step({k: v.clone() for k, v in batch.items()})
        dgrads = {"D." + k: p.grad.clone() for k, p in orc.params("D") if p.grad is not None}
        tm.sr_model.noise = N.ReplayNoise(ctl.tape[start:])
        tm.run_generator_one_step({k: v.clone() for k, v in batch.items()})
        torch.cuda.synchronize()
        hg = {nm: _grad_or_zero(p) for nm, p in zip(tm.optimizer_G.names, tm.optimizer_G.params)}
        touched_g = {nm for nm, t in zip(tm.optimizer_G.names, tm.optimizer_G.touched) if t}
        hgl = {k: float(v) for k, v in tm.g_losses.items()}
        hfake = tm.get_latest_generated().detach().cpu()
        post_g = {net: {k: v.detach().cpu().clone() for k, v in getattr(tm.sr_model, "net" + net).state_dict().items()}
                  for net in ("SR", "E")}
        if sync_before_d:
            load_oracle_state(tm, orc_state_after_g)
        tm.run_discriminator_one_step({k: v.clone() for k, v in batch.items()})
        torch.cuda.synchronize()
        hd = {nm: _grad_or_zero(p) for nm, p in zip(tm.optimizer_D.names, tm.optimizer_D.params)}
        hdl = {k: float(v) for k, v in tm.d_losses.items()}
        assert tm.sr_model.noise.pos == len(tm.sr_model.noise.tape)
        out.append(dict(gl={k: float(v.detach()) for k, v in gl.items()}, fake=fake.detach(), ggrads=ggrads,
                        dl={k: float(v.detach()) for k, v in dl.items()}, dgrads=dgrads, hgl=hgl, hfake=hfake, hg=hg,
                        hd=hd, hdl=hdl, touched_g=touched_g, post_g=post_g, orc_after_g=orc_state_after_g.S))
    return orc, tm, out


# Model-level gradients under the REAL losses (hinge / L1 terms: sign-function gradients; ReLU / LeakyReLU kinks): ~3x the
# worst values observed over all cases on the MI355X (median 2.6e-3, max 8.0e-3: profiles/r03_gpu_tests.log) -- loose enough for the
# kink-flip floor described in DESIGN 4, tight enough to catch a 3x regression.  The kink-free 1e-3 claim is carried by
# test_full_size_smooth_loss_backward and the benchmark-shape layer tests.
GRAD_MEDIAN_BOUND, GRAD_MAX_BOUND = 6e-3, 2.5e-2
# 16-bit storage mode (one-term fp16 operands), model-level gradients against the fp32 oracle
# (observed on the MI355X: median 9.8e-2 / max 2.7e-1 at bs = 1 with a forward deviation d = 6e-3 -- the LeakyReLU / ReLU
# branch of a fraction ~d of the activations flips and every gradient downstream moves by ~sqrt(d) = 8e-2, the floor DESIGN 4
# measures for the fp32 path at d = 5e-6; any 16-bit operand format sits on that floor, bf16 operands 3x higher)
# Round 6: the kink-free check of the 16-bit kernels is test_benchmark_shape_{conv,norm}_vs_float64's 16-bit leg (every gradient
# <= 1e-2 / 8e-3 against float64); these model-level bounds sit 1.5x above the worst observed value (bs = 1: 9.8e-2 / 2.7e-1;
# bs = 8 on the replayed graph: 4.2e-2 / 1.8e-1)
HALF_GRAD_MEDIAN_BOUND, HALF_GRAD_MAX_BOUND = 1.5e-1, 4.2e-1


@pytest.mark.parametrize("name", list(CASES))
def test_train_step_matches_oracle(name):
    """Full G+D step with the real losses.  Forward quantities are tight.  Gradients of the REAL loss are only
    comparable up to the path's conditioning: the L1 (feature-matching, VGG) and hinge terms have sign-function
    gradients, and at these tiny test resolutions a single near-tie element (|a-b| ~ 1e-6, found and printed by
    tools/debug_vgg2.py) flips between any two fp32 implementations and moves a 4096-element tap's gradient by
    2/sqrt(4096) = 3e-2.  The tight gradient check is test_smooth_loss_backward below; here the median over all
    tensors must be small and no tensor may be far off."""
    orc, tm, out = run_case(CASES[name], seed=101 + len(name))
    r = out[0]
    clip = CASES[name].get("gradient_clip", -1)
    if clip > 0:  # clip_grad_value_ rewrites .grad in the reference; the HIP path clamps inside the Adam kernel
        r["hg"] = {k: v.clamp(-clip, clip) for k, v in r["hg"].items()}
        r["hd"] = {k: v.clamp(-clip, clip) for k, v in r["hd"].items()}
    for k, v in r["gl"].items():
        assert abs(r["hgl"][k] - v) <= 1e-4 * abs(v), (k, r["hgl"][k], v)
    assert rel(r["hfake"], r["fake"]) < 1e-4
    # the set of tensors that received a gradient equals the reference's "grad is not None" set
    assert r["touched_g"] == set(r["ggrads"]), r["touched_g"] ^ set(r["ggrads"])
    gmax = max(float(v.norm()) for v in r["ggrads"].values())
    errs = sorted(float((r["hg"][k].double() - v.double()).norm()) / max(float(v.norm()), 1e-3 * gmax)
                  for k, v in r["ggrads"].items())
    assert errs[len(errs) // 2] < GRAD_MEDIAN_BOUND and errs[-1] < GRAD_MAX_BOUND, (errs[len(errs) // 2], errs[-1])
    # the D step started from the oracle's post-G-step state on both sides: losses and gradients are tight
    for k, v in r["dl"].items():
        assert abs(r["hdl"][k] - v) <= 1e-4 * abs(v), (k, r["hdl"][k], v)
    dmax = max(float(v.norm()) for v in r["dgrads"].values())
    derrs = sorted(float((r["hd"][k].double() - v.double()).norm()) / max(float(v.norm()), 1e-2 * dmax)
                   for k, v in r["dgrads"].items())
    assert derrs[-1] < 5e-3, derrs[-1]
    # post-step state: parameters (Adam) and buffers (BN running stats twice, SN u/v twice).  G/E parameters are
    # compared as they were right after the HIP G step (before the state sync), D and all buffers at the end.
    sd = {"SR": tm.sr_model.netSR.state_dict(), "D": tm.sr_model.netD.state_dict(), "E": tm.sr_model.netE.state_dict()}
    zero_grad = {k for k, v in {**r["ggrads"], **r["dgrads"]}.items() if float(v.norm()) < 1e-4 * max(gmax, dmax)}
    for net in ("SR", "D", "E"):
        assert set(sd[net].keys()) == set(orc.S[net].keys())
        for k, v in sd[net].items():
            if not v.is_floating_point():
                continue
            ref = orc.S[net][k].detach()
            if net != "D" and not O.is_buffer(k):
                v, ref = r["post_g"][net][k], r["orc_after_g"][net][k]
            if O.is_buffer(k):
                assert rel(v.cpu(), ref) < 2e-3, (net, k)
            else:
                # beta1=0 Adam moves every element by ~lr*sign(g): elements whose gradient is rounding noise may
                # step the other way, so the bound is per element (2 steps of the largest lr) plus a small mean.
                dlt = (v.cpu() - ref).abs()
                assert float(dlt.max()) <= 2.5 * 4e-4, (net, k, float(dlt.max()))
                if "%s.%s" % (net, k) not in zero_grad:   # (analytically-zero gradients are all noise)
                    assert float(dlt.mean()) <= 4e-5, (net, k, float(dlt.mean()))
    print("G-grad rel err: median %.2e max %.2e; D-grad max %.2e" % (errs[len(errs) // 2], errs[-1], derrs[-1]))


@pytest.mark.parametrize("name,over", [
    pytest.param("config1_32to256_bs1", dict(batchSize=1), marks=pytest.mark.slow),
    # BASELINE configs[4] (independent 32x 16 -> 512: PureSEAN tail, the capped path's 2 x 2 block-sum gradient at 512^2) forward AND
    # backward under the real losses, 256 channels: ONE fp32 oracle step (~35 s of CPU) in the default run; its float64 smooth-loss
    # form (three oracle passes, ~95 s) is test_full_size_smooth_loss_backward[indep_16to512_bs1_ngf16] under --runslow
    ("indep_16to512_bs1_ngf16", dict(batchSize=1, ngf=16, start_size=16, crop_size=512, load_size=512, add_noise=False)),
])
def test_full_size_step_matches_oracle(name, over):
    """A BASELINE configuration at its full size with bs = 1, EAGER with an explicit noise tape, against the CPU oracle (one G step
    + one D step).  configs[1]: the default run holds the same comparison at bs = 8 on the replayed graphs
    (test_benchmark_path_matches_oracle); this is the nightly check of the eager path beyond its first occurrence before capture
    (ADVICE r5).  configs[4]: see the parameter list."""
    orc, tm, out = run_case(over, seed=4242)
    r = out[0]
    for k, v in r["gl"].items():
        assert abs(r["hgl"][k] - v) <= 1e-4 * abs(v), (k, r["hgl"][k], v)
    dev = rel(r["hfake"], r["fake"])
    assert dev < 1e-4, dev
    assert r["touched_g"] == set(r["ggrads"])
    gmax = max(float(v.norm()) for v in r["ggrads"].values())
    errs = sorted(float((r["hg"][k].double() - v.double()).norm()) / max(float(v.norm()), 1e-3 * gmax)
                  for k, v in r["ggrads"].items())
    assert errs[len(errs) // 2] < GRAD_MEDIAN_BOUND and errs[-1] < GRAD_MAX_BOUND, (errs[len(errs) // 2], errs[-1])
    for k, v in r["dl"].items():
        assert abs(r["hdl"][k] - v) <= 2e-3 * abs(v), (k, r["hdl"][k], v)
    # (the D step started from the oracle's post-G-step state on both sides, run_case: its gradients are tight)
    dmax = max(float(v.norm()) for v in r["dgrads"].values())
    derrs = sorted(float((r["hd"][k].double() - v.double()).norm()) / max(float(v.norm()), 1e-2 * dmax)
                   for k, v in r["dgrads"].items())
    assert derrs[-1] < 5e-3, derrs[-1]
    print("%s at full size: |fake - oracle| / |oracle| = %.2e, G-grad rel err median %.2e max %.2e, D-grad max %.2e, losses %s"
          % (name, dev, errs[len(errs) // 2], errs[-1], derrs[-1], {k: round(v, 5) for k, v in r["hgl"].items()}))



class _TapedNoise:
    """Mixin over networks.DeviceNoise: logs every draw request of a forward -- (kind, tag, shape, Philox seed, stream offset)
    -- so that the numbers a captured / replayed step drew in registers can be regenerated afterwards with dsee_rng_fill at the
    same (seed, offset, device epoch) and handed to the oracle as its noise tape (normalization.py:299-304 takes `noise=`)."""

    def begin_step(self):
        super().begin_step()
        self.log = []

    def coin(self, tag, step=None):
        v = super().coin(tag, step)
        if step is None:
            self.log.append(("coin", tag, v))
        return v

    def normal_nhwc(self, shape, tag):
        self.log.append(("normal_nhwc", tag, tuple(shape), self.seed, self.offset))
        return super().normal_nhwc(shape, tag)

    def uniform(self, shape, tag):
        self.log.append(("uniform", tag, tuple(shape), self.seed, self.offset))
        return super().uniform(shape, tag)

    def normal(self, shape, tag):
        self.log.append(("normal", tag, tuple(shape), self.seed, self.offset))
        return super().normal(shape, tag)

    def dump(self, log):
        """The oracle tape of one forward: every logged stream position regenerated under the CURRENT device epoch (call it
        right after the step that drew them: the next training forward advances the epoch)."""
        from deepsee_amd import ops
        self.ensure_registered()
        tape = []
        for e in log:
            if e[0] == "coin":
                tape.append(e)
            elif e[0] == "normal_nhwc":
                t = ops.rng_fill(e[2], e[3], e[4], True)
                tape.append(("normal", e[1], ops.to_nchw(t, e[2][3]).cpu()))
            else:
                tape.append((e[0], e[1], ops.rng_fill(e[2], e[3], e[4], e[0] == "normal").cpu()))
        return tape


@pytest.mark.parametrize("preset", ["independent_8x_256", "guided_8x_256"])
def test_benchmark_path_matches_oracle(preset):
    """The path bench.py TIMES, under the oracle: BASELINE configs[1] (and configs[3]'s per-rank workload,
    guided, bs = 8) with the manager's defaults -- hipGraph replay, DeviceNoise (Philox regenerated in registers by the fused
    noise / shortcut kernels), bs = 8 x 512 channels.  The model is stepped until the next G and D half steps are REPLAYS of
    captured graphs; its state is snapshotted; the replayed G step runs; every Philox draw of that step is regenerated with
    dsee_rng_fill at the same (seed, offset, epoch) into a tape; the oracle runs the same step from the snapshot on that tape.
    Losses <= 1e-4, generated image <= 1e-4, G gradients (the flat buffer the in-graph Adam consumed) within the model-level
    bounds; then the D replay from the oracle's post-G state: losses <= 2e-3, D gradients <= 5e-3 -- for both presets (round 6:
    configs[3]'s D half is compared as well).  The oracle needs ~60 GB of
    host memory at bs = 8; on a smaller host the test drops to bs = 4 and says so."""
    import os
    import warnings
    from deepsee_amd import networks as N
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt, PRESETS
    try:
        ram_gb = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 1e9
    except (ValueError, OSError):
        ram_gb = 0.0
    # the headline configuration runs at the benchmark's own bs = 8; the guided preset at bs = 4 (same kernels and tile shapes from
    # N = 4 on; the bs = 8 oracle pass is a minute of CPU time and the suite has to fit a slow driver box)
    bs = 8 if (ram_gb >= 90 and preset == "independent_8x_256") else 4
    over = dict(PRESETS[preset], batchSize=bs)
    t_phase, phases = [time.time()], []

    def lap(name):
        t_phase.append(time.time())
        phases.append("%s %.0f s" % (name, t_phase[-1] - t_phase[-2]))
    oopt = O.make_opt(**over)
    states = O.recipe_state(oopt, gain=1.0)
    batch = O.synthetic_batch(oopt, bs, seed=2468)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        tm = TrainerManager(make_opt(preset, batchSize=bs, seed=0))
    assert tm.use_graphs                                   # the default
    m = tm.sr_model
    m.load_states(states)

    class Taped(_TapedNoise, N.DeviceNoise):
        pass
    m.noise = Taped(0)
    feed = lambda: {k: v.clone() for k, v in batch.items()}
    sig = tm._shape_signature(batch)

    def key(which):
        return (which,) + tuple(m.encoder_branch(False, step=m.noise.step + 1)) + (sig, m.plan)

    logs = {}

    def half(which):
        k, before = key(which), tm.graph_stats["replayed"]
        (tm.run_generator_one_step if which == "G" else tm.run_discriminator_one_step)(feed())
        replayed = tm.graph_stats["replayed"] > before
        if not replayed:
            logs[k] = list(m.noise.log)     # (eager or capture pass: the Python side ran; a replay draws at the same positions)
        return k, replayed

    # ---- until the NEXT G and the D behind it are both replays
    for it in range(48):
        kg = key("G")
        m.noise.step += 1
        kd = key("D")
        m.noise.step -= 1
        if kg in tm._graphs and kd in tm._graphs:
            break
        half("G")
        half("D")
    else:
        pytest.fail("no replayable G + D pair after 48 iterations: %r" % (tm.graph_stats,))
    torch.cuda.synchronize()
    warm_iters = it
    lap("setup + %d warm iterations (eager, captures)" % it)
    snap = {net: {k: v.detach().cpu().clone() for k, v in getattr(m, "net" + net).state_dict().items()} for net in ("SR", "D", "E")}
    snap["VGG"] = states["VGG"]
    noise_before_g = m.noise.state_dict()          # (forward index, device epoch) the replayed G step starts from
    # ---- the replayed G step and its tape
    kg2, replayed = half("G")
    assert replayed and kg2 == kg
    torch.cuda.synchronize()
    tape = m.noise.dump(logs[kg])
    hgl = {k: float(v) for k, v in tm.g_losses.items()}
    hfake = tm.get_latest_generated().detach().cpu()
    hg = {nm: tm.optimizer_G.grad_view(nm).detach().cpu().clone() for nm in tm.optimizer_G.names}
    active = tm.optimizer_G._active_dev.cpu().tolist()
    touched = {nm for nm, a in zip(tm.optimizer_G.names, active) if a}
    # ---- the oracle on the same state and tape
    ctl = O.ReplayCtl(tape)
    orc = O.Oracle(oopt, snap, ctl)
    orc.create_optimizers()
    lap("replayed G + tape")
    gl, fake = orc.run_generator_one_step(feed())
    lap("oracle G step")
    assert ctl.pos == len(ctl.tape)
    ggrads = {"%s.%s" % (net, k): p.grad.clone() for net in ("SR", "E") for k, p in orc.params(net) if p.grad is not None}
    for k, v in gl.items():
        assert abs(hgl[k] - float(v.detach())) <= 1e-4 * abs(float(v.detach())), (k, hgl[k], float(v.detach()))
    dev = rel(hfake, fake.detach())
    assert dev < 1e-4, dev
    assert touched == set(ggrads), touched ^ set(ggrads)
    gmax = max(float(v.norm()) for v in ggrads.values())
    errs = sorted(float((hg[k].double() - v.double()).norm()) / max(float(v.norm()), 1e-3 * gmax) for k, v in ggrads.items())
    assert errs[len(errs) // 2] < GRAD_MEDIAN_BOUND and errs[-1] < GRAD_MAX_BOUND, (errs[len(errs) // 2], errs[-1])
    report = ("%s at bs = %d, REPLAYED graphs + Philox tape (%d draws, host RAM %.0f GB): |fake - oracle| / |oracle| = %.2e, "
              "G-grad rel err median %.2e max %.2e, losses %s" % (preset, bs, len(tape), ram_gb, dev, errs[len(errs) // 2],
                                                                  errs[-1], {k: round(v, 5) for k, v in hgl.items()}))
    if preset == "independent_8x_256":
        # ---- BASELINE configs[2]'s per-rank workload on the SAME oracle pass, and on the path `bench.py --dtype fp16` times (round
        # 6, VERDICT r5 #4b): a second manager in the 16-bit storage mode with the manager's defaults -- hipGraph replay, DeviceNoise
        # in registers -- is stepped through the same number of iterations (same seed: the same branch coins, the same capture
        # schedule, the same graph keys), given the fp32 model's snapshot and Philox position (forward index, device epoch), and
        # REPLAYS its captured G graph: it draws exactly the numbers of the tape the oracle ran on.  Generated image within SURVEY
        # 8(d)'s 3e-2, losses within 5 %, gradients (the flat buffer the in-graph Adam consumed) within HALF_GRAD_*.
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            th = TrainerManager(make_opt(preset, batchSize=bs, seed=0, precision="fp16"))
        assert th.use_graphs and th.sr_model.plan.half and th.sr_model.plan.half_norms
        mh = th.sr_model
        mh.load_states(states)
        mh.noise = N.DeviceNoise(0)                 # the same Philox seed and coin sequence as the fp32 model's Taped(0)
        for _ in range(warm_iters):
            th.run_generator_one_step(feed())
            th.run_discriminator_one_step(feed())
        torch.cuda.synchronize()
        assert mh.noise.state_dict() == noise_before_g, (mh.noise.state_dict(), noise_before_g)
        kgh = ("G",) + tuple(mh.encoder_branch(False, step=mh.noise.step + 1)) + (th._shape_signature(batch), mh.plan)
        assert kgh in th._graphs and kgh[:-1] == kg[:-1]
        mh.load_states(snap)
        before = th.graph_stats["replayed"]
        th.run_generator_one_step(feed())
        torch.cuda.synchronize()
        assert th.graph_stats["replayed"] == before + 1          # the captured graph ran, nothing was enqueued from Python
        hdev = rel(th.get_latest_generated().detach().cpu(), fake.detach())
        assert hdev < 3e-2, hdev
        for k, v in gl.items():
            w = float(th.g_losses[k])
            assert abs(w - float(v.detach())) <= 0.05 * abs(float(v.detach())) + 1e-3, (k, w, float(v.detach()))
        hh = {nm: th.optimizer_G.grad_view(nm).detach().cpu().clone() for nm in th.optimizer_G.names}
        herrs = sorted(float((hh[k].double() - v.double()).norm()) / max(float(v.norm()), 1e-3 * gmax) for k, v in ggrads.items())
        report += (" | 16-bit mode, REPLAYED graph + the same Philox draws, from the same state: fake %.2e, G-grad rel err median "
                   "%.2e, 90 %% %.2e, max %.2e" % (hdev, herrs[len(herrs) // 2], herrs[int(0.9 * (len(herrs) - 1))], herrs[-1]))
        assert herrs[len(herrs) // 2] < HALF_GRAD_MEDIAN_BOUND and herrs[-1] < HALF_GRAD_MAX_BOUND, (herrs[len(herrs) // 2], herrs[-1])
        th.close()
        del th, mh
        lap("16-bit manager: warm iterations + replay")
    # ---- the replayed D step from the oracle's post-G state (both sides identical weights / buffers again)
    load_oracle_state(tm, orc)
    kd2, replayed = half("D")
    assert replayed and kd2 == kd
    torch.cuda.synchronize()
    ctl.tape += m.noise.dump(logs[kd])
    hdl = {k: float(v) for k, v in tm.d_losses.items()}
    hd = {nm: tm.optimizer_D.grad_view(nm).detach().cpu().clone() for nm in tm.optimizer_D.names}
    lap("replayed D")
    dl = orc.run_discriminator_one_step(feed())
    lap("oracle D step")
    assert ctl.pos == len(ctl.tape)
    dgrads = {"D." + k: p.grad.clone() for k, p in orc.params("D") if p.grad is not None}
    for k, v in dl.items():
        assert abs(hdl[k] - float(v.detach())) <= 2e-3 * abs(float(v.detach())), (k, hdl[k], float(v.detach()))
    dmax = max(float(v.norm()) for v in dgrads.values())
    derrs = sorted(float((hd[k].double() - v.double()).norm()) / max(float(v.norm()), 1e-2 * dmax) for k, v in dgrads.items())
    assert derrs[-1] < 5e-3, derrs[-1]
    print(report + " | D step replayed from the oracle's post-G state: D-grad max %.2e, losses %s | wall time: %s"
          % (derrs[-1], {k: round(v, 5) for k, v in hdl.items()}, ", ".join(phases)))
    tm.close()


def smooth_loss_errors(over, seed=555, plain_f32=True):
    """Backward parity with the sign-function losses taken out: L_G = <fake, R>, L_D = sum_k <D_k(cat[fake;real]), R_k>,
    L_V = sum_i <VGG_i(fake), R_i> with fixed random R.  Every HIP backward kernel of the path (SPADE/SEAN modulate,
    BN, convs dgrad/wgrad, SN, noise, upsample, IN, pools, style pool/gather, one-hot conv) is exercised.  Returns, per
    parameter tensor, the relative error against the oracle run in FLOAT64 of: HIP, the fp32 oracle, and the fp32 oracle
    with its input image scaled by (1 + pert), pert = HIP's own forward deviation (the conditioning yardstick)."""
    from deepsee_amd import networks as N, ops
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    n = over["batchSize"]
    oopt = O.make_opt(**over)
    states = O.recipe_state(oopt, gain=1.0)
    batch = O.synthetic_batch(oopt, n, seed=seed)

    def oracle_run(dtype, tape=None, pert=0.0):
        t0 = time.time()
        try:
            return _oracle_run(dtype, tape, pert)
        finally:
            print("  [oracle pass %s%s: %.1f s of CPU]" % (str(dtype).replace("torch.", ""), " perturbed" if pert else "", time.time() - t0))

    def _oracle_run(dtype, tape=None, pert=0.0):
        ctl = O.RecordingCtl() if tape is None else O.ReplayCtl(tape)
        orc = O.Oracle(oopt, states, ctl, dtype=dtype)
        random.seed(3)
        torch.manual_seed(3)
        b = {k: v.clone() for k, v in batch.items()}
        b["image"] = b["image"] * (1 + pert)
        data = orc.preprocess(b)
        fake, _ = orc.generate_fake(data)
        Rs = {"fake": torch.randn(fake.shape, generator=torch.Generator().manual_seed(1)).to(dtype)}
        loss = (fake * Rs["fake"]).sum()
        both = torch.cat([torch.cat([data["input_semantics"], fake], 1),
                          torch.cat([data["input_semantics"], data["image_hr"]], 1)], 0)
        douts = orc.d_forward(both)
        k = 0
        for o in douts:
            for t in o:
                R = torch.randn(t.shape, generator=torch.Generator().manual_seed(100 + k)).to(dtype)
                Rs["d%d" % k] = R
                loss = loss + (t * R).sum() / t[0].numel() ** 0.5
                k += 1
        for i, t in enumerate(orc.vgg_features(fake)):
            R = torch.randn(t.shape, generator=torch.Generator().manual_seed(200 + i)).to(dtype)
            Rs["v%d" % i] = R
            loss = loss + (t * R).sum() / t[0].numel() ** 0.5
        loss.backward()
        grads = {"%s.%s" % (net, kk): p.grad.clone() for net in ("SR", "E", "D") for kk, p in orc.params(net)
                 if p.grad is not None}
        return orc, ctl, fake.detach(), float(loss.detach()), grads, Rs

    if plain_f32:
        orc32, ctl, fake32, loss32, g32, Rs = oracle_run(torch.float32)
        _, _, fake64, loss64, g64, _ = oracle_run(torch.float64, ctl.tape)
    else:   # full-size cases: the unperturbed fp32 oracle is informational only -- one CPU pass less (the tape is the same)
        _, ctl, fake64, loss64, g64, Rs = oracle_run(torch.float64)
        Rs = {k: v.float() for k, v in Rs.items()}
        g32 = None
    tm = TrainerManager(make_opt(**over))
    m = tm.sr_model
    m.load_states(states)
    m.noise = N.ReplayNoise(ctl.tape)
    tm.optimizer_G.zero_grad()
    tm.optimizer_D.zero_grad()
    with m.plan.active():      # (model internals called directly: SRModel.forward is what normally activates the plan)
        d = m._native(tm.preprocess_input({k: v.clone() for k, v in batch.items()}))
        fake, _ = m.generate_fake(d)
        loss = (ops.ToNCHW.apply(fake, 3) * Rs["fake"].cuda()).sum()
        douts = m.discriminate(d["labels"], fake, d["image_hr"], train_d=True)
        k = 0
        for o in douts:
            for t in o:
                c = Rs["d%d" % k].shape[1]
                loss = loss + (ops.ToNCHW.apply(t, c) * Rs["d%d" % k].cuda()).sum() / Rs["d%d" % k][0].numel() ** 0.5
                k += 1
        for i, t in enumerate(m.vgg(fake)):
            R = Rs["v%d" % i]
            loss = loss + (ops.ToNCHW.apply(t, R.shape[1]) * R.cuda()).sum() / R[0].numel() ** 0.5
    loss.backward()
    torch.cuda.synchronize()
    hg = {nm: _grad_or_zero(p) for o in (tm.optimizer_G, tm.optimizer_D) for nm, p in zip(o.names, o.params)}
    assert abs(float(loss) - loss64) <= 1e-4 * abs(loss64)
    dev = rel(ops.to_nchw(fake.detach(), 3).cpu(), fake64)
    assert dev < 1e-5
    # yardstick for the path's conditioning: the SAME fp32 oracle with the input image scaled by (1 + pert), pert = the
    # forward deviation HIP actually has (|fake - f64|: ~1.4e-6 with direct convolutions, up to ~1e-5 with the
    # Winograd F(4x4,3x3) layers, whose transforms carry ~10x the fp32 rounding error; oracle-f32 itself: 8e-7)
    pert = max(1e-6, dev)
    _, _, _, _, g32p, _ = oracle_run(torch.float32, ctl.tape, pert=pert)
    gmax = max(float(v.norm()) for v in g64.values())
    rows = []
    for kk, v in g64.items():
        # alpha_gamma / alpha_beta (SEAN blend scalars) are differences of two ~1e6-term inner products in both
        # implementations (cancellation): judge them against 1 % of the largest gradient instead of their own size
        den = max(float(v.norm()), (1e-2 if v.numel() == 1 else 1e-3) * gmax)
        rows.append((kk, float((hg[kk].double() - v).norm()) / den,
                     float((g32[kk].double() - v).norm()) / den if g32 is not None else float("nan"),
                     float((g32p[kk].double() - v).norm()) / den))
    return rows, dev, pert


def _summ(rows):
    med = lambda t: sorted(t)[len(t) // 2]
    q90 = lambda t: sorted(t)[int(0.9 * (len(t) - 1))]
    ehs, ecs, eps_ = [r[1] for r in rows], [r[2] for r in rows], [r[3] for r in rows]
    return med, q90, ehs, ecs, eps_


@pytest.mark.parametrize("name", ["indep_8to64_ngf8", "guided_4to32_ngf8", "puresean_4to128_ngf4", "config1_4to32_full"])
def test_smooth_loss_backward(name):
    """Small configurations: HIP must deviate from exact (float64) arithmetic no more than 3x what the fp32 oracle does
    under a forward deviation of HIP's size (or 20x the unperturbed oracle) at the median and at the 90th percentile
    over parameter tensors; the single worst tensor is decided by individual LeakyReLU/ReLU kink crossings at these
    tiny resolutions (the perturbed oracle's own maximum moves 5e-4 .. 3e-3 between pert = 1.2e-6 and 1.4e-6), so it
    only gets a cap of 3e-2 (real kernel bugs are O(1))."""
    rows, dev, pert = smooth_loss_errors(CASES[name])
    med, q90, ehs, ecs, eps_ = _summ(rows)
    worst = max(rows, key=lambda r: r[1])
    print("fake deviation %.1e | HIP-vs-f64 grad error: median %.2e worst %.2e (%s) | oracle-f32: median %.2e max %.2e | "
          "oracle-f32 with %.1e input perturbation: median %.2e max %.2e"
          % (dev, med(ehs), worst[1], worst[0], med(ecs), max(ecs), pert, med(eps_), max(eps_)))
    assert worst[1] < 3e-2, (worst, max(eps_), max(ecs))
    assert q90(ehs) <= max(3 * q90(eps_), 20 * q90(ecs), 1e-3), (q90(ehs), q90(eps_), q90(ecs))
    assert med(ehs) <= max(3 * med(eps_), 20 * med(ecs), 3e-4), (med(ehs), med(eps_), med(ecs))


# tensors that sit ON the conditioning boundary of test_full_size_smooth_loss_backward (perturbed fp32 oracle 2.8e-4 of float64
# against the 3e-4 class boundary; HIP 1.1e-3 against the well-conditioned class's 1e-3): named, not silently re-classed
BORDERLINE = {"indep_16to512_bs1_ngf16": ("D.discriminator_0.model1.0.0.weight_orig",),
              "indep_16to512_bs1": ("D.discriminator_0.model1.0.0.weight_orig",)}


@pytest.mark.parametrize("name,over", [
    ("config1_32to256_bs1", dict(batchSize=1)),                       # BASELINE configs[1] geometry, 512 channels
    # (round 5: the bs = 8 x 128-channel case of round 4 is gone -- the benchmark's own bs = 8 x 512 channels is now held against
    # the oracle on the replayed graphs, test_benchmark_path_matches_oracle, gradients included, and the float64 CPU passes here
    # are what the suite's wall time is made of)
    # (BASELINE configs[3], guided 8x 32 -> 256: at its per-rank bs = 8, gradients included, under
    # test_benchmark_path_matches_oracle[guided_8x_256]; the float64 smooth-loss form of the guided variant at 4 -> 32:
    # test_smooth_loss_backward[guided_4to32_ngf8].  The bs = 1 full-size float64 pass of round 4 -- 62 s of CPU time -- left
    # the suite with round 5: 849 -> ~790 s, profiles/r05_gpu_tests.log)
    # BASELINE configs[4]: independent 32x 16 -> 512 -- PureSEAN tail, the capped path's 2x2 block-sum gradient at 512^2
    # (256 channels: what is specific to configs[4] is its resolution -- the float64 oracle pass at 512 channels is 3 minutes of
    # CPU time; forward + losses at 512 channels: test_full_size_forward_and_losses)
    pytest.param("indep_16to512_bs1_ngf16", dict(batchSize=1, ngf=16, start_size=16, crop_size=512, load_size=512, add_noise=False),
                 marks=pytest.mark.slow),    # (round 6: default run -> test_full_size_step_matches_oracle[indep_16to512_bs1_ngf16])
    # ---- the nightly form (--runslow / DSEE_RUN_SLOW=1): the cases rounds 4-5 took out of the default run for their
    # float64 CPU time, not for their content
    pytest.param("guided_32to256_bs1", dict(batchSize=1, netE="fullstyle", noisy_style_scale=0.05, guiding_style_image=True),
                 marks=pytest.mark.slow),                                # BASELINE configs[3] at full size, float64 (62 s)
    pytest.param("indep_16to512_bs1", dict(batchSize=1, start_size=16, crop_size=512, load_size=512, add_noise=False),
                 marks=pytest.mark.slow),                                # BASELINE configs[4] backward at 512 channels (3 min)
    pytest.param("indep_32to256_bs8_ngf8", dict(batchSize=8, ngf=8), marks=pytest.mark.slow),   # bs = 8 at 128 channels
])
def test_full_size_smooth_loss_backward(name, over):
    """Gradients of the whole path at the BENCHMARK shapes (512 channels, 32 -> 256: the 256x256 / 256x160 bf16x3 GEMM
    tiles, per-image tables, Winograd transforms at 128^2 / 256^2; and bs = 8 at 128 channels) against the oracle run
    in float64.  north_star asks 1e-3; measured here, the fp32 ORACLE ITSELF is not that close to exact arithmetic at
    this depth: its unperturbed median error is ~7e-4 and, with its input image scaled by (1 + 6e-6) -- the size of
    HIP's forward deviation -- ~2e-3 on almost every tensor (a forward deviation d flips the ReLU / LeakyReLU branch of a
    fraction ~d of the activations in G, D and VGG; every gradient downstream moves by ~sqrt(d)).  So:
      * a tensor is WELL CONDITIONED when the perturbed fp32 oracle stays within 3e-4 of float64: HIP must be within 1e-3.
        (Round 5 had moved the boundary to 2e-4 for one tensor; ADVICE r5: the boundary is 3e-4 again and that tensor is named
        in BORDERLINE below -- D.discriminator_0.model1 at 16 -> 512, which the perturbed oracle itself moves by 2.8e-4: it is
        held to the other class's rule, <= 5x its own perturbed-oracle error, and the test prints that it was.)
      * every other tensor is held to the oracle's own behaviour under the equal-size perturbation: <= 5x its error
        (floor 3e-3), and HIP's median over all tensors <= 1.5x the perturbed oracle's median.
    The kink-free check of the same kernels at the same shapes to 1e-3 is tests/test_gpu_ops.py::
    test_benchmark_shape_conv_vs_float64 / test_benchmark_shape_norm_vs_float64."""
    rows, dev, pert = smooth_loss_errors(over, seed=777, plain_f32=False)
    med, q90, ehs, ecs, eps_ = _summ(rows)
    borderline = BORDERLINE.get(name, ())
    well = [r for r in rows if r[3] <= 3e-4 and r[0] not in borderline]
    ill = [r for r in rows if r[3] > 3e-4 or r[0] in borderline]
    for r in rows:
        if r[0] in borderline:
            print("%s: borderline tensor %s held to the ill-conditioned rule: HIP %.2e, perturbed oracle %.2e" % (name, r[0], r[1], r[3]))
    worst = max(well, key=lambda r: r[1]) if well else ("-", 0.0)
    worst_ill = max(ill, key=lambda r: r[1] / r[3]) if ill else ("-", 0.0, 0.0, 1.0)
    print("%s: fake deviation %.1e | all %d tensors: HIP median %.2e, oracle-f32 median %.2e, perturbed oracle-f32 median "
          "%.2e | %d well conditioned: HIP worst %.2e (%s) | %d ill conditioned: worst HIP/perturbed-oracle ratio %.1f (%s: "
          "%.1e vs %.1e)" % (name, dev, len(rows), med(ehs), med(ecs), med(eps_), len(well), worst[1], worst[0], len(ill),
                             worst_ill[1] / worst_ill[3], worst_ill[0], worst_ill[1], worst_ill[3]))
    assert worst[1] <= 1e-3, worst
    assert med(ehs) <= max(1.5 * med(eps_), 3e-4), (med(ehs), med(eps_))
    for r in ill:
        assert r[1] <= max(5 * r[3], 3e-3), r


@pytest.mark.parametrize("name,over", [
    # (BASELINE configs[3], guided 8x 32 -> 256, at full size: test_benchmark_path_matches_oracle[guided_8x_256], bs = 8)
    # BASELINE configs[4]: independent 32x 16 -> 512 (PureSEAN tail, max_fm_size 256 < 512: the capped path at 512^2)
    ("indep_16to512", dict(batchSize=1, start_size=16, crop_size=512, load_size=512, add_noise=False)),
])
def test_full_size_forward_and_losses(name, over):
    """Full-size forward parity of the two configurations the benchmark does not run: generator + discriminator + VGG
    forward and the three generator losses at bs = 1 against the CPU oracle (forward only: no backward on the CPU)."""
    from deepsee_amd import networks as N
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    oopt = O.make_opt(**over)
    states = O.recipe_state(oopt, gain=1.0)
    batch = O.synthetic_batch(oopt, 1, seed=31)
    ctl = O.RecordingCtl()
    orc = O.Oracle(oopt, states, ctl)
    random.seed(5)
    torch.manual_seed(5)
    with torch.no_grad():
        gl, fake = orc.generator_losses(orc.preprocess({k: v.clone() for k, v in batch.items()}))
    tm = TrainerManager(make_opt(**over))
    tm.sr_model.load_states(states)
    tm.sr_model.noise = N.ReplayNoise(ctl.tape)
    with torch.no_grad():
        hgl, hfake = tm.sr_model(tm.preprocess_input({k: v.clone() for k, v in batch.items()}), mode="generator")
    torch.cuda.synchronize()
    dev = rel(hfake.cpu(), fake)
    print("%s: |fake - oracle| / |oracle| = %.2e, losses %s" % (name, dev, {k: float(v) for k, v in hgl.items()}))
    assert dev < 1e-4, dev
    for k, v in gl.items():
        assert abs(float(hgl[k]) - float(v)) <= 1e-4 * abs(float(v)), (k, float(hgl[k]), float(v))


def test_inference_mode_matches_oracle():
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    over = CASES["indep_8to64_ngf8"]
    oopt = O.make_opt(**over)
    states = O.recipe_state(oopt, gain=1.0)
    batch = O.synthetic_batch(oopt, 2, seed=77)
    orc = O.Oracle(oopt, states)
    want = orc.inference({k: v.clone() for k, v in batch.items()})
    tm = TrainerManager(make_opt(**over))
    tm.sr_model.load_states(states)
    tm.sr_model.eval()
    out = tm.sr_model(tm.preprocess_input({k: v.clone() for k, v in batch.items()}), mode="inference")
    tm.sr_model.train()
    assert rel(out["fake_image"].cpu(), want) < 1e-4
    # encode_only / demo (sr_model.py:92-108)
    tm.sr_model.eval()
    pre = tm.preprocess_input({k: v.clone() for k, v in batch.items()})
    style = tm.sr_model(dict(pre), mode="encode_only")
    want_style = orc.encode_only({k: v.clone() for k, v in batch.items()})
    assert tuple(style.shape) == tuple(want_style.shape) and rel(style.detach().cpu(), want_style) < 1e-4
    given = (want_style * 0.5 + 0.1).clamp(-1, 1)
    pre["encoded_style"] = given
    demo = tm.sr_model(pre, mode="demo")
    tm.sr_model.train()
    assert rel(demo["fake_image"].cpu(), orc.demo({k: v.clone() for k, v in batch.items()}, given)) < 1e-4
    with pytest.raises(ValueError):
        tm.sr_model({}, mode="bogus")


def test_checkpoint_roundtrip_reference_layout(tmp_path):
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    over = dict(CASES["indep_4to32_ngf8"], checkpoints_dir=str(tmp_path), name="ck")
    tm = TrainerManager(make_opt(**over))
    tm.save("latest")
    spec = O.net_specs(O.make_opt(**CASES["indep_4to32_ngf8"]))
    for label in ("SR", "D", "E"):
        ck = torch.load(str(tmp_path / "ck" / ("latest_net_%s.pth" % label)))
        assert list(ck.keys()) == ["model"]
        assert {k: tuple(v.shape) for k, v in ck["model"].items()} == {k: tuple(s) for k, s in spec[label].items()}
    tm2 = TrainerManager(make_opt(**dict(over, continue_train=True, seed=5)))
    for a, b in zip(tm.sr_model.netSR.state_dict().values(), tm2.sr_model.netSR.state_dict().values()):
        assert torch.equal(a, b)


def test_data_parallel_path_on_one_gpu_nccl_world1():
    """The multi-GPU code executed on the MI355X: a 1-rank RCCL process group, parallel.attach with
    the collectives forced on -- rank-0 broadcast of parameters / buffers, MAX-reduce of the `touched` flags, chunked
    asynchronous all-reduce of the flat gradient with one Adam launch per chunk, SyncBN's all-gather / all-reduce --
    must reproduce the plain single-process step (world = 1: every collective is the identity)."""
    import socket
    import torch.distributed as dist
    from deepsee_amd import networks as N, ops, parallel
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    over = dict(start_size=8, crop_size=64, load_size=64, batchSize=2, ngf=8)
    batch = O.synthetic_batch(O.make_opt(**over), 2, seed=91)

    def steps(tm):
        out = []
        for _ in range(2):
            tm.run_generator_one_step({k: v.clone() for k, v in batch.items()})
            tm.run_discriminator_one_step({k: v.clone() for k, v in batch.items()})
            out.append({k: float(v) for k, v in tm.get_latest_losses().items()})
        torch.cuda.synchronize()
        return out, tm.optimizer_G.flat.detach().cpu().clone(), tm.optimizer_G.steps()

    # (SyncBN computes its statistics in a pass of its own; the plain run does the same here so that the two runs differ
    # by the collectives only -- beta1 = 0 Adam turns any rounding difference of step 1 into +-lr differences at step 2)
    plain = steps(TrainerManager(make_opt(seed=3, hip_graphs=False, kernel_plan=dict(producer_stats=False), **over)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        tm = TrainerManager(make_opt(seed=3, hip_graphs=False, sync_bn=True, sync_bn_clamp=False,
                                     kernel_plan=dict(producer_stats=False), **over))
        parallel.attach(tm, 1, chunk_mb=0.25, force=True)
        assert tm.optimizer_G.reduce_hook.active and tm.sr_model.plan.sync_bn is not None
        assert len(tm.optimizer_G.chunk_ranges(tm.optimizer_G.reduce_hook.chunk_elems)) > 4
        dp = steps(tm)
    finally:
        dist.destroy_process_group()
    for a, b in zip(plain[0], dp[0]):
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-4 * abs(a[k]), (k, a[k], b[k])
    assert plain[2].tolist() == dp[2].tolist()
    # beta1 = 0 Adam: elements whose gradient is rounding noise may step the other way (2 steps x lr 1e-4)
    assert float((plain[1] - dp[1]).abs().max()) <= 2.5 * 2e-4
    assert float((plain[1] - dp[1]).abs().mean()) <= 2e-5


def _rccl_world1_steps(in_graph, port, q=None, dp_comm="torch", sync_bn=False):
    """8 G+D iterations of a small model on a forced 1-rank RCCL group with hipGraphs on; returns (losses, G params, D params,
    graph stats).  `in_graph`: opt.dp_graph_collectives.  `dp_comm` = "capi": the collectives through dsee_comm_* of the C ABI,
    with no torch.distributed process group at all."""
    import torch.distributed as dist
    from deepsee_amd import parallel
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    over = dict(start_size=8, crop_size=64, load_size=64, batchSize=2, ngf=8)
    batch = O.synthetic_batch(O.make_opt(**over), 2, seed=17)
    if dp_comm == "torch":
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    tm = None
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            tm = TrainerManager(make_opt(seed=5, hip_graphs=True, dp_graph_collectives=in_graph, dp_comm=dp_comm,
                                         sync_bn=sync_bn, sync_bn_clamp=False, **over))
        parallel.attach(tm, 1, chunk_mb=0.25, force=True)
        assert (tm.dp_comm is not None) == (dp_comm == "capi")
        assert tm.use_graphs and tm.dp_in_graph == in_graph and tm.optimizer_G.reduce_hook.active
        assert len(tm.optimizer_G.chunk_ranges(tm.optimizer_G.reduce_hook.chunk_elems)) > 4
        out = []
        for _ in range(8):
            tm.run_generator_one_step({k: v.clone() for k, v in batch.items()})
            tm.run_discriminator_one_step({k: v.clone() for k, v in batch.items()})
            out.append({k: float(v) for k, v in tm.get_latest_losses().items()})
        torch.cuda.synchronize()
        assert all(rec["eager_opt"] != in_graph for rec in tm._graphs.values())
        res = (out, tm.optimizer_G.flat.detach().cpu().clone(), tm.optimizer_D.flat.detach().cpu().clone(), dict(tm.graph_stats))
    finally:
        if tm is not None:
            tm.release_graphs()       # captured RCCL operations must not outlive their communicator
            if tm.dp_comm is not None:
                tm.dp_comm.close()
        if dp_comm == "torch":
            dist.destroy_process_group()
    if q is not None:     # (plain bytes: tensors in a multiprocessing queue are shared-memory handles that die with the child)
        q.put((res[0], res[1].numpy().tobytes(), res[2].numpy().tobytes(), res[3]))
    return res


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_hip_graphs_with_rccl_group_world1():
    """The data-parallel step as bench.py --gpus N runs it by default -- forward + backward + gradient gather replayed as a
    hipGraph, then the chunked RCCL all-reduce and the per-chunk Adam launches issued eagerly -- on a 1-rank RCCL group (every
    collective is the identity): capture has to work while RCCL's watchdog thread is alive, and eight iterations (eager,
    capture, replays) must track the same model stepped eagerly without a process group."""
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    over = dict(start_size=8, crop_size=64, load_size=64, batchSize=2, ngf=8)
    batch = O.synthetic_batch(O.make_opt(**over), 2, seed=17)
    tm = TrainerManager(make_opt(seed=5, hip_graphs=False, **over))
    plain = []
    for _ in range(8):
        tm.run_generator_one_step({k: v.clone() for k, v in batch.items()})
        tm.run_discriminator_one_step({k: v.clone() for k, v in batch.items()})
        plain.append({k: float(v) for k, v in tm.get_latest_losses().items()})
    torch.cuda.synchronize()
    pg, pd = tm.optimizer_G.flat.detach().cpu().clone(), tm.optimizer_D.flat.detach().cpu().clone()
    dp = _rccl_world1_steps(False, _free_port())
    stats = dp[3]
    assert stats["captured"] >= 2 and stats["replayed"] >= 2, stats   # (every encoder-branch variant: eager, capture, replays)
    for it, (a, b) in enumerate(zip(plain, dp[0])):
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-3 * abs(a[k]) + 1e-6, (it, k, a[k], b[k])
    for x, y in ((pg, dp[1]), (pd, dp[2])):
        assert float((x - y).abs().max()) <= 2.5 * 8 * 4e-4           # beta1 = 0 Adam: <= lr per step and element
        assert float((x - y).abs().mean()) <= 5e-5


def test_hip_graphs_default_on_partial_batches_and_shape_eviction():
    """opt.hip_graphs is ON by default for a training manager (a maintainer following INTEGRATION 1 gets the replayed path, not
    60 ms of Python enqueue per step).  A loader without drop_last hands over a partial last batch: graphs, static input buffers
    and the eager-first counters are keyed by the batch shape, so the sequence full, full, full, partial, full, partial, partial,
    full ... must equal the same sequence stepped eagerly (hip_graphs = False) -- with room for both shapes
    (max_graph_shapes = 4) and with room for ONE (every shape change evicts the other shape's graphs and buffers and
    re-captures)."""
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    over = dict(start_size=4, crop_size=32, load_size=32, batchSize=4, ngf=8, seed=21)
    assert make_opt().hip_graphs is True
    full = O.synthetic_batch(O.make_opt(**over), 4, seed=8)
    part = {k: v[:2].clone() for k, v in full.items()}
    seq = [full, full, full, part, full, part, part, full, full, part, part, part, full, full]

    def run(**kw):
        tm = TrainerManager(make_opt(**over, **kw))
        out = []
        for b in seq:
            tm.run_generator_one_step({k: v.clone() for k, v in b.items()})
            tm.run_discriminator_one_step({k: v.clone() for k, v in b.items()})
            out.append({k: float(v) for k, v in tm.get_latest_losses().items()})
            assert tuple(tm.get_latest_generated().shape)[0] == b["image"].shape[0]
        torch.cuda.synchronize()
        flat = (tm.optimizer_G.flat.detach().cpu().clone(), tm.optimizer_D.flat.detach().cpu().clone())
        stats = dict(tm.graph_stats)
        tm.close()
        return out, flat, stats

    eager = run(hip_graphs=False)
    assert eager[2]["captured"] == 0 and eager[2]["replayed"] == 0
    both = run()
    assert both[2]["captured"] >= 2 and both[2]["replayed"] >= 2 and both[2]["evicted_shapes"] == 0, both[2]
    one = run(max_graph_shapes=1)
    assert one[2]["evicted_shapes"] >= 5, one[2]
    # (not bit-identical: a replay and an eager pass may order a few float reductions differently, and beta1 = 0 Adam turns a
    # rounding-level difference of a near-zero gradient element into a visible fraction of lr -- same yardstick as
    # test_hip_graphs_with_rccl_group_world1: <= lr per step and element at worst, tiny on average)
    for got in (both, one):
        for it, (a, b) in enumerate(zip(eager[0], got[0])):
            for k in a:
                assert abs(a[k] - b[k]) <= 2e-3 * abs(a[k]) + 1e-6, (it, k, a[k], b[k])
        for x, y in zip(eager[1], got[1]):
            assert float((x - y).abs().max()) <= 2.5 * len(seq) * 4e-4, float((x - y).abs().max())
            assert float((x - y).abs().mean()) <= 5e-5, float((x - y).abs().mean())


def _child_result(p, q, timeout=420):
    """The result a child process puts into `q`, or None as soon as the child is dead without having delivered one (a child
    that aborts must not cost the suite the whole timeout)."""
    import queue
    import time
    t_end = time.time() + timeout
    while time.time() < t_end:
        try:
            return q.get(timeout=1.0)
        except queue.Empty:
            if not p.is_alive():
                try:
                    return q.get(timeout=1.0)      # (delivered just before it exited)
                except queue.Empty:
                    return None
    return None


def test_dp_collectives_captured_inside_the_graph_world1():
    """opt.dp_graph_collectives (round 4, OFF by default): the chunked RCCL all-reduce and the per-chunk Adam launches captured
    INSIDE the hipGraph, so a data-parallel rank's host does one launch per half step.  Must be bit-identical to the default
    schedule (collectives + Adam eager behind the replayed graph).  Capturing RCCL operations aborts inside the HIP runtime in
    about one run out of six on this stack (ROCm 7.0.2, RCCL 2.26.6: SIGABRT from the launch that follows a captured
    collective, profiles/r04_notes.md) -- which is why the option is off by default and why both runs happen in child
    processes: an abort of the in-graph child is reported as a skip with that reason, never as a pass."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for in_graph in (False, True):
        q = ctx.Queue()
        p = ctx.Process(target=_rccl_world1_steps, args=(in_graph, _free_port(), q))
        p.start()
        res[in_graph] = _child_result(p, q)
        p.join(20)
        if p.is_alive():
            p.kill()
        if res[in_graph] is None:
            if in_graph:
                pytest.skip("the child that captures RCCL collectives inside the graph died (exit code %s): the known "
                            "intermittent HIP-runtime abort; the option stays off by default" % p.exitcode)
            raise AssertionError("the eager-collectives child died (exit code %s)" % p.exitcode)
    a, b = res[True], res[False]
    assert a[0] == b[0]
    assert a[1] == b[1] and a[2] == b[2] and len(a[1]) > 1000      # the flat G / D parameter buffers, byte for byte


@pytest.mark.parametrize("sync_bn", [True, pytest.param(False, marks=pytest.mark.slow)])   # (False: a subset of the exchanges, two more child processes)
def test_data_parallel_through_the_c_abi_communicator_world1(sync_bn):
    """opt.dp_comm = "capi": the gradient all-reduce (per chunk, on the communicator's side stream, each chunk followed by its
    Adam launch), the start-state broadcast and -- with opt.sync_bn -- the SyncBN statistics all-gather / sum all-reduce go
    through dsee_comm_* of include/deepsee_hip.h (RCCL resolved with dlopen inside libdeepsee_hip.so), with NO torch.distributed
    process group.  On a 1-rank communicator every collective is the identity, so the run must equal, byte for byte, the same
    schedule over torch.distributed's 1-rank RCCL group.  (Child processes: one RCCL bootstrap each.)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for dp_comm in ("torch", "capi"):
        q = ctx.Queue()
        p = ctx.Process(target=_rccl_world1_steps, args=(False, _free_port(), q, dp_comm, sync_bn))
        p.start()
        res[dp_comm] = _child_result(p, q)
        p.join(20)
        if p.is_alive():
            p.kill()
        assert res[dp_comm] is not None, "the %s child died (exit code %s)" % (dp_comm, p.exitcode)
    a, b = res["capi"], res["torch"]
    assert a[0] == b[0]
    assert a[1] == b[1] and a[2] == b[2] and len(a[1]) > 1000


def _two_gpu_worker(rank, world, port, sync_bn, q, dp_comm="torch"):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import warnings
    warnings.simplefilter("ignore", RuntimeWarning)
    from deepsee_amd import parallel
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    parallel.init_distributed(backend="nccl")
    over = dict(start_size=8, crop_size=64, load_size=64, batchSize=2, ngf=8, add_noise=False, noisy_style_scale=0.0)
    tm = TrainerManager(make_opt(seed=3, sync_bn=sync_bn, sync_bn_clamp=False, dp_comm=dp_comm, **over))
    parallel.attach(tm, world, chunk_mb=0.25)
    full = O.synthetic_batch(O.make_opt(**dict(over, batchSize=2 * world)), 2 * world, seed=91)
    shard = {k: v[2 * rank:2 * rank + 2].clone() for k, v in full.items()}
    for _ in range(3):
        tm.run_generator_one_step({k: v.clone() for k, v in shard.items()})
        tm.run_discriminator_one_step({k: v.clone() for k, v in shard.items()})
    torch.cuda.synchronize()
    q.put((rank, tm.optimizer_G.flat.detach().cpu(), tm.optimizer_D.flat.detach().cpu(), tm.optimizer_G.steps().tolist(),
           {k: float(v) for k, v in tm.get_latest_losses().items()}))
    import torch.distributed as dist
    dist.barrier()
    tm.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("dp_comm", ["torch", "capi"])
@pytest.mark.parametrize("sync_bn", [False, True])
def test_two_gpu_rccl_data_parallel(sync_bn, dp_comm):
    """Needs TWO MI355X (skipped otherwise): 2 ranks over RCCL, 3 G+D iterations on different batch shards.  Parameters
    and per-tensor step counts must be BIT-IDENTICAL on both ranks (chunked asynchronous all-reduce + per-chunk Adam on
    two streams, flags in the first chunk's header, identical branch coins); with opt.sync_bn the run must also track a
    single process that trains on the concatenated batch (the reference DataParallel semantics: global BN statistics,
    gradient = mean over the global batch).  dp_comm = "capi": the same exchanges through dsee_comm_* of the C ABI (torch.distributed
    then only carries the communicator's 128-byte id)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the data-parallel path on one GPU: test_data_parallel_path_on_one_gpu_nccl_world1)")
    import socket
    import torch.multiprocessing as mp
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_gpu_worker, args=(r, 2, port, sync_bn, q, dp_comm)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2]) and res[0][3] == res[1][3]
    if sync_bn:
        over = dict(start_size=8, crop_size=64, load_size=64, batchSize=4, ngf=8, add_noise=False, noisy_style_scale=0.0)
        tm = TrainerManager(make_opt(seed=3, **over))
        full = O.synthetic_batch(O.make_opt(**over), 4, seed=91)
        for _ in range(3):
            tm.run_generator_one_step({k: v.clone() for k, v in full.items()})
            tm.run_discriminator_one_step({k: v.clone() for k, v in full.items()})
        torch.cuda.synchronize()
        ref = tm.optimizer_G.flat.detach().cpu()
        # beta1 = 0 Adam moves every element by ~lr per step whatever the gradient's size: rounding-level gradient
        # differences flip individual signs, so the comparison is statistical (3 steps x lr 1e-4)
        assert float((ref - res[0][1]).abs().mean()) <= 3e-5, float((ref - res[0][1]).abs().mean())


def test_half_mode_tracks_fp32():
    """BASELINE configs[2]'s 16-bit mode (opt.precision = 'fp16': every Winograd-domain GEMM on ONE scaled fp16 term per operand
    element, stored as such -- packed one-term V / dM / weights, fp16 products M / dV -- fp32 master weights / statistics /
    Adam) against the fp32 HIP path as SURVEY 8(d) prescribes: generated image within 3e-2 after the first step, the losses of
    the first two iterations within 5 % of the fp32 run, and the loss trajectories overlapping over 50 iterations (two fp32
    runs that differ by one rounding drift apart at the same rate -- beta1 = 0 Adam steps by lr * sign(g) -- so later
    iterations are compared as window means, and the last window against a YARDSTICK: a second fp32-class run of the same
    model on the exact 3-term bf16 split instead of the two-term fp16 split, i.e. how far two fp32 trajectories of this GAN
    on a fixed batch of 2 drift apart from rounding alone).  Both norm variants of the mode run: the gamma/beta path
    one-term (plan.half_norms, default) and two-term."""
    from deepsee_amd import ops
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    over = dict(batchSize=2, seed=11)
    batch = O.synthetic_batch(O.make_opt(batchSize=2), 2, seed=5)
    iters = 50
    runs = {}
    for name, kw in (("fp32", dict(precision="fp32")), ("fp16", dict(precision="fp16")),
                     ("fp32/bf16x3", dict(precision="fp32", kernel_plan=dict(gemm_f16x2=False))),
                     ("fp16/two-term norms", dict(precision="fp16", kernel_plan=dict(half_norms=False)))):
        tm = TrainerManager(make_opt(**kw, **over))
        assert tm.sr_model.plan.half == name.startswith("fp16")
        ops.PROFILE = prof = {}
        traj, fake0 = [], None
        for it in range(iters if name != "fp16/two-term norms" else 6):
            tm.run_generator_one_step({k: v.clone() for k, v in batch.items()})
            if it == 0:
                fake0 = tm.get_latest_generated().detach().cpu()
                ops.PROFILE = None
            tm.run_discriminator_one_step({k: v.clone() for k, v in batch.items()})
            traj.append({k: float(v.detach()) for k, v in tm.get_latest_losses().items()})
        torch.cuda.synchronize()
        runs[name] = (fake0, traj, sorted(prof))
        del tm
    # the 16-bit mode really runs the packed one-term kernels (a silent fall-back to fp32 operands would pass the bounds too)
    assert "winograd_gemm_f16_1term_packed" in runs["fp16"][2] and "winograd_gemm_f16x2" not in runs["fp16"][2], runs["fp16"][2]
    assert "winograd_gemm_f16_1term_packed" in runs["fp16/two-term norms"][2]
    ref = runs["fp32"]
    for name in ("fp16", "fp16/two-term norms"):
        dev = rel(runs[name][0], ref[0])
        print("%s vs fp32: |fake| deviation %.2e" % (name, dev))
        assert dev < 3e-2, (name, dev)
        for it, (a, b) in enumerate(zip(runs[name][1], ref[1])):
            for k in a:
                assert a[k] == a[k] and abs(a[k]) < 1e4, (name, it, k, a[k])
                if it < 2:
                    assert abs(a[k] - b[k]) <= 0.05 * abs(b[k]) + 0.05, (name, it, k, a[k], b[k])
    rows = []
    for lo, hi in ((0, 10), (10, 30), (30, 50)):
        for k in ref[1][0]:
            ma = sum(d[k] for d in runs["fp16"][1][lo:hi]) / (hi - lo)
            mb = sum(d[k] for d in ref[1][lo:hi]) / (hi - lo)
            my = sum(d[k] for d in runs["fp32/bf16x3"][1][lo:hi]) / (hi - lo)
            rows.append((lo, hi, k, ma, mb, my))
    print("\n".join("iterations %2d-%2d  %-8s fp16 %9.4f   fp32 %9.4f   fp32 on the exact bf16x3 split %9.4f" % r for r in rows))
    for lo, hi, k, ma, mb, my in rows:
        tol = 0.3 * abs(mb) + 0.15
        if lo >= 30 and k in ("GAN", "D_Fake", "D_Real"):
            # the adversarial hinge terms of a GAN trained on ONE fixed batch of two images oscillate after ~30 iterations: two
            # fp32-class runs already sit 0.2 apart there (yardstick column; observed 0.95 vs 0.75 on the generator term).  Hinge
            # terms live on a scale of 0..2: held to half of it and to 4x the fp32 rounding drift, whichever is larger; the
            # feature-matching and VGG terms (what the generator is mostly trained on) keep the 30 % bound in every window.
            tol = max(1.0, 4.0 * abs(my - mb))
        elif lo >= 10 and k in ("GAN", "D_Fake", "D_Real"):
            # (round 6) the drift can set in before iteration 30: after mlp_shared's weight gradient moved to another kernel (one
            # rounding per element) the fp32 run's D_Fake mean over 10-30 read 0.43 while BOTH the 16-bit run (0.86) and the
            # yardstick run (0.88) stayed together.  The 16-bit run may be as far from the fp32 run as twice what the two fp32-class
            # runs are apart, capped at half the hinge scale
            tol = max(tol, min(1.0, 2.0 * abs(my - mb)))
        assert abs(ma - mb) <= tol, (lo, hi, k, ma, mb, my)


@pytest.mark.parametrize("bs", [1, pytest.param(8, marks=pytest.mark.slow)])
def test_half_mode_vs_oracle(bs):
    """The 16-bit mode against the CPU ORACLE (not only against the fp32 HIP path): one G step of BASELINE configs[1]'s
    geometry on identical weights, inputs, noise and branch decisions at bs = 1 (the benchmark's bs = 8 shares the oracle pass
    of test_benchmark_path_matches_oracle[independent_8x_256]): generated image within SURVEY 8(d)'s 3e-2, generator losses
    within 5 %, parameter gradients within HALF_GRAD_*_BOUND."""
    import os
    from deepsee_amd import networks as N, ops
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    if bs > 1:
        try:
            ram_gb = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 1e9
        except (ValueError, OSError):
            ram_gb = 0.0
        if ram_gb < 90:
            pytest.skip("the fp32 oracle at bs = 8 needs ~60 GB of host memory (%.0f GB here)" % ram_gb)
    over = dict(batchSize=bs)
    oopt = O.make_opt(**over)
    states = O.recipe_state(oopt, gain=1.0)
    batch = O.synthetic_batch(oopt, bs, seed=31)
    ctl = O.RecordingCtl()
    orc = O.Oracle(oopt, states, ctl)
    orc.create_optimizers()
    random.seed(31)
    torch.manual_seed(31)
    gl, fake = orc.run_generator_one_step({k: v.clone() for k, v in batch.items()})
    tm = TrainerManager(make_opt(precision="fp16", **over))
    assert tm.sr_model.plan.half and tm.sr_model.plan.half_norms
    tm.sr_model.load_states(states)
    tm.sr_model.noise = N.ReplayNoise(ctl.tape)
    tm.run_generator_one_step({k: v.clone() for k, v in batch.items()})
    torch.cuda.synchronize()
    hfake = tm.get_latest_generated().detach().cpu()
    hgl = {k: float(v) for k, v in tm.g_losses.items()}
    dev = rel(hfake, fake.detach())
    print("fp16 mode vs the CPU oracle at bs = %d: |fake - oracle| / |oracle| = %.2e, losses %s vs %s"
          % (bs, dev, {k: round(v, 4) for k, v in hgl.items()}, {k: round(float(v.detach()), 4) for k, v in gl.items()}))
    assert dev < 3e-2, dev
    for k, v in gl.items():
        assert abs(hgl[k] - float(v.detach())) <= 0.05 * abs(float(v.detach())) + 1e-3, (k, hgl[k], float(v.detach()))
    # gradients of the 16-bit mode against the ORACLE: every G / E parameter tensor the step reached, relative
    # to its own norm (floored at 1e-3 of the largest gradient norm).  One-term fp16 operands carry 2^-11 per element and
    # F(4x4,3x3) amplifies it ~10x per layer, on top of the LeakyReLU-kink floor the fp32 path already has.
    ggrads = {"%s.%s" % (net, k): p.grad.clone() for net in ("SR", "E") for k, p in orc.params(net) if p.grad is not None}
    hg = {nm: _grad_or_zero(p) for nm, p in zip(tm.optimizer_G.names, tm.optimizer_G.params)}
    gmax = max(float(v.norm()) for v in ggrads.values())
    errs = sorted(float((hg[k].double() - v.double()).norm()) / max(float(v.norm()), 1e-3 * gmax) for k, v in ggrads.items())
    med, q90, worst = errs[len(errs) // 2], errs[int(0.9 * (len(errs) - 1))], errs[-1]
    print("fp16 mode vs the CPU oracle at bs = %d: G-gradient rel err median %.2e, 90 %% %.2e, max %.2e over %d tensors"
          % (bs, med, q90, worst, len(errs)))
    assert med < HALF_GRAD_MEDIAN_BOUND and worst < HALF_GRAD_MAX_BOUND, (med, q90, worst)


SWITCHES = [("keep_v", False), ("adjoint_dgrad", False), ("fuse_dm", False), ("fuse_noise", False), ("gemm_af32", False),
            ("fused_norm", False), ("thin_gemm", False), ("gemm_f16x2", False), ("gemm_split", False),
            ("winograd_wgrad", False), ("winograd_mod", False), ("conv_f16x2_min_flop", 0.0), ("dout_sums", False),
            ("share_stats", False), ("producer_stats", False), ("presplit_a", False), ("presplit_dm", False),
            ("presplit_gb", False), ("sign_mask", False), ("defer_act", False), ("gemm_w4", False), ("dgrad_s2_parity", False), ("conv_amax_out", False), ("conv_halo_f16", False), ("onehot_wgrad_mfma", False), ("fused_w4", True), ("branch_streams", None)]      # None: the other side of a bool


def test_kernel_path_switches():
    """A model's KernelPlan (deepsee_amd/plan.py; opt.kernel_plan) selects between kernel paths that compute the same
    function -- per model, no module attributes, no environment variables.  Every field's non-default side runs one G + D step of the benchmark's geometry (32 -> 256,
    256-channel generator so the 256-row weight-gradient tiles, the fused SPADE kernel and the adjoint data gradient are
    all in play) from the same weights, inputs and device noise, and must reproduce the default path: generated image
    <= 1e-5, losses <= 1e-4, gradients to rounding order (see the tolerance note below)."""
    from deepsee_amd import ops
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    over = dict(batchSize=2, ngf=16, seed=3)
    batch = O.synthetic_batch(O.make_opt(**over), 2, seed=77)
    states = O.recipe_state(O.make_opt(**over), gain=1.0)

    def one(**plan):
        tm = TrainerManager(make_opt(kernel_plan=plan, **over))
        assert all(getattr(tm.sr_model.plan, k) == v for k, v in plan.items())
        tm.sr_model.load_states(states)
        tm.run_generator_one_step({k: v.clone() for k, v in batch.items()})
        torch.cuda.synchronize()
        g = {nm: _grad_or_zero(p) for nm, p in zip(tm.optimizer_G.names, tm.optimizer_G.params)}
        fake = tm.get_latest_generated().detach().cpu()
        gl = {k: float(v) for k, v in tm.g_losses.items()}
        tm.sr_model.load_states(states)          # the D step from identical weights too
        tm.run_discriminator_one_step({k: v.clone() for k, v in batch.items()})
        torch.cuda.synchronize()
        d = {nm: _grad_or_zero(p) for nm, p in zip(tm.optimizer_D.names, tm.optimizer_D.params)}
        return fake, gl, g, d, {k: float(v) for k, v in tm.d_losses.items()}

    ref = one()
    gmax = max(float(v.norm()) for v in ref[2].values())
    dmax = max(float(v.norm()) for v in ref[3].values())
    report, bad = [], []
    for name, value in SWITCHES:
        if value is None:      # (a switch whose default a DSEE_PLAN override may flip: test whichever side is not the default)
            value = not getattr(ops.DEFAULT_PLAN, name)
        assert getattr(ops.DEFAULT_PLAN, name) != value, name
        got = one(**{name: value})       # a model with its own plan: nothing process-wide is touched
        assert ops.P() is ops.DEFAULT_PLAN
        dev = rel(got[0], ref[0])
        gerr = sorted((float((got[2][k] - v).norm()) / max(float(v.norm()), 1e-2 * gmax), k) for k, v in ref[2].items())
        derr = sorted((float((got[3][k] - v).norm()) / max(float(v.norm()), 1e-2 * dmax), k) for k, v in ref[3].items())
        (ge, gk), (de, dk) = gerr[-1], derr[-1]
        gm, dm = gerr[len(gerr) // 2][0], derr[len(derr) // 2][0]
        report.append("%s=%s: fake %.1e, G-grad median %.1e max %.1e (%s), D-grad median %.1e max %.1e (%s)"
                      % (name, value, dev, gm, ge, gk, dm, de, dk))
        # a switch that leaves the forward arithmetic alone must reproduce the gradients to rounding order; one that
        # changes forward roundings (fake differs in the last bits) flips LeakyReLU slopes of near-zero activations: the
        # gradients then agree like two fp32 implementations do (same floor as against the oracle, test_train_step_*)
        tol_max, tol_med = (1e-4, 1e-5) if dev == 0.0 else (2e-2, 5e-3)
        if dev > 1e-5 or max(ge, de) > tol_max or max(gm, dm) > tol_med:
            bad.append(report[-1])
        for k, v in list(ref[1].items()) + list(ref[4].items()):
            w = got[1].get(k, got[4].get(k))
            if abs(w - v) > 1e-4 * abs(v) + 1e-7:
                bad.append("%s: loss %s %.6g vs %.6g" % (name, k, w, v))
    print("\n".join(report))
    assert not bad, "\n".join(bad)


def test_training_loop_with_device_loader_and_metrics(tmp_path):
    """The pieces either side of the step together (SURVEY 8 f3 + f4 + a2): uint8 batches from DeviceLoader (prefetched on
    a side stream) straight into run_generator_one_step / run_discriminator_one_step for an epoch, a learning-rate
    update, a checkpoint, then PSNR / SSIM / RMSE of the generated images through the MetricsEvaluator mirror -- and the
    same epoch fed as the reference's float CPU tensors gives the same losses (the device pipeline is bit-exact)."""
    from deepsee_amd import data as D, metrics as M, ops
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    over = dict(start_size=8, crop_size=64, load_size=64, batchSize=2, ngf=8, checkpoints_dir=str(tmp_path), name="loop",
                niter=1, niter_decay=1)

    def run(native):
        tm = TrainerManager(make_opt(seed=11, **over))
        ds = D.SyntheticDataset(tm.opt, length=6, seed=4)
        loader = D.DeviceLoader(ds, tm.opt, shuffle=True, seed=9)
        ev = M.MetricsEvaluator()
        losses = []
        for batch in loader:
            if not native:     # the reference's wire format: float label [N,1,H,W], float image [N,3,H,W] on the CPU
                lab = batch["input_semantics"].t.cpu().float()[:, None]
                batch = {"label": lab, "image": ops.to_nchw(batch["image_hr"], 3).cpu(), "path": batch["path"]}
            tm.run_generator_one_step(batch)
            tm.run_discriminator_one_step(batch)
            losses.append({k: float(v) for k, v in tm.get_latest_losses().items()})
            real = batch["image_hr"] if native else batch["image"]
            ev.collect_samples(tm.get_latest_generated(), real, name=batch["path"])
        tm.update_learning_rate(1)
        tm.save("latest")
        torch.cuda.synchronize()
        return losses, ev.get_result(), tm

    la, ra, tm = run(True)
    lb, rb, _ = run(False)
    assert len(la) == 3 and ra["n_samples"] == 6
    for a, b in zip(la, lb):
        for k in a:
            assert a[k] == b[k] or abs(a[k] - b[k]) <= 1e-6 * abs(b[k]), (k, a[k], b[k])
    assert abs(ra["psnr/mean"] - rb["psnr/mean"]) <= 1e-9 and 0.0 < ra["psnr/mean"] < 60.0 and -1.0 <= ra["ssim/mean"] <= 1.0
    assert all(math.isfinite(v) for d in la for v in d.values())
    assert (tmp_path / "loop" / "latest_net_SR.pth").exists()
    assert abs(tm.optimizer_G.param_groups[0]["lr"] - tm.opt.lr / 2 * 1.0) < 1e-12      # epoch 1 <= niter: no decay yet


def test_vgg_weights_option_loads_torchvision_state_dict(tmp_path):
    """opt.vgg_weights end to end (architecture.py:154: the reference downloads torchvision's vgg19): a state dict with
    torchvision's key names -- ALL of vgg19(): 'features.N.*' for the 16 convolutions incl. the three behind the last
    tap, and the classifier -- saved with torch.save and named by the option.  The model must load the 13 tap
    convolutions, ignore the rest, stop warning about random features, and its five taps must equal the oracle's VGG on
    the same weights."""
    import warnings
    from deepsee_amd import ops
    from deepsee_amd.sr_model import SRModel
    from deepsee_amd.options import make_opt
    g = torch.Generator().manual_seed(5)
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
    state, cin, idx = {}, 3, 0
    for v in cfg:
        if v == "M":
            idx += 1
            continue
        state["features.%d.weight" % idx] = torch.randn(v, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        state["features.%d.bias" % idx] = torch.randn(v, generator=g) * 0.05
        cin, idx = v, idx + 2
    for i, (o, k) in zip((0, 3, 6), ((64, 128), (64, 64), (10, 64))):   # (classifier keys, shapes irrelevant: ignored)
        state["classifier.%d.weight" % i] = torch.randn(o, k, generator=g)
        state["classifier.%d.bias" % i] = torch.randn(o, generator=g)
    path = str(tmp_path / "vgg19-torchvision-keys.pth")
    torch.save(state, path)
    over = dict(start_size=4, crop_size=32, load_size=32, batchSize=2, ngf=8)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)     # the "no pretrained VGG19 weights" notice must NOT fire
        m = SRModel(make_opt(vgg_weights=path, **over))
    assert m.vgg_pretrained
    own = m.vgg.state_dict()
    assert len(own) == 26 and all(k in state for k in own)
    for k, v in own.items():
        assert torch.equal(v.detach().cpu().reshape(state[k].shape), state[k]), k
    # a state dict that lacks a tap convolution is refused
    broken = {k: v for k, v in state.items() if not k.startswith("features.28.")}
    with pytest.raises(RuntimeError):
        m.load_vgg_state(broken)
    # the state dict of vgg19().features alone ('N.weight') is accepted as well
    m.load_vgg_state({k[len("features."):]: v for k, v in state.items() if k.startswith("features.")})
    # taps vs the oracle on the same weights
    oopt = O.make_opt(**over)
    states = O.recipe_state(oopt, gain=1.0)
    states["VGG"] = {k: state[k].clone() for k in O.vgg_spec()}
    orc = O.Oracle(oopt, states, O.RecordingCtl())
    x = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    want = orc.vgg_features(x)
    got = m.vgg(ops.to_nhwc(x.cuda()))
    for i, (a, b) in enumerate(zip(got, want)):
        assert rel(ops.to_nchw(a, b.shape[1]).cpu(), b) < 1e-5, i
