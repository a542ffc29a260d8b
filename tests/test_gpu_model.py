"""End-to-end parity of the HIP train step (TrainerManager on MI355X) against the CPU oracle on identical recipe
weights, inputs, noise tensors and branch decisions.  Outputs/losses are held to 1e-3 rel (north_star) — in
practice ~1e-5; gradients to the reference's own noise floor (see tests/test_oracle_golden.py)."""
import random

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
    "config1_4to32_full": dict(start_size=4, crop_size=32, load_size=32, batchSize=2),
    "puresean_4to128_ngf4": dict(start_size=4, crop_size=128, load_size=512, batchSize=2, ngf=4, add_noise=False,
                                 max_fm_size=64),
    "clip_nottur_4to32": dict(start_size=4, crop_size=32, load_size=32, batchSize=2, ngf=8, add_noise=False,
                              no_TTUR=True, gradient_clip=0.01),
}


def run_case(over, seed, iters=1):
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
        dl = orc.run_discriminator_one_step({k: v.clone() for k, v in batch.items()})
        dgrads = {"D." + k: p.grad.clone() for k, p in orc.params("D") if p.grad is not None}
        tm.sr_model.noise = N.ReplayNoise(ctl.tape[start:])
        tm.run_generator_one_step({k: v.clone() for k, v in batch.items()})
        torch.cuda.synchronize()
        hg = {nm: p.grad.detach().cpu().clone() for nm, p in zip(tm.optimizer_G.names, tm.optimizer_G.params)}
        touched_g = {nm for nm, t in zip(tm.optimizer_G.names, tm.optimizer_G.touched) if t}
        hgl = {k: float(v) for k, v in tm.g_losses.items()}
        hfake = tm.get_latest_generated().detach().cpu()
        tm.run_discriminator_one_step({k: v.clone() for k, v in batch.items()})
        torch.cuda.synchronize()
        hd = {nm: p.grad.detach().cpu().clone() for nm, p in zip(tm.optimizer_D.names, tm.optimizer_D.params)}
        hdl = {k: float(v) for k, v in tm.d_losses.items()}
        assert tm.sr_model.noise.pos == len(tm.sr_model.noise.tape)
        out.append(dict(gl={k: float(v.detach()) for k, v in gl.items()}, fake=fake.detach(), ggrads=ggrads,
                        dl={k: float(v.detach()) for k, v in dl.items()}, dgrads=dgrads, hgl=hgl, hfake=hfake, hg=hg,
                        hd=hd, hdl=hdl, touched_g=touched_g))
    return orc, tm, out


@pytest.mark.parametrize("name", list(CASES))
def test_train_step_matches_oracle(name):
    orc, tm, out = run_case(CASES[name], seed=101 + len(name))
    r = out[0]
    for k, v in r["gl"].items():
        assert abs(r["hgl"][k] - v) <= 1e-3 * abs(v), (k, r["hgl"][k], v)
    assert rel(r["hfake"], r["fake"]) < 1e-3
    # the set of tensors that received a gradient equals the reference's "grad is not None" set
    assert r["touched_g"] == set(r["ggrads"]), r["touched_g"] ^ set(r["ggrads"])
    gmax = max(float(v.norm()) for v in r["ggrads"].values())
    worst = 0.0
    for k, v in r["ggrads"].items():
        e = float((r["hg"][k].double() - v.double()).norm()) / max(float(v.norm()), 1e-3 * gmax)
        worst = max(worst, e)
        assert e < 5e-3, (k, e)
    for k, v in r["dl"].items():
        assert abs(r["hdl"][k] - v) <= 2e-3 * abs(v), (k, r["hdl"][k], v)
    dmax = max(float(v.norm()) for v in r["dgrads"].values())
    for k, v in r["dgrads"].items():
        e = float((r["hd"][k].double() - v.double()).norm()) / max(float(v.norm()), 1e-3 * dmax)
        assert e < 5e-2, (k, e)
    # post-step state: parameters (Adam) and buffers (BN running stats twice, SN u/v twice)
    sd = {"SR": tm.sr_model.netSR.state_dict(), "D": tm.sr_model.netD.state_dict(), "E": tm.sr_model.netE.state_dict()}
    zero_grad = {k for k, v in {**r["ggrads"], **r["dgrads"]}.items() if float(v.norm()) < 1e-4 * max(gmax, dmax)}
    for net in ("SR", "D", "E"):
        assert list(sd[net].keys()) == list(orc.S[net].keys())
        for k, v in sd[net].items():
            if "%s.%s" % (net, k) in zero_grad or not v.is_floating_point():
                continue
            assert rel(v.cpu(), orc.S[net][k].detach()) < 1e-3, (net, k)
    print("worst G-grad rel err %.2e" % worst)


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
