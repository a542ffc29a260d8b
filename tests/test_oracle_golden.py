"""The oracle (oracle/deepsee_oracle.py) against the fixtures that oracle/gen_golden.py wrote
from the REAL reference (numbers in tests/golden/*.json are the reference's outputs).
CPU only; no /root/reference needed."""
import glob
import json
import os
import random

import pytest
import torch
import torch.nn.functional as F

from oracle import deepsee_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLD, "*.json"))
               if not p.endswith(("layer_kats.json", "host_logic.json", "metrics.json")))

# Bounds: losses / outputs are tight; gradients are bounded by the reference's OWN noise floor
# (oracle/noise_floor.py: a 1e-7 relative input perturbation moves G-step grads by 2.6e-3 and
# D-step grads by 2.3e-2, because L1/hinge gradients are sign functions and beta1=0 Adam is sign-like).
TOL_OUT, TOL_DLOSS, TOL_GG, TOL_DG = 1e-4, 2e-3, 5e-3, 5e-2


def slice_of(t, k=64):
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(k, f.numel())).long()
    return f[idx]


def close(a, b, tol):
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm()) <= tol * max(float(b.norm()), 1e-12)


def test_cases_present():
    assert len(CASES) >= 5


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_fixture(case):
    rec = json.load(open(os.path.join(GOLD, case + ".json")))
    opt = O.make_opt(**rec["opt"])
    states = O.recipe_state(opt, gain=1.0)
    batch = O.synthetic_batch(opt, rec["n"], seed=rec["batch_seed"])
    orc = O.Oracle(opt, states)
    orc.create_optimizers()
    assert [len(g["params"]) for g in orc.opt_G.param_groups] == rec["opt_groups_G"]
    assert [len(g["params"]) for g in orc.opt_D.param_groups] == rec["opt_groups_D"]

    random.seed(rec["rng_seed"])
    torch.manual_seed(rec["rng_seed"])
    fake = orc.inference({k: v.clone() for k, v in batch.items()})
    assert close(slice_of(fake), rec["inference"]["fake_slice"], TOL_OUT)
    assert abs(float(fake.norm()) - rec["inference"]["fake_norm"]) <= TOL_OUT * rec["inference"]["fake_norm"]
    # encode_only / demo modes: the reference's style matrix, and its generator output for the explicit
    # style matrix 0.5 * style + 0.1 (the transformation gen_golden.py applied)
    style = orc.encode_only({k: v.clone() for k, v in batch.items()})
    assert close(slice_of(style), rec["inference"]["style_slice"], TOL_OUT)
    assert abs(float(style.norm()) - rec["inference"]["style_norm"]) <= TOL_OUT * rec["inference"]["style_norm"]
    demo = orc.demo({k: v.clone() for k, v in batch.items()}, (style * 0.5 + 0.1).clamp(-1, 1))
    assert close(slice_of(demo), rec["inference"]["demo_fake_slice"], TOL_OUT)
    assert abs(float(demo.norm()) - rec["inference"]["demo_fake_norm"]) <= TOL_OUT * rec["inference"]["demo_fake_norm"]

    random.seed(rec["rng_seed"])
    torch.manual_seed(rec["rng_seed"])
    for it, want in enumerate(rec["iters"]):
        loose = 1 if it == 0 else 300
        gl, fake = orc.run_generator_one_step({k: v.clone() for k, v in batch.items()})
        for k, v in want["g_losses"].items():
            assert abs(float(gl[k].detach()) - v) <= TOL_OUT * loose * abs(v), (k, float(gl[k].detach()), v)
        assert close(slice_of(fake), want["fake_slice"], TOL_OUT * loose)
        grads = {"%s/%s" % (n, k): p.grad for n in ("SR", "E") for k, p in orc.params(n) if p.grad is not None}
        dl = orc.run_discriminator_one_step({k: v.clone() for k, v in batch.items()})
        # recorded after the D step (the branch coins are redrawn there)
        assert want["branch"]["full"] == orc.last_encoded_style_is_full
        assert want["branch"]["noisy"] == orc.last_encoded_style_is_noisy
        for k, v in want["d_losses"].items():
            assert abs(float(dl[k].detach()) - v) <= TOL_DLOSS * loose * abs(v), (k, float(dl[k].detach()), v)
        grads.update({"D/" + k: p.grad for k, p in orc.params("D") if p.grad is not None})
        assert set(grads) == set(want["grad_norms"])
        gmax = max(want["grad_norms"].values())
        for k, n_ref in want["grad_norms"].items():
            tol = (TOL_DG if k.startswith("D/") else TOL_GG) * (1 if it == 0 else 4)
            assert abs(float(grads[k].norm()) - n_ref) <= tol * max(n_ref, 1e-3 * gmax), (k, float(grads[k].norm()), n_ref)
        for k, sl in want["grad_slices"].items():
            tol = (TOL_DG if k.startswith("D/") else TOL_GG) * (1 if it == 0 else 4)
            ref = torch.tensor(sl, dtype=torch.float64)
            got = slice_of(grads[k], 16).double()
            assert float((got - ref).norm()) <= tol * max(float(ref.norm()), 1e-3 * gmax * (len(sl) / grads[k].numel()) ** 0.5) \
                or float((got - ref).norm()) <= tol * want["grad_norms"][k], k
        for k, n_ref in want["state_norms"].items():
            net, key = k.split("/", 1)
            if k in want["zero_grad_keys"]:
                continue
            assert abs(float(orc.S[net][key].detach().float().norm()) - n_ref) <= 1e-3 * max(n_ref, 1e-6), k


def test_layer_kats():
    rec = json.load(open(os.path.join(GOLD, "layer_kats.json")))
    kats = rec["kats"]
    torch.manual_seed(rec["seg_seed"])
    L, C = 19, 8
    seg = O.onehot_labels(torch.randint(0, L, (2, 1, 16, 16)).float(), L)
    style = torch.rand(2, L, 128) * 2 - 1
    kinds = {"spade": ("spade", 256), "sean": ("sean", 8), "sean_fmcap": ("sean", 8), "puresean_fmcap": ("puresean", 8)}
    for nm, (kind, _) in kinds.items():
        want = kats[nm]
        opt = O.make_opt(max_fm_size=8)
        spec = {}
        p = "n"
        spec[p + ".param_free_norm.running_mean"] = (C,)
        spec[p + ".param_free_norm.running_var"] = (C,)
        spec[p + ".param_free_norm.num_batches_tracked"] = ()
        spec[p + ".mlp_shared.0.weight"] = (128, L, 3, 3)
        spec[p + ".mlp_shared.0.bias"] = (128,)
        if kind in ("spade", "sean"):
            for q in ("mlp_gamma", "mlp_beta"):
                spec["%s.%s.weight" % (p, q)] = (C, 128, 3, 3)
                spec["%s.%s.bias" % (p, q)] = (C,)
        if kind in ("sean", "puresean"):
            for q in ("mlp_style_gamma", "mlp_style_beta"):
                spec["%s.%s.weight" % (p, q)] = (C, 128, 3, 3)
                spec["%s.%s.bias" % (p, q)] = (C,)
        if kind == "sean":
            spec[p + ".alpha_beta"] = (1,)
            spec[p + ".alpha_gamma"] = (1,)
        st = {k: O.recipe_tensor("kat_" + nm, k[2:], s, 1.0) for k, s in spec.items()}
        orc = O.Oracle(opt, {"SR": st})
        x = torch.randn(2, C, want["res"], want["res"], generator=torch.Generator().manual_seed(want["x_seed"]))
        y = orc._norm(kind, orc.S["SR"], "n", x, seg, style)
        assert close(slice_of(y, 32), want["y_slice"], 1e-5), nm
        assert close(slice_of(orc.S["SR"]["n.param_free_norm.running_mean"], 8), want["running_mean"], 1e-5)
        assert close(slice_of(orc.S["SR"]["n.param_free_norm.running_var"], 8), want["running_var"], 1e-5)
    p = [[torch.randn(2, 1, 5, 5, generator=torch.Generator().manual_seed(i))] for i in (1, 2)]
    assert abs(float(O.Oracle.hinge(p, True, False)) - kats["hinge"]["g"]) < 1e-6
    assert abs(float(O.Oracle.hinge(p, False, True)) - kats["hinge"]["d_fake"]) < 1e-6
    assert abs(float(O.Oracle.hinge(p, True, True)) - kats["hinge"]["d_real"]) < 1e-6
    feat = torch.randn(2, 6, 8, 8, generator=torch.Generator().manual_seed(3))
    sm = O.style_pool(feat, seg)
    assert close(slice_of(sm, 32), kats["style_pool"]["slice"], 1e-5)
    img = torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(9)) * 2 - 1
    assert close(slice_of(O.bicubic_down(img, 4), 48), kats["bicubic"]["slice"], 1e-6)
    ap = F.avg_pool2d(img, 3, 2, [1, 1], count_include_pad=False)
    assert close(slice_of(ap, 32), kats["avgpool"]["slice"], 1e-6)


# ------------------------------------------------------------------ host-logic rows pinned by gen_golden.host_logic()
HOST = json.load(open(os.path.join(GOLD, "host_logic.json")))


@pytest.mark.parametrize("tag", sorted(HOST["lr_schedule"]))
def test_oracle_lr_schedule_matches_reference(tag):
    """TrainerManager.update_learning_rate (trainer_manager.py:76-96): the reference's per-epoch learning rates of every
    param group (recorded from the real TrainerManager) vs the oracle's restatement."""
    rec = HOST["lr_schedule"][tag]
    over = {k: v for k, v in rec["opt"].items() if k != "lr"}
    opt = O.make_opt(start_size=4, crop_size=32, load_size=32, batchSize=2, ngf=2, nef=4, ndf=4, **over)
    orc = O.Oracle(opt, O.init_state(opt))
    orc.create_optimizers()
    for row in rec["rows"]:
        orc.update_learning_rate(row["epoch"])
        assert [g["lr"] for g in orc.opt_G.param_groups] == pytest.approx(row["G"], rel=1e-12, abs=1e-18)
        assert [g["lr"] for g in orc.opt_D.param_groups] == pytest.approx(row["D"], rel=1e-12, abs=1e-18)
        assert orc.old_lr == pytest.approx(row["old_lr"], rel=1e-12, abs=1e-18)


def test_oracle_syncbn_master_matches_reference():
    """The DataParallel branch of SynchronizedBatchNorm2d (batchnorm.py:128-145) on two replicas: mean, clamp(var, eps)
    inv_std (channel 7 is constant: var = 0 -> the clamp, not `+ eps`, decides) and the running statistics."""
    rec = HOST["syncbn"]
    x = torch.tensor(rec["x"]).reshape(rec["shape"])
    c = x.shape[1]
    mean, inv_std, rm, rv, _ = O.sync_bn_master([x[:2], x[2:]], torch.zeros(c), torch.ones(c))
    assert close(mean, rec["mean"], 1e-6) and close(inv_std, rec["inv_std"], 1e-6)
    assert close(rm, rec["running_mean"], 1e-6) and close(rv, rec["running_var"], 1e-6)
    assert float(inv_std[7]) == pytest.approx(1e-5 ** -0.5, rel=1e-3)          # constant channel: clamp(0, eps)^-1/2
    # clamp vs `+ eps` differ on small nonzero variances: var = eps/2 -> clamp gives eps^-1/2, `+eps` (1.5 eps)^-1/2
    half = torch.full((1, 1, 2, 1), 0.0)
    half[0, 0, 0, 0], half[0, 0, 1, 0] = (0.5e-5) ** 0.5, -((0.5e-5) ** 0.5)
    _, inv, _, _, _ = O.sync_bn_master([half], torch.zeros(1), torch.ones(1))
    assert float(inv[0]) == pytest.approx(1e-5 ** -0.5, rel=1e-4)


@pytest.mark.parametrize("tag", sorted(HOST["init"]))
def test_oracle_init_state_statistics_match_reference(tag):
    """init_weights (base_network.py:28-59): per-tensor std / zero-ness / ranges of a freshly built reference model vs
    the oracle's init_state (a different RNG stream: statistics, not values)."""
    rec = HOST["init"][tag]
    opt = O.make_opt(**rec["opt"])
    st = O.init_state(opt, seed=5)
    check_init_stats({"%s/%s" % (n, k): v for n in ("SR", "D", "E") for k, v in st[n].items()}, rec["stats"])


def check_init_stats(tensors, stats):
    assert set(tensors) == set(stats), set(tensors) ^ set(stats)
    for k, ref in stats.items():
        v = tensors[k].detach().float()
        n = ref["numel"]
        assert v.numel() == n, k
        if ref["norm"] == 0.0:                                     # biases, noise weights, running_mean
            assert float(v.abs().max()) == 0.0, k
        elif k.endswith(("weight_u", "weight_v")):                 # normalised N(0,1) vectors
            assert abs(float(v.norm()) - 1.0) < 1e-5 and abs(ref["norm"] - 1.0) < 1e-5, k
        elif k.endswith("running_var"):
            assert float(v.min()) == 1.0 == float(v.max()), k
        elif k.endswith(("alpha_beta", "alpha_gamma")):            # U(0,1) scalars
            assert 0.0 <= float(v) < 1.0 and 0.0 <= ref["min"] < 1.0, k
        elif n >= 64:                                              # conv weights: N(0, std) with the reference's std
            tol = 6.0 / (2.0 * n) ** 0.5 + 1e-3                    # 6 sigma of the sampling error of a std estimate, x2 sides
            assert abs(float(v.std()) - ref["std"]) <= 2 * tol * ref["std"], (k, float(v.std()), ref["std"])
            assert abs(float(v.mean())) <= 6.0 * ref["std"] / n ** 0.5 + 1e-9, k
        else:
            assert float(v.abs().max()) <= 8 * max(ref["std"], abs(ref["max"]), abs(ref["min"])) + 1e-9, k


# ------------------------------------------------------------------ evaluation metrics pinned by gen_golden.metrics()
METRICS = json.load(open(os.path.join(GOLD, "metrics.json")))["cases"]


@pytest.mark.parametrize("rec", METRICS, ids=lambda r: r["kind"])
def test_oracle_psnr_ssim_rmse_match_reference(rec):
    """MetricsEvaluator.collect_samples' PSNR / SSIM / RMSE (evaluation.py:88-137): the oracle's restatement against the
    numbers the reference's own tensor2im / calculate_psnr / calculate_ssim produced for the same seeded images."""
    n, h, w = rec["shape"]
    fake, real = O.metric_case_inputs(rec["seed"], n, h, w, rec["kind"])
    got = O.psnr_ssim_rmse(fake, real)
    for i in range(n):
        ref_psnr = float(rec["psnr"][i])
        assert float(got[i, 0]) == ref_psnr or abs(float(got[i, 0]) - ref_psnr) <= 1e-12 * abs(ref_psnr)
        assert abs(float(got[i, 1]) - rec["ssim"][i]) <= 1e-10
        assert abs(float(got[i, 2]) - rec["rmse"][i]) <= 1e-6 * rec["rmse"][i] + 1e-12
