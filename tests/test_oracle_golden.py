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
               if not p.endswith("layer_kats.json"))

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
