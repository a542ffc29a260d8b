"""CPU-only checks: the C-ABI library loads and exports every symbol of include/deepsee_hip.h, the host-side
mirror of the reference interface (state-dict layout, block plan, option presets, conv geometry, packed gamma/beta
row order) and the data-parallel gradient exchange on gloo with world_size 2."""
import ctypes
import os
import re
import sys

import pytest
import torch
import torch.multiprocessing as mp

from oracle import deepsee_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_library_loads_and_exports_header_symbols():
    from deepsee_amd import lib as L
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    so = ctypes.CDLL(L.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "deepsee_hip.h")).read()
    names = sorted(set(re.findall(r"\b(dsee_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing
    so.dsee_version.restype = ctypes.c_int
    assert so.dsee_version() >= 100
    # pure host helpers (no GPU needed)
    assert so.dsee_conv_kpad(3, 3, 512) == 4608 and so.dsee_conv_kpad(4, 4, 24) == 384
    assert so.dsee_conv_wrows(3) == 128 and so.dsee_conv_wrows(512) == 512
    # argument validation returns an error code + message instead of launching
    so.dsee_last_error.restype = ctypes.c_char_p
    rc = so.dsee_conv2d_fwd(None, None, None, None, None, 0, None, 0, ctypes.c_float(0.2), None)
    assert rc == -1 and b"argument check failed" in so.dsee_last_error()


def test_coarse_entry_points_validate_before_they_launch():
    """The coarse entry points (csrc/coarse.cpp) size their caller-owned workspace on the host and refuse bad arguments with an
    error code and a message before any HIP call (so this runs without a GPU)."""
    from deepsee_amd import lib as L
    so = L.lib()
    n1 = so.dsee_sean_norm_fwd_workspace(8, 256, 256, 512, 19, 1)
    n0 = so.dsee_sean_norm_fwd_workspace(8, 256, 256, 512, 19, 0)
    t = 8 * 64 * 64
    # embedding + V2 + U2 dominate: [N,H,W,160] fp32, 36 T 160 two fp16 terms, 36 N 1024 160 two fp16 terms
    assert n1 >= 8 * 256 * 256 * 160 * 4 + 36 * t * 160 * 4 + 36 * 8 * 1024 * 160 * 4 and n1 < 2.2e9
    assert n0 < n1 and n0 >= 8 * 256 * 256 * 128 * 4 + 36 * t * 128 * 4
    nb = so.dsee_spade_resblock_fwd_workspace(8, 256, 256, 512, 19, 1)
    assert nb >= n1 + 2 * 8 * 256 * 256 * 512 * 4 + 2 * 36 * t * 512 * 4 and nb < 9e9
    args = [None] * 24 + [8, 256, 256, 512, None, 0]
    assert len(args) + 1 == len(so.dsee_sean_norm_fwd.argtypes)
    args[1:5] = [256, 256, 0, 19]
    args[13:18] = [1, 1e-5, 0.1, 1.0, 0.2]
    rc = so.dsee_sean_norm_fwd(*args, None)
    assert rc == -1 and b"NULL argument" in so.dsee_last_error()
    rc = so.dsee_spade_resblock_fwd(*([None] * 6), None, 256, 256, 0, 19, None, None, 0, 1, 1e-5, 0.1, 0.2, 8, 256, 256, 512, None,
                                    0, None)
    assert rc == -1 and b"NULL argument" in so.dsee_last_error()
    # round 6: the training pair -- saved area and workspaces sized on the host, NULL / shape validation before any HIP call
    sb = so.dsee_spade_resblock_saved_bytes(8, 64, 64, 512, 19, 1)
    t64 = 8 * 16 * 16
    assert sb >= 3 * 8 * 64 * 64 * 512 * 4 + 2 * 36 * t64 * 512 * 4 + 2 * 36 * t64 * 160 * 4 and sb < 0.7e9
    assert so.dsee_spade_resblock_saved_bytes(8, 64, 64, 512, 19, 0) < sb
    assert so.dsee_spade_resblock_train_fwd_workspace(8, 64, 64, 512, 19, 1) > 36 * t64 * 512 * 4
    assert so.dsee_spade_resblock_bwd_workspace(8, 64, 64, 512, 19, 1, 256, 256, 2) > 36 * t64 * 1024 * 4
    assert len(so.dsee_spade_resblock_train_fwd.argtypes) == 26 and len(so.dsee_spade_resblock_bwd.argtypes) == 26
    rc = so.dsee_spade_resblock_train_fwd(*([None] * 8), 256, 256, 2, 19, None, None, 1e-5, 0.1, 0.2, 8, 64, 64, 512, None, 0, None, 0,
                                          None)
    assert rc == -1 and b"NULL argument" in so.dsee_last_error()
    rc = so.dsee_spade_resblock_bwd(*([None] * 6), 256, 256, 2, 19, *([None] * 6), 0.2, 8, 64, 64, 512, None, 0, None, 0, None)
    assert rc == -1 and b"NULL argument" in so.dsee_last_error()


def test_ctypes_prototypes_come_from_the_header():
    """deepsee_amd.lib binds every entry point with the argtypes / restype parsed from include/deepsee_hip.h: a call with
    the wrong arity, or a float where the header says int, raises instead of corrupting the stack; long / size_t /
    uint64_t arguments are passed at their declared width."""
    import ctypes as C
    from deepsee_amd import lib as L
    protos = L.header_prototypes()
    so = L.lib()
    assert len(protos) >= 75 and all(getattr(so, n).argtypes == a for n, (_, a) in protos.items())
    assert protos["dsee_gemm_bf16x3_af32"][1][3] is C.c_long and protos["dsee_rng_fill"][1][2] is C.c_uint64
    assert protos["dsee_conv2d_wgrad_workspace"][0] is C.c_size_t and protos["dsee_last_error"][0] is C.c_char_p
    # a long that does not fit an int: truncated to 32 bits it would be M = 100 -> 2 partial rows instead of 1024
    assert so.dsee_channel_dot_workspace(100, 512) == 2 * 512 * 4
    assert so.dsee_channel_dot_workspace((1 << 33) + 100, 512) == 1024 * 512 * 4
    with pytest.raises(TypeError):
        so.dsee_conv_kpad(3, 3)
    with pytest.raises(C.ArgumentError):
        so.dsee_conv_kpad(3, 3, 1.5)
    if not torch.cuda.is_available():
        return
    with pytest.raises(L.DseeError):
        L.call("act_fwd", None, None, 4)                                          # arity checked before the call


def test_product_refuses_to_run_without_gpu():
    from deepsee_amd.options import make_opt
    from deepsee_amd.sr_model import SRModel
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        SRModel(make_opt("independent_8x_32"))


@pytest.mark.parametrize("over", [
    dict(), dict(start_size=4, crop_size=32, load_size=32), dict(start_size=16, crop_size=512, load_size=512, add_noise=False),
    dict(netE="fullstyle", noisy_style_scale=0.05, guiding_style_image=True), dict(ngf=8, add_noise=False)])
def test_state_dict_layout_matches_reference_spec(over):
    """Keys/shapes of the build's modules == the oracle's spec, which gen_golden.py asserts equal to the real
    reference's state_dict(): checkpoints interchange."""
    from deepsee_amd import networks as N
    from deepsee_amd.options import make_opt
    from deepsee_amd.sr_model import block_plan
    opt = make_opt(**over)
    oopt = O.make_opt(**over)
    assert block_plan(opt) == O.block_plan(oopt)
    spec = O.net_specs(oopt)
    nets = {"SR": N.DeepSEESR(opt, block_plan(opt)), "D": N.MultiscaleDiscriminator(opt), "E": N.StyleEncoder(opt),
            "VGG": N.VGG19Taps()}
    for label, net in nets.items():
        got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        want = {k: tuple(s) for k, s in spec[label].items()}
        assert got == want, (label, set(got) ^ set(want))
    if "full" not in opt.netE:
        n_main = sum(1 for k, _ in nets["SR"].named_parameters()) + sum(
            1 for k, _ in nets["E"].named_parameters() if "mini" not in k)
        n_low = sum(1 for k, _ in nets["E"].named_parameters() if "mini" in k)
        orc = O.Oracle(oopt, O.init_state(oopt))
        orc.create_optimizers()
        assert [len(g["params"]) for g in orc.opt_G.param_groups] == [n_main, n_low]


def test_option_presets():
    from deepsee_amd.options import make_opt
    o = make_opt("guided_8x_256")
    assert o.netE == "fullstyle" and o.guiding_style_image and o.noisy_style_scale == 0.05 and o.start_size == 32
    o = make_opt("independent_32x_512")
    assert o.start_size == 16 and o.crop_size == 512 and not o.add_noise and o.load_size == 512
    assert make_opt().lr == 2e-4 and make_opt().beta1 == 0.0 and make_opt().beta2 == 0.9


def test_conv_geometry_and_packing_helpers():
    from deepsee_amd import lib as L, ops
    g = L.geom_fwd(2, 17, 17, 24, 32, 4, 2, 2)
    assert (g.Ho, g.Wo) == (9, 9) and (g.mul, g.off, g.kdir, g.dshift) == (2, -2, 1, 0)
    d = L.geom_dgrad(g)
    assert (d.Hi, d.Ho, d.Cin, d.Cout) == (9, 17, 32, 24) and (d.mul, d.off, d.kdir, d.dshift) == (1, 2, -1, 1)
    g = L.geom_fwd(1, 8, 8, 128, 256, 3, 1, 1, ups=1)
    assert (g.Ho, g.Wo) == (16, 16)
    for c in (8, 64, 96, 512):
        idx, rows = ops.packed_perm(c, "cpu")
        assert rows == (c + 63) // 64 * 128
        real = idx[idx < 2 * c]
        assert sorted(real.tolist()) == list(range(2 * c))          # every gamma/beta row appears exactly once
        for p in range(rows):                                        # gamma at p, beta of the same channel at p+32
            if (p % 64) < 32 and idx[p] < 2 * c:
                assert idx[p + 32] == idx[p] + c


GOLD = os.path.join(ROOT, "tests", "golden")


def test_init_weights_statistics_match_reference():
    """deepsee_amd.sr_model.init_weights on the build's parameter holders vs the per-tensor statistics of a
    freshly initialised REFERENCE model (tests/golden/host_logic.json, written by oracle/gen_golden.py): xavier /
    kaiming std of every conv weight incl. spectral-norm weight_orig, biases / noise weights / running_mean exactly 0,
    running_var 1, alpha in [0,1), ||u|| = ||v|| = 1."""
    import json
    from deepsee_amd import networks as N
    from deepsee_amd.options import make_opt
    from deepsee_amd.sr_model import block_plan, init_weights
    from tests.test_oracle_golden import check_init_stats
    host = json.load(open(os.path.join(GOLD, "host_logic.json")))
    for tag, rec in host["init"].items():
        opt = make_opt(**rec["opt"])
        gen = torch.Generator().manual_seed(17)
        nets = {"SR": N.DeepSEESR(opt, block_plan(opt)), "D": N.MultiscaleDiscriminator(opt), "E": N.StyleEncoder(opt)}
        for net in nets.values():
            init_weights(net, opt.init_type, opt.init_variance, gen)
        check_init_stats({"%s/%s" % (n, k): v for n, net in nets.items() for k, v in net.state_dict().items()},
                         rec["stats"])


def test_update_learning_rate_matches_reference_schedule():
    """TrainerManager.update_learning_rate against the learning rates the REAL reference's TrainerManager set
    on every param group at every epoch (tests/golden/host_logic.json; trainer_manager.py:76-96)."""
    import json
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    host = json.load(open(os.path.join(GOLD, "host_logic.json")))

    class Opt:   # optimizer stand-in with the surface update_learning_rate touches
        def __init__(self, lrs):
            self.param_groups = [{"lr": lr} for lr in lrs]

    for tag, rec in host["lr_schedule"].items():
        over = {k: v for k, v in rec["opt"].items() if k != "lr"}
        opt = make_opt(**over)
        assert opt.lr == rec["opt"]["lr"]
        tm = object.__new__(TrainerManager)
        tm.opt, tm.old_lr = opt, opt.lr
        lr_g, lr_d = (opt.lr, opt.lr) if opt.no_TTUR else (opt.lr / 2, opt.lr * 2)
        tm.optimizer_G, tm.optimizer_D = Opt([lr_g, lr_g / 4]), Opt([lr_d])
        for row in rec["rows"]:
            tm.update_learning_rate(row["epoch"])
            assert [g["lr"] for g in tm.optimizer_G.param_groups] == pytest.approx(row["G"], rel=1e-12, abs=1e-18)
            assert [g["lr"] for g in tm.optimizer_D.param_groups] == pytest.approx(row["D"], rel=1e-12, abs=1e-18)
            assert tm.old_lr == pytest.approx(row["old_lr"], rel=1e-12, abs=1e-18)


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import random
    import torch.distributed as dist
    from deepsee_amd import parallel
    r, _, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(10_007 * 4, generator=g)
    mine = flat.clone()
    hook = parallel.GradAllReduce(world, chunk_mb=0.05)   # ~13k floats per chunk -> several chunks
    scale = hook(flat)
    other = torch.randn(10_007 * 4, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    ok = abs(scale - 0.5) < 1e-12 and torch.allclose(flat, mine + other, atol=1e-6)
    random.seed(1234)                                      # same branch coins on every rank
    coins = [random.random() for _ in range(4)]
    t = torch.tensor(coins, dtype=torch.float64)
    dist.all_reduce(t)
    ok = ok and torch.allclose(t / world, torch.tensor(coins, dtype=torch.float64))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_grad_allreduce_gloo_world2():
    from deepsee_amd import parallel
    assert parallel.chunk_bounds(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert parallel.chunk_bounds(8, 3)[0] == (0, 4)        # chunk sizes stay 16-byte aligned
    assert _spawn(_dp_worker) == [(0, True), (1, True)]


def _spawn(fn, world=2, timeout=180):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() * 7 + hash(fn.__name__)) % 2000)
    procs = [ctx.Process(target=fn, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    return sorted(res)


def _init_gloo(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    from deepsee_amd import parallel
    r, _, w = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    return parallel


def _flat_adam_worker(rank, world, port, q):
    """Drives the REAL FlatAdam.step host logic (descriptor upload, `touched` MAX-reduce, block-aligned chunking,
    per-chunk wait -> update, step bump, lr change) under gloo; only the Adam kernel launch itself is replaced by a torch
    emulation reading the same device descriptors (the HIP kernel cannot run without a GPU)."""
    import numpy as np
    import torch.distributed as dist
    parallel = _init_gloo(rank, world, port)
    from deepsee_amd.optim import ADAM_DT, BLOCK, FlatAdam

    class CpuAdam(FlatAdam):
        launches = 0

        def _gather(self):      # torch emulation of dsee_grad_gather: flat gradient (zeros without a gradient) + active flags
            self.grad.zero_()
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                if p.grad is not None:
                    self.grad[o:o + p.numel()].copy_(p.grad.reshape(-1))
                self._active_dev[i] = 0 if p.grad is None else 1
                self.grad[i] = 0.0 if p.grad is None else 1.0      # header flag (rides in the first all-reduced chunk)

        def _launch(self, b0, b1, grad_scale, clip):
            CpuAdam.launches += 1
            desc = np.frombuffer(self.desc_dev.numpy().tobytes(), dtype=ADAM_DT)
            bt = self.block_tensor.numpy()
            b1_, b2_ = self.betas
            for blk in range(b0, b1):
                d = desc[bt[blk]]
                if not d["active"]:
                    continue
                lo = int(d["offset"]) + (blk - int(d["first_block"])) * BLOCK
                hi = min(int(d["offset"] + d["numel"]), lo + BLOCK)
                step = int(d["step"]) + 1
                g = self.grad[lo:hi] * grad_scale
                if clip > 0:
                    g = g.clamp(-clip, clip)
                self.exp_avg[lo:hi].mul_(b1_).add_(g, alpha=1 - b1_)
                self.exp_avg_sq[lo:hi].mul_(b2_).addcmul_(g, g, value=1 - b2_)
                denom = self.exp_avg_sq[lo:hi].sqrt() / (1 - b2_ ** step) ** 0.5 + self.eps
                self.flat[lo:hi].sub_(float(d["lr"]) / (1 - b1_ ** step) * self.exp_avg[lo:hi] / denom)

    sizes = [(3000,), (5,), (32, 32), (2500,), (7,), (1100,)]
    gen = torch.Generator().manual_seed(5)
    init = [torch.randn(s, generator=gen) for s in sizes]
    params = [torch.nn.Parameter(t.clone()) for t in init]
    named = [("p%d" % i, p) for i, p in enumerate(params)]
    opt = CpuAdam([{"params": named[:4], "lr": 1e-2}, {"params": named[4:], "lr": 2.5e-3}], betas=(0.5, 0.9))
    opt.reduce_hook = parallel.GradAllReduce(world, chunk_mb=2000 * 4 / (1 << 20))       # ~2000 floats per chunk
    ranges = opt.chunk_ranges(opt.reduce_hook.chunk_elems)
    ok = len(ranges) >= 3 and ranges[0][2] == 0 and ranges[-1][3] == opt.total
    ok = ok and all(a[3] == b[2] and a[1] == b[0] for a, b in zip(ranges, ranges[1:]))      # tile blocks and elements
    # reference: plain torch Adam on the rank-averaged gradients, every tensor that ANY rank touched is active
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    ropt = torch.optim.Adam([{"params": ref[:4], "lr": 1e-2}, {"params": ref[4:], "lr": 2.5e-3}], betas=(0.5, 0.9))
    for it in range(3):
        coef = [[torch.randn(s, generator=torch.Generator().manual_seed(1000 * it + 10 * r + i))
                 for i, s in enumerate(sizes)] for r in range(world)]
        # tensor 3 gets a gradient on rank 0 only, tensor 4 on no rank in iteration 1 (skipped: its step must not advance)
        def touched(r, i):
            return not (i == 3 and r == 1) and not (i == 4 and it == 1)
        opt.zero_grad()
        sum((p * coef[rank][i]).sum() for i, p in enumerate(params) if touched(rank, i)).backward()
        if it == 2:
            opt.param_groups[0]["lr"] = 5e-3                                                   # update_learning_rate
            ropt.param_groups[0]["lr"] = 5e-3
        opt.step()
        ropt.zero_grad()
        for i, p in enumerate(ref):
            gs = [coef[r][i] for r in range(world) if touched(r, i)]
            p.grad = sum(gs) / world if gs else None
        ropt.step()
    err = max(float((p.detach() - r.detach()).abs().max()) for p, r in zip(params, ref))
    steps = opt.steps().tolist()
    ok = ok and err < 2e-6 and steps == [3, 3, 3, 3, 2, 3] and CpuAdam.launches == 3 * len(ranges)
    # every rank ends with the same parameters
    t = opt.flat.clone()
    dist.all_reduce(t)
    ok = ok and torch.allclose(t / world, opt.flat, atol=0, rtol=0)
    q.put((rank, bool(ok), err, steps))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_adam_chunked_allreduce_gloo_world2():
    res = _spawn(_flat_adam_worker)
    assert [r[:2] for r in res] == [(0, True), (1, True)], res


def _syncbn_worker(rank, world, port, q):
    """SyncBN-over-RCCL protocol under gloo: the product's exchange functions (parallel.gather_stats,
    parallel.allreduce_sums) around torch emulations of the three kernels (dsee_norm_stats_local, dsee_norm_stats_merge,
    the dsee_modulate_bwd reduce/apply halves), against the oracle's restatement of the reference's DataParallel branch
    on the concatenated batch (pinned to the reference by tests/golden/host_logic.json)."""
    import torch.distributed as dist
    parallel = _init_gloo(rank, world, port)
    from oracle import deepsee_oracle as O
    c, eps, mom = 8, 1e-5, 0.1
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(4, c, 6, 6, generator=g) * 3 + 1).double()
    x[:, 7] = 0.25
    shards = [x[:2], x[2:]]
    mine = shards[rank]
    # dsee_norm_stats_local: (mean, M2) of this rank's shard
    flat = mine.transpose(0, 1).reshape(c, -1)
    local = torch.stack([flat.mean(1), ((flat - flat.mean(1, keepdim=True)) ** 2).sum(1)])
    rows = parallel.gather_stats(local, world)
    ok = tuple(rows.shape) == (world, 2, c)
    # dsee_norm_stats_merge: Chan's update in rank order, clamp(var, eps)
    n = mean = m2 = 0.0
    cnt = flat.shape[1]
    for r in range(world):
        d = rows[r, 0] - mean
        nt = n + cnt
        mean = mean + d * cnt / nt
        m2 = m2 + rows[r, 1] + d * d * n * cnt / nt
        n = nt
    var = m2 / n
    inv_std = var.clamp(min=eps) ** -0.5
    rm, rv = mom * mean, (1 - mom) * 1.0 + mom * var * n / (n - 1)
    xs = [s.clone().requires_grad_(True) for s in shards]
    omean, oinv, orm, orv, outs = O.sync_bn_master(xs, torch.zeros(c).double(), torch.ones(c).double(), eps, mom)
    for a, b in ((mean, omean), (inv_std, oinv), (rm, orm), (rv, orv)):
        ok = ok and torch.allclose(a, b.detach(), rtol=1e-9, atol=1e-12)
    # backward: sum d, sum d*xhat over the GLOBAL batch (allreduce_sums), dx = inv_std (d - S0/M - xhat S1/M)
    R = [torch.randn(s.shape, generator=torch.Generator().manual_seed(9 + i)).double() for i, s in enumerate(shards)]
    sum((o * r).sum() for o, r in zip(outs, R)).backward()
    xhat = (mine - mean[None, :, None, None]) * inv_std[None, :, None, None]
    d = R[rank]
    sums = torch.stack([d.transpose(0, 1).reshape(c, -1).sum(1), (d * xhat).transpose(0, 1).reshape(c, -1).sum(1)])
    parallel.allreduce_sums(sums, world)
    m_tot = world * cnt
    dx = inv_std[None, :, None, None] * (d - sums[0][None, :, None, None] / m_tot
                                         - xhat * sums[1][None, :, None, None] / m_tot)
    live = [ch for ch in range(c) if ch != 7]          # the clamped channel has no variance gradient in the reference
    ok = ok and torch.allclose(dx[:, live], xs[rank].grad[:, live], rtol=1e-8, atol=1e-10)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_syncbn_exchange_protocol_gloo_world2():
    assert _spawn(_syncbn_worker) == [(0, True), (1, True)]


def test_device_input_pipeline_host_side(tmp_path):
    """SURVEY 8 f3, host half: datasets yield the uint8 wire format, FolderDataset pairs files by stem and applies the
    reference's resize + crop (base_dataset.py:87-116,171-201), DeviceLoader shards an epoch disjointly over the ranks;
    the oracle's restatement of ToTensor + Normalize + flip is what the HIP kernels are checked against on the GPU."""
    import numpy as np
    from PIL import Image
    from deepsee_amd import data as D
    from deepsee_amd.options import make_opt
    opt = make_opt(start_size=4, crop_size=32, load_size=40, batchSize=2)
    ds = D.SyntheticDataset(opt, length=10, seed=3)
    s = ds[4]
    assert s["label"].dtype == np.uint8 and s["label"].shape == (32, 32) and int(s["label"].max()) < opt.label_nc
    assert s["image"].dtype == np.uint8 and s["image"].shape == (32, 32, 3) and s["flip"] in (0, 1)
    assert np.array_equal(ds[4]["image"], s["image"])                      # deterministic per index
    for d_ in ("lab", "img"):
        os.makedirs(str(tmp_path / d_))
    rng = np.random.default_rng(0)
    for k in range(3):
        lab = rng.integers(0, 19, size=(48, 48), dtype=np.uint8)
        lab[0, 0] = 255
        Image.fromarray(lab).save(str(tmp_path / "lab" / ("%d.png" % k)))
        Image.fromarray(rng.integers(0, 256, size=(48, 48, 3), dtype=np.uint8)).save(str(tmp_path / "img" / ("%d.png" % k)))
    fd = D.FolderDataset(opt, str(tmp_path / "lab"), str(tmp_path / "img"), seed=1)
    assert len(fd) == 3
    it = fd[1]
    assert it["label"].shape == (32, 32) and it["image"].shape == (32, 32, 3) and it["path"].endswith("1.png")
    # the crop is the same window of the NEAREST-resized label and the BICUBIC-resized image
    full = np.asarray(Image.open(str(tmp_path / "lab" / "1.png")).resize((40, 40), Image.NEAREST))
    assert any(np.array_equal(it["label"], full[y:y + 32, x:x + 32]) for y in range(9) for x in range(9))
    # options: opt.no_flip is honoured (an explicit argument overrides it), unsupported preprocess modes and multi-channel
    # label files are refused instead of silently mis-handled
    import pytest
    assert all(D.FolderDataset(make_opt(start_size=4, crop_size=32, load_size=40, no_flip=True), str(tmp_path / "lab"),
                               str(tmp_path / "img"), seed=s_)[0]["flip"] == 0 for s_ in range(6))
    assert any(D.FolderDataset(opt, str(tmp_path / "lab"), str(tmp_path / "img"), seed=s_)[0]["flip"] == 1 for s_ in range(6))
    with pytest.raises(ValueError):
        D.FolderDataset(make_opt(start_size=4, crop_size=32, load_size=40, preprocess_mode="scale_width"),
                        str(tmp_path / "lab"), str(tmp_path / "img"))
    with pytest.raises(ValueError):
        D.FolderDataset(opt, str(tmp_path / "img"), str(tmp_path / "img"))[0]     # RGB files as label maps
    # sharding: two ranks, disjoint and equally long
    l0 = D.DeviceLoader(ds, opt, shard=(0, 2), seed=5)
    l1 = D.DeviceLoader(ds, opt, shard=(1, 2), seed=5)
    i0, i1 = l0.indices(), l1.indices()
    assert len(i0) == len(i1) == 5 and not set(i0) & set(i1) and len(l0) == 2
    b = l0.collate([ds[i] for i in i0[:2]])
    assert b["label"].dtype == torch.uint8 and tuple(b["image"].shape) == (2, 32, 32, 3) and b["flip"].dtype == torch.uint8
    # oracle restatement of the device half (the GPU test compares the HIP kernels with exactly this)
    img, lab = O.device_pipeline_reference(b["image"], b["label"], b["flip"], opt.label_nc)
    assert tuple(img.shape) == (2, 3, 32, 32) and float(img.min()) >= -1.0 and float(img.max()) <= 1.0
    k = int(torch.nonzero(b["flip"])[0]) if int(b["flip"].sum()) else None
    if k is not None:
        assert torch.equal(lab[k, 0], b["label"][k].flip(-1).float())


def test_bf16x3_split_arithmetic_emulated_on_cpu():
    """The numerical argument behind deepsee_amd/csrc/gemm_bf16x3.hip, emulated with torch on the CPU: an fp32 value is
    EXACTLY the sum of three bf16 terms, and the six partial products of weight above 2^-26 accumulated in fp32 give a
    GEMM error (vs float64) no larger than a plain fp32 GEMM's; with only three products it is ~16x larger."""
    import torch
    g = torch.Generator().manual_seed(0)
    a = torch.randn(256, 512, generator=g) * torch.rand(256, 1, generator=g).exp()
    b = torch.randn(512, 192, generator=g)

    def split(x):
        x0 = x.bfloat16().float()
        r = x - x0
        x1 = r.bfloat16().float()
        r = r - x1
        x2 = r.bfloat16().float()
        assert torch.equal(x0 + x1 + x2, x) and float((r - x2).abs().max()) == 0.0
        return x0, x1, x2

    a0, a1, a2 = split(a)
    b0, b1, b2 = split(b)
    ref = a.double() @ b.double()
    err = lambda y: float((y.double() - ref).norm() / ref.norm())
    acc = torch.zeros(256, 192)
    for k in range(0, 512, 16):                      # one MFMA = 16 k's; smallest terms first, one fp32 accumulator
        s = slice(k, k + 16)
        for x, y in ((a2, b0), (a1, b1), (a0, b2), (a1, b0), (a0, b1), (a0, b0)):
            acc = acc + x[:, s] @ y[s]
    e6, e32 = err(acc), err(a @ b)
    e3 = err(a0 @ b0 + (a0 @ b1 + a1 @ b0))
    assert e6 <= 1.5 * e32 and e6 < 5e-7, (e6, e32)
    assert e3 > 8 * e6                               # the three small products are what makes it fp32-accurate


def test_f16x2_split_arithmetic_emulated_on_cpu():
    """The numerical argument behind the fp16x2 form of deepsee_amd/csrc/gemm_bf16x3.hip, emulated with torch on the
    CPU: after scaling by a power of two that brings the largest element into [2^13, 2^14), an fp32 value differs from
    the sum of two fp16 terms (2 x 11 significand bits) by at most 2^-22 of its magnitude -- rms 2^-24, the size of one
    fp32 rounding -- or 2^-25 absolute where the second term goes subnormal, and the three products a1*b0 + a0*b1 + a0*b0
    accumulated in fp32 give a GEMM error (vs float64) no larger than a plain fp32 GEMM's (the accumulator's rounding
    dominates both) -- also on data spread over e^(+-8)."""
    import math
    g = torch.Generator().manual_seed(0)

    def split(x):
        amax = float(x.abs().max())
        s = 2.0 ** (13 - math.floor(math.log2(amax)))
        xs = x * s
        assert 2.0 ** 13 <= float(xs.abs().max()) < 2.0 ** 14
        h0 = xs.half().float()
        h1 = (xs - h0).half().float()
        err = (xs - h0 - h1).abs()
        assert bool((err <= torch.maximum(xs.abs() * 2.0 ** -22, torch.tensor(2.0 ** -25))).all())
        big = xs.abs() > 1.0
        assert float(((err[big] / xs.abs()[big]) ** 2).mean().sqrt()) < 2.0 ** -23.5
        return h0, h1, s

    for spread in (0.0, 2.0):
        a = torch.randn(256, 512, generator=g) * (torch.randn(256, 512, generator=g) * spread).exp()
        b = torch.randn(512, 192, generator=g) * 0.02
        a0, a1, sa = split(a)
        b0, b1, sb = split(b)
        ref = a.double() @ b.double()
        err = lambda y: float((y.double() - ref).norm() / ref.norm())
        acc = torch.zeros(256, 192)
        for k in range(0, 512, 16):                  # one MFMA = 16 k's; smallest terms first, one fp32 accumulator
            sl = slice(k, k + 16)
            for x, y in ((a1, b0), (a0, b1), (a0, b0)):
                acc = acc + x[:, sl] @ y[sl]
        e3, e32 = err(acc / (sa * sb)), err(a @ b)
        e1 = err((a0 @ b0) / (sa * sb))
        assert e3 <= 1.5 * e32 and e3 < 5e-7, (spread, e3, e32)
        assert e1 > 100 * e3                          # one product alone is fp16 accuracy: the cross terms carry it


def test_style_matrix_csv_roundtrip(tmp_path):
    """deepsee_amd.util.save_style_matrix writes what the reference's util/util.py:150-158 writes (numpy.savetxt with
    ',' delimiter) and load_style_matrix reads it back bit-exactly in fp32."""
    import numpy as np
    import torch
    from deepsee_amd.util import load_style_matrix, save_style_matrix
    m = (torch.rand(19, 128, generator=torch.Generator().manual_seed(3)) * 2 - 1)
    p = str(tmp_path / "sub" / "style.csv")
    save_style_matrix(m, p, create_dir=True)
    want = str(tmp_path / "ref.csv")
    np.savetxt(want, np.array(m), delimiter=",")
    assert open(p).read() == open(want).read()
    assert torch.equal(load_style_matrix(p, device="cpu"), m)
    import pytest
    with pytest.raises(AssertionError):
        save_style_matrix(m[None], p)


def test_rank_core_sets_partition_the_allowed_cores():
    """deepsee_amd.parallel.rank_core_set: the ranks of a node get disjoint, contiguous, equally sized core sets that
    cover the allowed cores (8 launch threads on one host must not share cores); more ranks than cores degrade to sharing."""
    from deepsee_amd import parallel
    cores = list(range(4, 36))
    sets = [parallel.rank_core_set(r, 8, cores) for r in range(8)]
    assert all(len(s) == 4 for s in sets)
    assert sorted(c for s in sets for c in s) == cores
    assert all(s == list(range(s[0], s[0] + 4)) for s in sets)
    assert parallel.rank_core_set(5, 8, [0, 1]) == [0, 1]          # fewer cores than ranks: everybody keeps what there is
    assert parallel.pin_rank_cores(0, 1) is None                   # single rank: untouched


def test_presplit_path_selection_rules():
    """Host-side rules that decide when a Winograd layer takes round 3's pre-split operand kernels (deepsee_amd/ops.py): the
    input's maximum must have been written by its producer, the GEMM must be big enough for 256 x 256 tiles, and -- when the
    weight gradient will read the same V2 -- the reduction width must be one the transpose-read TN kernel takes."""
    import torch
    from deepsee_amd import ops
    x = torch.zeros(1)
    assert not ops._presplit_ok(x, 32768, 32768, 512)                       # no maximum known in advance
    x.dsee_amax = torch.zeros(2048)
    assert ops._presplit_ok(x, 32768, 32768, 512)                           # 512 -> 512 @256^2, bs = 8
    assert ops._presplit_ok(x, 2048, 2048, 512)                             # @64^2: 288 x 2 tiles of 256 x 256
    assert not ops._presplit_ok(x, 512, 512, 512)                           # @32^2: too few tiles, 128 x 128 kernel
    assert not ops._presplit_ok(x, 32768, 32768, 384)                       # output width not a multiple of 256
    assert not ops._presplit_ok(x, 32768, 4096 + 64, 512)                   # group size not a multiple of 256
    assert ops._presplit_ok(x, 32768, 32768, 512, k_s=160, keep=True) and ops._presplit_ok(x, 32768, 32768, 512, 256, True)
    assert not ops._presplit_ok(x, 32768, 32768, 512, k_s=96, keep=True)    # TN kernel: 160 or a multiple of 128 columns
    with ops.KernelPlan(presplit_a=False).active():
        assert not ops._presplit_ok(x, 32768, 32768, 512)
    assert ops.P() is ops.DEFAULT_PLAN and ops._presplit_ok(x, 32768, 32768, 512)
    # the channel sums that ride in the A dY A^T pass / the statistics rows of a producer need 256 % (C/4) == 0
    assert ops._dout_sums_ok(8, 8, 512, 2) and not ops._dout_sums_ok(8, 4, 512, 2) and not ops._dout_sums_ok(8, 8, 512, 1)
    assert ops._stats_rows_ok(512) and ops._stats_rows_ok(64) and not ops._stats_rows_ok(96 * 4 + 4)


def test_carried_operand_bound_is_dropped_after_an_in_place_update():
    """a gradient tensor carries max |g| (written by its producer) to the kernel that pre-splits A g A^T with it.
    The autograd engine sums fan-in gradients IN PLACE into one addend: the sum keeps that addend's Python attributes but may
    be up to 2x larger.  The bound is stored with the tensor version it was measured at and ignored once the version moved."""
    import torch
    from deepsee_amd import ops
    g = torch.ones(4)
    slot = torch.zeros(2048)
    ops.tag_amax(g, slot)
    assert ops.carried_amax(g) is slot
    assert ops.carried_amax(g.contiguous()) is slot          # (the same object)
    g.add_(torch.ones(4))                                    # what InputBuffer::add does to the first gradient
    assert ops.carried_amax(g) is None
    assert not ops._presplit_ok(g, 32768, 32768, 512)
    w = torch.ones(4)
    w.dsee_amax = slot                                       # weights: tagged without a version, always trusted
    assert ops.carried_amax(w) is slot
    # a fan-in in a real graph: y = f(x) + h(x) where both branches return tagged gradients
    class Tagged(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2
        @staticmethod
        def backward(ctx, dy):
            dx = dy * 2
            ops.tag_amax(dx, slot)
            return dx
    seen = []
    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()
        @staticmethod
        def backward(ctx, dy):
            seen.append(ops.carried_amax(dy))
            return dy
    x = torch.ones(4, requires_grad=True)
    p = Probe.apply(x)
    (Tagged.apply(p) + Tagged.apply(p)).sum().backward()
    assert seen == [None]                                    # the engine's sum of two tagged gradients carries no bound
    seen.clear()
    x = torch.ones(4, requires_grad=True)
    Tagged.apply(Probe.apply(x)).sum().backward()
    assert seen == [slot] or seen[0] is slot                 # single consumer: the bound arrives


def test_kernel_plan_is_per_model_and_per_thread():
    """deepsee_amd/plan.py: a plan is an immutable value; activating one is thread-local and nests; an autograd node's
    backward runs under the plan its forward recorded even when another plan is active on the calling thread."""
    import threading
    import torch
    from deepsee_amd import ops, plan as PL
    from deepsee_amd.options import make_opt
    assert PL.current() is PL.DEFAULT_PLAN and not PL.DEFAULT_PLAN.half
    a = PL.from_opt(make_opt(precision="fp16"))
    b = PL.from_opt(make_opt(kernel_plan=dict(gemm_f16x2=False)))
    assert a.half and a.gemm_f16x2 and not b.half and not b.gemm_f16x2
    with pytest.raises(Exception):
        a.half = False                                 # frozen
    with pytest.raises(ValueError):
        PL.from_opt(make_opt(precision="bf16"))
    with pytest.raises(TypeError):
        PL.from_opt(make_opt(kernel_plan=dict(no_such_switch=1)))
    with a.active():
        assert ops.P() is a and ops._split_kind(512, 512) == 3
        with b.active():
            assert ops.P() is b and ops._split_kind(512, 512) == 1
        assert ops.P() is a
        seen = []
        t = threading.Thread(target=lambda: seen.append(PL.current()))
        t.start(); t.join()
        assert seen == [PL.DEFAULT_PLAN]               # another thread is not affected
    assert ops.P() is PL.DEFAULT_PLAN and ops._split_kind(512, 512) == 2

    class Node(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            ctx.plan = ops.P()
            return x * 2

        @staticmethod
        @ops._under_plan
        def backward(ctx, dy):
            seen.append(ops.P())
            return dy * 2
    seen.clear()
    x = torch.ones(2, requires_grad=True)
    with a.active():
        y = Node.apply(x)
    with b.active():
        y.sum().backward()
    assert seen == [a]
