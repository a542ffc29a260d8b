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
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


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
