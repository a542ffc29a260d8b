#!/usr/bin/env python
"""Benchmark of the DeepSEE G+D training step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = TrainerManager.run_generator_one_step + run_discriminator_one_step (train.py:40-44) on one synthetic
batch that is already resident in HBM.  Default workload: independent 8x 32->256, 19-class blocky masks, bs=8 per
GPU, fp32 (BASELINE.json configs[1]; weak scaling: the per-GPU batch is fixed as N grows).  `--config` selects the
other BASELINE configurations (guided_8x_256 = configs[3], independent_32x_512 = configs[4]) for separate lines.
`python bench.py --gpus N` without a launcher starts the N ranks itself.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     — the dominant kernel (by summed time in a step), measured live with HIP events on the launch
                 stream in one extra instrumented step after the timed region.  Both of its rooflines are computed:
                 matrix cores (algorithmic fp32 FLOPs = 2*M*N*K of every launch / summed duration, against the peak of
                 its arithmetic: 157.3 TFLOP/s for v_mfma_f32; dense bf16/fp16 MFMA peak / 6 for bf16x3, / 3 for fp16x2,
                 deepsee_amd/csrc/gemm_bf16x3.hip) and HBM (algorithmic bytes / summed duration against 8 TB/s); `bound`
                 / `frac` are the one it sits closer to.  `traffic` = HBM bytes per launch from rocprofv3 PMC passes
                 (profiles/r02_pmc_traffic.json, provenance in `traffic_source`).
  spade_fused  — the fused SPADE/SEAN kernel north_star's 70 % target names (round 3: gamma/beta Winograd GEMM with the
                 output transform folded in registers + BN-normalise + modulate + LeakyReLU, dsee_spade_fused_fwd): its
                 algorithmic HBM bytes / its time against 8 TB/s, and the operand bytes it pulls through the L2 -> LDS
                 path (the resource that bounds it) against the 34.5 TB/s the guide measured for L2.
  norm_forward — the WHOLE normalisation forward at the top resolution (statistics, embedding, transforms, fused kernel)
                 on SURVEY 8(d)'s algorithmic bytes (3.49 GB at N = 8, 256^2) against 8 TB/s.
  host_enqueue_ms_per_step — wall time the Python thread spends inside step() (launch enqueue; no device sync inside).
  f32_mfma_only / bf16x3_exact — the same step with every GEMM kept on v_mfma_f32_32x32x2_f32 / on the exact 3-term
                 bf16 split (6 MFMA products), 3 steps each, rank 0 / N=1.
  cpu_baseline — the oracle (CPU restatement of the reference path, oracle/deepsee_oracle.py) timed on this box's
                 host cores: one G+D iteration at bs=1 of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Round 6 (profiles/r06_mfma_power_cap.txt, r06_gemm_ablation.md): what the 1 400 W socket cap leaves of the dense 16-bit matrix
# peak when the operands toggle -- v_mfma_f32_32x32x16_f16 back to back on random operands RESIDENT IN REGISTERS (no LDS, L2 or
# HBM traffic): 1 646 TFLOP/s at ~1.67 GHz (all-zero operands: 2 460 at 2.395 GHz).  The Winograd-domain GEMMs run on this limit
# (socket at 1 378-1 403 W, shader clock 1.54-1.64 GHz), so their roofline carries it next to the datasheet peak.
F16_MFMA_POWER_CAP_TFLOPS = 1646.5
FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md
F16_MFMA_PEAK_TFLOPS = 2516.6   # dense bf16 / fp16 MFMA (same guide)
HBM_PEAK_GBPS = 8000.0          # HBM3E spec (same guide; ~6.3 TB/s achievable by a float4 copy)
L2_PEAK_GBPS = 34500.0          # aggregate L2 -> CU bandwidth measured in the same guide (64 B/clk/CU)

CONFIGS = {
    # name: (preset, per-GPU batch, BASELINE.json reference)
    "independent_8x_256": ("independent_8x_256", 8, "configs[1]"),
    "guided_8x_256": ("guided_8x_256", 8, "configs[3]"),
    "independent_32x_512": ("independent_32x_512", 1, "configs[4]"),
}


def kernel_peak(name):
    """fp32-equivalent matrix-core peak of a kernel: an fp32 multiply-add costs 6 bf16 MFMA products (bf16x3), 3 fp16
    MFMA products (fp16x2, same MFMA rate) or one v_mfma_f32 product."""
    if "1term" in name:
        return F16_MFMA_PEAK_TFLOPS
    if "bf16x3" in name:
        return F16_MFMA_PEAK_TFLOPS / 6.0
    if "f16x2" in name or "spade_fused" in name:
        return F16_MFMA_PEAK_TFLOPS / 3.0
    return FP32_MFMA_PEAK_TFLOPS


def synthetic_batch(opt, n, seed, device):
    """blocky 19-class label map (16x16 cells, nearest-upsampled) + uniform [-1,1] image."""
    g = torch.Generator().manual_seed(seed)
    h = opt.crop_size

    def pair():
        cells = torch.randint(0, opt.label_nc, (n, 1, 16, 16), generator=g).float()
        label = torch.nn.functional.interpolate(cells, size=(h, h), mode="nearest")
        image = torch.rand(n, 3, h, h, generator=g) * 2 - 1
        return label.to(device), image.to(device)

    label, image = pair()
    out = {"label": label, "image": image}
    if getattr(opt, "guiding_style_image", False):
        out["guiding_label"], out["guiding_image"] = pair()
    return out


def effective_cores():
    """Host cores this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    """CPU model string of the host (/proc/cpuinfo)."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_ram_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemTotal"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline(bs=1, iters=3):
    """The oracle's G+D iteration (32->256) on the host cores: 1 untimed warm-up + `iters` timed iterations at batch `bs`
    (BASELINE.md 3).  The default (bs = 1, 3 iterations) is ~30 s of CPU work; `--cpu-baseline-bs 8` times the benchmark's
    own batch (needs >= 64 GB of host RAM, ~7 GB RSS per image, minutes of CPU time)."""
    from oracle import deepsee_oracle as O
    cores = effective_cores()
    torch.set_num_threads(cores)
    small = O.make_opt(start_size=4, crop_size=32, load_size=32, batchSize=1)
    o = O.Oracle(small, O.init_state(small, seed=0))
    b = O.synthetic_batch(small, 1, seed=1)
    o.run_generator_one_step(b)           # untimed: library warm-up on a tiny config
    ram = host_ram_gb()
    if bs > 1 and ram < 64:
        bs = 1
    opt = O.make_opt(batchSize=bs)
    orc = O.Oracle(opt, O.init_state(opt, seed=0))
    batch = O.synthetic_batch(opt, bs, seed=1234)
    times = []
    for it in range(iters + 1):           # iteration 0 is the warm-up at the real size
        t0 = time.perf_counter()
        orc.run_generator_one_step(batch)
        orc.run_discriminator_one_step(batch)
        if it:
            times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    return {"value": bs / dt, "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model(), "host_ram_gb": round(ram, 1), "batch": bs, "iterations": len(times),
            "s_per_iteration": [round(t, 2) for t in times],
            "sample": "1 warm-up + %d timed G+D iterations, bs=%d, independent 8x 32->256 fp32, oracle/deepsee_oracle.py "
                      "(PyTorch-CPU restatement pinned to the reference), %.1f s per iteration" % (len(times), bs, dt)}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-execute this script under torch.distributed.run with one
    rank per GPU (the in-process counterpart of the reference's DataParallel wrap, base_manager.py:15-23) and pass the
    single JSON line of rank 0 through."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, effective_cores() // n))))   # each rank pins itself to its own core
    # set (deepsee_amd.parallel.pin_rank_cores)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (counters cannot be read live)."""
    for name in ("r06_pmc_traffic_fp16.json" if "1term" in kernel else "r06_pmc_traffic.json", "r06_pmc_traffic.json",
                 "r05_pmc_traffic_fp16.json" if "1term" in kernel else "r05_pmc_traffic.json", "r05_pmc_traffic.json",
                 "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            rec = json.load(open(path)).get(kernel)
            if rec:
                return rec["bytes_per_launch"], "profiles/%s: %s" % (name, rec["source"])
    return None, None


def dp_report(tm, step, fence, world, rank, n, dev, elapsed_local, elapsed, steps):
    """world > 1: what a reader needs to check an N-rank run without trusting `n_gpus`.  Every rank calls this
    (collectives inside); returns the `dp` object on every rank.
      ranks_seen        sum of ones through the SAME communicator the gradients travel on
      per_rank          every rank's own wall time over the timed region -> img/s min / max
      allreduce         the two flat gradient buffers (G + E, D) reduced alone with the step's chunking, HIP events on the
                        stream the collectives run on: ms per step, bytes, bus GB/s = 2 (N-1)/N bytes / t (ring convention)
      compute_only      the same step with the collectives switched off (every rank steps on its own gradient), all ranks
                        at once -> `exposed_comm_ms` = step - compute-only step (what the all-reduce costs after overlap)
      n1_reference      rank 0 stepping ALONE while the other ranks wait: the 1-process number of this job, to be compared with
                        the N = 1 line of the same box"""
    import torch.distributed as dist
    hook = tm.optimizer_G.reduce_hook
    comm = getattr(hook, "comm", None)
    backend = dist.get_backend() if dist.is_initialized() else "none"

    def allreduce_(t):
        if comm is not None:
            comm.all_reduce_sum_(t)
        else:
            dist.all_reduce(t)
        return t

    ones = allreduce_(torch.ones(1, device=dev))
    torch.cuda.synchronize()
    times = torch.zeros(world, dtype=torch.float64, device=dev)
    times[rank] = elapsed_local
    dist.all_reduce(times)
    per_rank = [n * steps / float(t) for t in times.tolist()]
    # ---- the gradient exchange alone (buffers as the last step left them; sums of sums are harmless here: scratch copies)
    bufs = [o.grad.clone() for o in (tm.optimizer_G, tm.optimizer_D) if o is not None]
    nbytes = sum(b.numel() * 4 for b in bufs)
    reps = 5
    stream = comm.stream if comm is not None else torch.cuda.current_stream()
    fence()
    ar_ms = []
    for it in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if comm is not None:
            stream.wait_stream(torch.cuda.current_stream())
        e0.record(stream)
        works = []
        for o, b in zip((tm.optimizer_G, tm.optimizer_D), bufs):
            works += hook.start(b, [(lo, hi) for _, _, lo, hi in o.chunk_ranges(hook.chunk_elems)])
        for w in works:
            if w is not None:
                w.wait()
        e1.record(stream if comm is not None else torch.cuda.current_stream())
        torch.cuda.synchronize()
        if it:
            ar_ms.append(e0.elapsed_time(e1))
    ar = sorted(ar_ms)[len(ar_ms) // 2]
    t = torch.tensor([ar], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ar = float(t)

    def timed_steps(k=3):
        step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / k

    # ---- compute only: no collectives, all ranks concurrently (the parameters of the ranks drift apart from here on: this
    #      runs AFTER everything that is reported as the job's throughput)
    fence()
    hook.active = False
    try:
        dt = timed_steps()
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        compute_only = float(t)
        fence()
        alone = timed_steps() if rank == 0 else 0.0
        fence()
        t = torch.tensor([alone], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        alone = float(t)
    finally:
        hook.active = True
    step_s = elapsed / steps
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        rccl = "unknown"
    return {
        "ranks_seen": int(round(float(ones))), "world": world, "backend": backend,
        "communicator": "dsee_comm_* (C ABI, RCCL via dlopen)" if comm is not None else "torch.distributed (%s)" % backend,
        "rccl_version": rccl,
        "per_rank_img_s": {"min": min(per_rank), "max": max(per_rank), "all": [round(v, 3) for v in per_rank]},
        "allreduce": {"bytes_per_step": nbytes, "chunk_mb": hook.chunk_elems * 4 / 2 ** 20, "ms_per_step_alone": ar,
                      "bus_gbps": 2.0 * (world - 1) / world * nbytes / (ar * 1e-3) / 1e9 if ar > 0 else None,
                      "note": "G+E and D flat gradient buffers reduced alone with the step's chunking (median of %d), HIP events "
                              "on the collective stream, max over ranks; bus GB/s in the ring convention 2 (N-1)/N bytes / t" % reps},
        "compute_only": {"ms_per_step": compute_only * 1e3, "img_s": n * world / compute_only,
                         "note": "same step, collectives off, all ranks at once (3 steps, max over ranks)"},
        "exposed_comm_ms_per_step": (step_s - compute_only) * 1e3,
        "n1_reference": {"ms_per_step": alone * 1e3, "img_s": n / alone if alone > 0 else None,
                         "note": "rank 0 stepping alone, collectives off, the other ranks idle (3 steps): the 1-process rate "
                                 "inside this job"},
        "scaling_vs_n1_reference": (n * world / step_s) / (n / alone) if alone > 0 else None,
    }


def plan_overrides(pairs):
    """--plan FIELD=VALUE ...: values are Python literals, fields those of deepsee_amd.plan.KernelPlan."""
    import ast
    from deepsee_amd.plan import KernelPlan
    out = {}
    for kv in pairs:
        key, sep, val = kv.partition("=")
        if not sep or key not in KernelPlan.fields():
            raise SystemExit("bench.py --plan: %r is not FIELD=VALUE with FIELD one of %s" % (kv, ", ".join(KernelPlan.fields())))
        try:
            out[key] = ast.literal_eval(val)
        except (ValueError, SyntaxError):
            raise SystemExit("bench.py --plan %s: %r is not a Python literal" % (key, val))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="independent_8x_256")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-bs", type=int, default=1,
                    help="batch of the CPU baseline (8 = the benchmark's own batch: >= 64 GB of host RAM, minutes of CPU time)")
    ap.add_argument("--cpu-baseline-iters", type=int, default=3)
    ap.add_argument("--no-f32-run", action="store_true", help="skip the extra v_mfma_f32-only measurement")
    ap.add_argument("--batch-per-gpu", type=int, default=0)
    ap.add_argument("--no-graphs", action="store_true", help="enqueue every kernel from Python instead of replaying hipGraphs")
    ap.add_argument("--dtype", choices=["fp32", "fp16"], default="fp32",
                    help="fp16: BASELINE configs[2]'s 16-bit arithmetic (one-term scaled-fp16 matrix-core GEMMs, fp32 master "
                         "weights): a separate line")
    ap.add_argument("--arith", choices=["f16x2", "bf16x3", "f32"], default="f16x2",
                    help="how the fp32 Winograd-domain GEMMs run on the matrix cores: two-term fp16 split (default), exact "
                         "3-term bf16 split, or v_mfma_f32")
    ap.add_argument("--dp-comm", choices=["torch", "capi"], default="torch",
                    help="--gpus N > 1: the collectives through torch.distributed (RCCL backend) or through dsee_comm_* of the C ABI")
    ap.add_argument("--plan", nargs="*", default=[], metavar="FIELD=VALUE",
                    help="KernelPlan fields of the model (deepsee_amd/plan.py), e.g. --plan fused_norm=False (A/B runs)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)

    from deepsee_amd import ops, parallel
    from deepsee_amd.managers import TrainerManager
    from deepsee_amd.options import make_opt
    import torch.distributed as dist

    rank, local, world = parallel.init_distributed()
    if world != args.gpus:   # never report a number for a world size other than the one asked for
        raise SystemExit("bench.py: --gpus %d but the process group has %d rank(s)" % (args.gpus, world))
    dev = torch.device("cuda", local)
    preset, n_default, ref = CONFIGS[args.config]
    n = args.batch_per_gpu or n_default
    headline = args.config == "independent_8x_256"
    # the kernel-path choices of this run are the MODEL's plan (deepsee_amd/plan.py), not module state
    opt = make_opt(preset, batchSize=n, seed=0, precision=args.dtype, hip_graphs=not args.no_graphs, dp_comm=args.dp_comm,
                   kernel_plan=dict(dict(gemm_split=args.arith != "f32", gemm_f16x2=args.arith == "f16x2"),
                                    **plan_overrides(args.plan)))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)   # (the no-pretrained-VGG notice: synthetic benchmark)
        tm = TrainerManager(opt)            # encoder-branch coins: DeviceNoise's own RNG, identical on every rank
    parallel.attach(tm, world)              # gradient all-reduce hooks, rank-0 broadcast, per-rank noise seed
    batch = synthetic_batch(opt, n, 1234 + rank, dev)

    def step():
        tm.run_generator_one_step(batch)
        tm.run_discriminator_one_step(batch)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # hipGraphs: every encoder-branch variant runs eagerly once and is captured on its second occurrence; keep warming up
    # (untimed) until four consecutive steps were pure replays, so that no capture lands in the timed region
    extra_warmup, quiet = 0, 0
    while tm.use_graphs and quiet < 4 and extra_warmup < 96:
        before = tm.graph_stats["eager"] + tm.graph_stats["captured"]
        step()
        extra_warmup += 1
        quiet = quiet + 1 if tm.graph_stats["eager"] + tm.graph_stats["captured"] == before else 0
    fence()
    replays_before = tm.graph_stats["replayed"]
    host_steps = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        th = time.perf_counter()
        step()
        host_steps.append(time.perf_counter() - th)    # launch enqueue only: nothing inside step() waits for the device
    fence()
    # Python time of a step whose launches did not have to wait for queue space: once the host is more than a few graph
    # launches ahead of the GPU, hipGraphLaunch blocks until the device catches up and the time measured is the GPU's, not
    # the host's -- the median of the first four timed steps (queues empty after the fence) is the enqueue cost itself
    host = sorted(host_steps[:4])[len(host_steps[:4]) // 2] * args.steps
    elapsed = elapsed_local = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    # ---- one extra instrumented step: per-launch HIP events around every MFMA kernel and the fused SPADE kernel
    from deepsee_amd import lib as _L
    ops.PROFILE = {}
    ops.PROFILE_BYTES.clear()
    calls_before = _L.CALLS
    step()
    torch.cuda.synchronize()
    capi_calls_per_step = _L.CALLS - calls_before
    prof, ops.PROFILE = ops.PROFILE, None
    kernels = {}
    for name, recs in prof.items():
        ms = sum(s.elapsed_time(e) for s, e, _ in recs)
        kernels[name] = {"launches": len(recs), "ms": ms, "tflop": sum(f for _, _, f in recs) / 1e12,
                         "gb": ops.PROFILE_BYTES.get(name, 0.0) / 1e9}
    fused_is_r3 = "spade_fused_fwd" in kernels
    fused = kernels.pop("spade_fused_fwd", None) or kernels.pop("spade_modulate_fused", None)
    norm_fwd = {k: kernels.pop(k) for k in list(kernels) if k.startswith("norm_forward@")}
    conv_fwd = {k: kernels.pop(k) for k in list(kernels) if k.startswith("conv_forward@")}
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    kd = kernels[dom]
    tf = kd["tflop"] / (kd["ms"] / 1e3)
    gbps = kd["gb"] / (kd["ms"] / 1e3) if kd["gb"] else 0.0
    peak = kernel_peak(dom)
    f_mfma, f_hbm = tf / peak, gbps / HBM_PEAK_GBPS
    mfma_ms = sum(k["ms"] for k in kernels.values())

    base_plan = tm.sr_model.plan

    def short_run(note, **plan):
        """3 steps of the same model under a plan of its own (eager: the graphs were captured under the base plan)."""
        tm.use_graphs = False
        tm.sr_model.plan = base_plan.replace(**plan)
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(3):
            step()
        fence()
        dt = (time.perf_counter() - t1) / 3
        return {"value": n / dt, "unit": "img/s", "ms_per_step": dt * 1e3, "steps": 3, "note": note}

    f32_only = bf16x3 = None
    if world == 1 and base_plan.gemm_split and not args.no_f32_run and headline and not base_plan.half:
        f32_only = short_run("same step, Winograd-domain GEMMs on v_mfma_f32_32x32x2_f32 instead of split operands",
                             gemm_split=False)
        if base_plan.gemm_f16x2:
            bf16x3 = short_run("same step, every split GEMM on the EXACT 3-term bf16 split (6 MFMA products per fp32 "
                               "multiply-add; the SPADE/SEAN forward then takes the round-2 GEMM + output-transform path)",
                               gemm_f16x2=False)
        tm.sr_model.plan = base_plan
        tm.use_graphs = not args.no_graphs

    dp = dp_report(tm, step, fence, world, rank, n, dev, elapsed_local, elapsed, args.steps) if world > 1 else None

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        kind = "f16x2" if (base_plan.gemm_split and base_plan.gemm_f16x2) else ("bf16x3" if base_plan.gemm_split else "f32")
        if base_plan.half:
            kind = "fp16"
        arithmetic = {
            "fp16": "half-precision compute mode (BASELINE configs[2]'s 16-bit arithmetic): Winograd-domain GEMMs with operands "
                    "scaled by powers of two and rounded to ONE fp16 term, one MFMA product, fp32 accumulate, products stored as "
                    "scaled fp16 (fp16, not bf16: F(4x4,3x3) amplifies operand rounding ~10x); activations, statistics, master "
                    "weights and Adam in fp32; generated image within 3e-2 of the fp32 path "
                    "(tests/test_gpu_model.py::test_half_mode_tracks_fp32)",
            "f16x2": "fp32 storage and accumulation; the wide 3x3 layers run as Winograd F(4x4,3x3) GEMMs on the fp16 "
                     "matrix cores: every fp32 operand is scaled by an exact power of two and split into two fp16 terms "
                     "(residual <= 2^-22, rms 2^-24), 3 MFMA products per multiply-add, fp32 accumulate; error vs float64 "
                     "equal to a CPU sgemm's (tests/test_gpu_conv.py::test_gemm_f16x2_is_fp32_accurate); --arith bf16x3 "
                     "selects the exact 3-term bf16 split (6 products).  The generated image deviates 2e-6 ... 5e-6 from the CPU "
                     "oracle (bound 1e-4), and that deviation is the rounding of the fp32 F(4x4,3x3) transforms (the oracle's own "
                     "distance to float64 is 8e-7), not of the split products; through the LeakyReLU / ReLU kinks it is what "
                     "holds the model-level gradient comparison at a median of 1e-3 ... 2e-3 (DESIGN 4)",
            "bf16x3": "fp32 storage and accumulation; Winograd-domain GEMMs from exact 3-term bf16 operand splits (6 bf16 "
                      "MFMA products)",
            "f32": "fp32 (v_mfma_f32_32x32x2_f32)"}[kind]
        traffic, traffic_src = pmc_traffic(dom)
        # a kernel below half of BOTH datasheet rooflines is bound by neither of them.  Round 6 measured what it IS bound by: the
        # socket power cap (profiles/r06_gemm_ablation.md) -- `power` below prices it against the MFMA rate that cap allows
        split_products = round(F16_MFMA_PEAK_TFLOPS / peak) if peak > 200 else 0
        power_peak = F16_MFMA_POWER_CAP_TFLOPS / split_products if split_products else None
        # (measured for the Winograd-domain GEMM families; a direct-convolution kernel that ends up dominant -- configs[4] at N = 1 in
        # the 16-bit mode -- runs at 2.3-2.45 GHz and is latency / issue bound)
        if not dom.startswith("winograd"):
            power_peak = None
        bound = ("neither datasheet roofline: socket power cap" if (max(f_mfma, f_hbm) < 0.5 and power_peak) else
                 ("neither (issue/latency)" if max(f_mfma, f_hbm) < 0.5 else ("mfma" if f_mfma >= f_hbm else "hbm")))
        roof = {"bound": bound, "closer_to": "mfma" if f_mfma >= f_hbm else "hbm", "kernel": dom,
                "achieved": tf if f_mfma >= f_hbm else gbps, "peak": peak if f_mfma >= f_hbm else HBM_PEAK_GBPS,
                "unit": "TFLOP/s" if f_mfma >= f_hbm else "GB/s", "frac": max(f_mfma, f_hbm),
                "traffic": traffic, "traffic_source": traffic_src,
                "mfma": {"achieved_tflops_fp32_equiv": tf, "peak": peak, "frac": f_mfma,
                         "peak_note": "%.1f dense fp16/bf16 MFMA TFLOP/s / %d products per fp32 multiply-add"
                                      % (F16_MFMA_PEAK_TFLOPS, round(F16_MFMA_PEAK_TFLOPS / peak)) if peak > 200
                         else "v_mfma_f32 dense peak"},
                "hbm": {"achieved_gbps_algorithmic": gbps, "peak": HBM_PEAK_GBPS, "frac": f_hbm},
                "power": None if not power_peak else {
                    "socket_cap_w": 1400, "mfma_only_tflops_under_cap_fp16_random_operands": F16_MFMA_POWER_CAP_TFLOPS,
                    "peak_fp32_equiv_under_cap": power_peak, "frac_of_power_capped_peak": tf / power_peak,
                    "source": "profiles/r06_mfma_power_cap.txt (MFMAs only, operands in registers, rocm-smi beside it), "
                              "profiles/r06_gemm_ablation.md (this kernel back to back: 1 378-1 403 W, 1.54 GHz)"},
                "launches_per_step": kd["launches"], "avg_launch_ms": kd["ms"] / kd["launches"],
                "algorithmic_gb_per_launch": (kd["gb"] / kd["launches"]) or None,
                "algorithmic_tflop_per_step": kd["tflop"],
                "mfma_kernels_ms_per_step": mfma_ms,
                "all_mfma_kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                         "tflops": round(v["tflop"] / (v["ms"] / 1e3), 2),
                                         "frac": round(v["tflop"] / (v["ms"] / 1e3) / kernel_peak(k), 3),
                                         "gbps": round(v["gb"] / (v["ms"] / 1e3), 1) if v["gb"] else None}
                                     for k, v in kernels.items()}}
        losses = {k: float(v.detach()) for k, v in tm.get_latest_losses().items()}
        if not all(math.isfinite(v) for v in losses.values()):
            raise RuntimeError("bench.py: non-finite losses after the timed steps %r -- the measurement is void" % losses)
        out = {
            "metric": "train-step images/sec (G+D fwd+bwd), 8x 32->256 bs=8" if headline
                      else "train-step images/sec (G+D fwd+bwd), %s bs=%d" % (args.config, n),
            "value": n * world / (elapsed / args.steps),
            "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "arithmetic": arithmetic,
            "config": {"workload": "%s, 19-class masks, bs=%d per GPU, %s G+D train step (BASELINE.json %s)"
                                   % ({"independent_8x_256": "independent 8x 32->256", "guided_8x_256": "guided 8x 32->256",
                                       "independent_32x_512": "independent 32x 16->512"}[args.config], n, args.dtype, ref),
                       "global_batch": n * world, "parallelism": "dp%d" % world,
                       "losses": losses},
            "roofline": roof,
        }
        # whole-model rate in the reference's dense-convolution FLOP count (conv MACs x 2 of the reference's
        # graph per image and iteration); the Winograd layers execute 2.25x fewer multiplies than that
        dense = {"independent_8x_256": 7.15, "guided_8x_256": 7.15, "independent_32x_512": 22.6}[args.config]
        out["model_flops"] = {"reference_dense_tflop_per_image": dense, "achieved_tflops": out["value"] * dense,
                              "fp32_matrix_peak_tflops": 157.3 * world,
                              "note": "N * F_iter / t_iter with SURVEY 8(d)'s F_iter; exceeds the fp32 MFMA peak because "
                                      "the wide layers run on the fp16 matrix cores in the Winograd domain"}
        out["peak_hbm_gb"] = round(torch.cuda.max_memory_allocated(dev) / 1e9, 2)   # of 288 GB (kept V / M tensors included)
        out["launches_per_step"] = {"c_abi_entry_points": capi_calls_per_step,
                                    "note": "dsee_* calls of one eager G + D step (an entry point is 1-3 kernel launches); the "
                                            "rocprofv3 count of ALL kernels incl. ATen glue is in profiles/r06_kernel_stats_all.md"}
        out["host_enqueue_ms_per_step"] = host / args.steps * 1e3
        out["host_ms_per_step_incl_queue_backpressure"] = sum(host_steps) / args.steps * 1e3
        out["hip_graphs"] = {"enabled": bool(tm.use_graphs), "captured": sorted("/".join(map(str, k[:3])) for k in tm._graphs),
                             "extra_warmup_steps": extra_warmup,
                             "timed_half_steps_replayed": tm.graph_stats["replayed"] - replays_before, "timed_half_steps": 2 * args.steps,
                             "parallel_branches": bool(tm.sr_model.plan.branch_streams),
                             "note": "G and D step replayed as hipGraphs (one per encoder-branch variant; first occurrence "
                                     "eager, second captured); host_enqueue_ms_per_step is the Python time per step.  "
                                     "parallel_branches (the two D scales and VGG fake / real as parallel graph branches, "
                                     "--plan branch_streams=True: ~1.5 ms per step, profiles/r05_bench_branch_streams.json) is off "
                                     "by default: replaying such graphs segfaults inside the HIP runtime late in a long-lived "
                                     "process (profiles/r05_graph_branch_segv.txt)"}
        if fused and not fused_is_r3:
            # 16-bit mode / --arith bf16x3: the norms take the round-2 path (gamma/beta GEMM, then this fused output transform)
            g = fused["gb"] / (fused["ms"] / 1e3)
            out["spade_fused"] = {
                "kernel": "wino43_output_modulate (round-2 form: output transform + BN-normalise + SPADE/SEAN modulate + "
                          "LeakyReLU over the gamma/beta GEMM's product M; the fused round-3 kernel exists in the two-term "
                          "fp16x2 form only)",
                "bound": "hbm", "launches": fused["launches"], "ms_per_step": fused["ms"], "achieved": g,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": g / HBM_PEAK_GBPS,
                "bytes_note": "bytes the kernel moves: M read, x read, h and scale written"}
        elif fused:
            g = fused["gb"] / (fused["ms"] / 1e3)
            out["spade_fused"] = {
                "kernel": "spade_fused_fwd (gamma/beta Winograd GEMM, output transform folded in registers, BN-normalise + "
                          "SPADE/SEAN modulate + LeakyReLU epilogue; the Winograd-domain product never reaches HBM)",
                "bound": "l2-lds operand path", "launches": fused["launches"], "ms_per_step": fused["ms"],
                "achieved": g, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": g / HBM_PEAK_GBPS,
                "bytes_note": "algorithmic HBM bytes of the kernel: split V (4 B x 36/16 x K per pixel) read, x read, h [+ "
                              "scale] written",
                "mfma_tflops_fp32_equiv": fused["tflop"] / (fused["ms"] / 1e3),
                "mfma_frac": fused["tflop"] / (fused["ms"] / 1e3) / kernel_peak("spade_fused"),
                "operand_path": {"gb_per_step": ops.PROFILE_OPERAND_GB, "achieved_gbps": ops.PROFILE_OPERAND_GB / (fused["ms"] / 1e3),
                                 "peak": L2_PEAK_GBPS, "frac": ops.PROFILE_OPERAND_GB / (fused["ms"] / 1e3) / L2_PEAK_GBPS,
                                 "note": "a 64 tile x 64 row workgroup pulls 2 x 64 x K x 4 B of split operands per transform "
                                         "position through the L2 -> LDS path (LDS-DMA, 64 B/clk/CU)"}}
        if norm_fwd:
            top = max(norm_fwd, key=lambda k: norm_fwd[k]["gb"] / norm_fwd[k]["launches"])
            v = norm_fwd[top]
            g = v["gb"] / (v["ms"] / 1e3)
            out["roofline"]["norm_forward"] = {
                "what": "whole SPADE/SEAN normalisation forward (statistics + embedding + transforms + fused kernel) at the top "
                        "resolution, " + top.split("@")[1],
                "launches": v["launches"], "ms_per_call": v["ms"] / v["launches"],
                "algorithmic_gb_per_call": v["gb"] / v["launches"], "achieved": g, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": g / HBM_PEAK_GBPS,
                "bytes_note": "SURVEY 8(d): 4 N R^2 (C [x, statistics pass] + C [x, apply pass] + C [out] + 128 [embedding]) + N R^2"}
        if conv_fwd:
            top = max(conv_fwd, key=lambda k: conv_fwd[k]["gb"] / conv_fwd[k]["launches"])
            v = conv_fwd[top]
            g, tfl = v["gb"] / (v["ms"] / 1e3), v["tflop"] / (v["ms"] / 1e3)
            pk = kernel_peak("winograd_gemm_" + kind) if kind in ("f16x2", "bf16x3") else (
                F16_MFMA_PEAK_TFLOPS if kind == "fp16" else FP32_MFMA_PEAK_TFLOPS)
            out["roofline"]["conv_layer"] = {
                "what": "one whole 3x3 convolution forward at the top resolution (input transform + Winograd-domain GEMM + "
                        "output transform), " + top.split("@")[1],
                "launches": v["launches"], "ms_per_call": v["ms"] / v["launches"],
                "algorithmic_gb_per_call": v["gb"] / v["launches"], "hbm_achieved_gbps": g, "hbm_frac": g / HBM_PEAK_GBPS,
                "winograd_tflop_per_call": v["tflop"] / v["launches"], "mfma_achieved_tflops_fp32_equiv": tfl,
                "mfma_peak": pk, "mfma_frac": tfl / pk,
                "bytes_note": "algorithmic bytes = x read + y written (4 N R^2 (Cin + Cout)); the Winograd formulation "
                              "additionally writes and re-reads V and M (DESIGN 3)"}
        if base_plan.half:
            # BASELINE configs[2] names bf16; this line runs scaled fp16.  The deviation and why, where the driver reads it
            # (VERDICT r4 #7b; numbers: DESIGN 6.2, tests/test_gpu_model.py::test_half_mode_vs_oracle / test_benchmark_path_*)
            out["baseline_dtype_deviation"] = {
                "baseline_names": "bf16", "this_line_runs": "fp16 operands scaled by exact powers of two, fp32 accumulate",
                "why": "F(4x4,3x3) amplifies operand rounding ~10x: a Winograd layer on bf16 operands (8 significand bits) is "
                       "off by 2.6 % per layer, on scaled fp16 operands (11 bits) by 0.33 %; fp16's 5-bit exponent suffices "
                       "because every operand tensor carries a power-of-two scale derived from its producer's bound",
                "per_layer_relative_error": {"winograd_bf16_operands": 2.6e-2, "winograd_bf16_operands_and_products": 3.2e-2,
                                             "winograd_scaled_fp16_operands": 3.3e-3,
                                             "winograd_scaled_fp16_operands_and_products": 4.0e-3,
                                             "direct_conv_bf16_operands": 2.4e-3},
                "generated_image_vs_fp32_path": {"bf16_operands_first_version": 7.9e-2, "scaled_fp16": 1.0e-2},
                "generated_image_vs_cpu_oracle": {"bs1": 6.0e-3, "bs8": 8.6e-3, "bound": 3e-2},
                "what_a_bf16_path_would_need": "a direct (non-Winograd) bf16 MFMA implicit GEMM: 2.25x the MFMA work of "
                                               "F(4x4,3x3) at 0.24 % per layer",
                "direct_bf16_measured_ceiling": {
                    "mfma_only_bf16_random_operands_tflops": 1734.3, "mfma_only_bf16_zero_operands_tflops": 2459.9,
                    "needed_to_beat_the_fp16_winograd_layer_tflops": 1270.0,
                    "note": "512 -> 512 @256^2, N = 8 is 2.47 TFLOP direct: 1.42 ms at the MFMA rate the 1 400 W socket cap allows "
                            "with NOTHING but MFMAs on random operands in registers; the loops that also move operands reach 0.50-0.67 "
                            "of that ceiling (2.1-2.8 ms), the fp16 Winograd layer takes 1.95 ms -- profiles/r06_direct_bf16.txt"}}
        if dp:
            out["dp"] = dp
        if f32_only:
            out["f32_mfma_only"] = f32_only
        if bf16x3:
            out["bf16x3_exact"] = bf16x3
        if world == 1 and not args.no_cpu_baseline and headline and not base_plan.half:
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_bs, args.cpu_baseline_iters)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        tm.close()                   # (graphs may hold captured RCCL operations: gone before their communicator)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
