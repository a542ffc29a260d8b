#!/usr/bin/env python
"""Benchmark of the DeepSEE G+D training step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = TrainerManager.run_generator_one_step + run_discriminator_one_step (train.py:40-44) on one synthetic
batch that is already resident in HBM: independent 8x 32->256, 19-class blocky masks, bs=8 per GPU, fp32
(BASELINE.json configs[1]; weak scaling: the per-GPU batch is fixed as N grows).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     — the dominant kernel (by summed time in a step), measured live with HIP events on the launch
                 stream in one extra instrumented step after the timed region: algorithmic fp32 FLOPs (2*M*N*K of
                 every launch of that kernel) / summed duration.  Peak: 157.3 TFLOP/s for the v_mfma_f32 kernels;
                 for the bf16x3 kernels (fp32 multiply = 6 exact bf16 MFMA products, deepsee_amd/csrc/gemm_bf16x3.hip)
                 the dense bf16 MFMA peak / 6 = 419.4 TFLOP/s of fp32 work.
  f32_mfma_only— the same step with every GEMM kept on v_mfma_f32_32x32x2_f32 (DSEE_F32_MFMA=1 path), rank 0 / N=1.
  cpu_baseline — the oracle (CPU restatement of the reference path, oracle/deepsee_oracle.py) timed on this box's
                 host cores: one G+D iteration at bs=1 of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md
BF16_MFMA_PEAK_TFLOPS = 2516.6  # dense (same guide); a bf16x3 fp32 multiply-add costs 6 bf16 MFMA products


# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `bench.py --steps 2 --warmup 1` (372 launches of the kernel):
# FETCH_SIZE 0.2816 GB reported -> x2 (gfx950 correction, MI355X_MICROARCH.md) = 0.563 GB, WRITE_SIZE 0.546 GB per launch
PMC_TRAFFIC_BYTES_PER_LAUNCH = {"winograd_gemm_bf16x3": 1.109e9}


def kernel_peak(name):
    """fp32-equivalent peak of a kernel: an fp32 multiply-add costs 6 bf16 MFMA products (bf16x3), 3 fp16 MFMA products
    (fp16x2, same MFMA rate) or one v_mfma_f32 product."""
    if "bf16x3" in name:
        return BF16_MFMA_PEAK_TFLOPS / 6.0
    if "f16x2" in name:
        return BF16_MFMA_PEAK_TFLOPS / 3.0
    return FP32_MFMA_PEAK_TFLOPS
N_PER_GPU = 8


def synthetic_batch(opt, n, seed, device):
    """blocky 19-class label map (16x16 cells, nearest-upsampled) + uniform [-1,1] image."""
    g = torch.Generator().manual_seed(seed)
    h = opt.crop_size
    cells = torch.randint(0, opt.label_nc, (n, 1, 16, 16), generator=g).float()
    label = torch.nn.functional.interpolate(cells, size=(h, h), mode="nearest")
    image = torch.rand(n, 3, h, h, generator=g) * 2 - 1
    return {"label": label.to(device), "image": image.to(device)}


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


def cpu_baseline():
    """Oracle G+D iteration at bs=1 (32->256) on the host cores; ~10-30 s of CPU work."""
    from oracle import deepsee_oracle as O
    cores = effective_cores()
    torch.set_num_threads(cores)
    small = O.make_opt(start_size=4, crop_size=32, load_size=32, batchSize=1)
    o = O.Oracle(small, O.init_state(small, seed=0))
    b = O.synthetic_batch(small, 1, seed=1)
    o.run_generator_one_step(b)           # untimed: library warm-up on a tiny config
    opt = O.make_opt()
    orc = O.Oracle(opt, O.init_state(opt, seed=0))
    batch = O.synthetic_batch(opt, 1, seed=1234)
    t0 = time.perf_counter()
    orc.run_generator_one_step(batch)
    orc.run_discriminator_one_step(batch)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 G+D iteration, bs=1, independent 8x 32->256 fp32, oracle/deepsee_oracle.py (PyTorch-CPU "
                      "restatement pinned to the reference), %.1f s" % dt}


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
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-run", action="store_true", help="skip the extra v_mfma_f32-only measurement")
    ap.add_argument("--batch-per-gpu", type=int, default=N_PER_GPU)
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
    n = args.batch_per_gpu
    opt = make_opt("independent_8x_256", batchSize=n, seed=0)
    tm = TrainerManager(opt)                # encoder-branch coins: DeviceNoise's own RNG, identical on every rank
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
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    # ---- one extra instrumented step: per-launch HIP events around every MFMA conv kernel
    ops.PROFILE = {}
    ops.PROFILE_BYTES.clear()
    step()
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    kernels = {}
    for name, recs in prof.items():
        ms = sum(s.elapsed_time(e) for s, e, _ in recs)
        kernels[name] = {"launches": len(recs), "ms": ms, "tflop": sum(f for _, _, f in recs) / 1e12}
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    kd = kernels[dom]
    achieved = kd["tflop"] / (kd["ms"] / 1e3)
    peak = kernel_peak(dom)
    mfma_ms = sum(k["ms"] for k in kernels.values())

    f32_only = None
    if world == 1 and ops.GEMM_SPLIT and not args.no_f32_run:
        ops.GEMM_SPLIT = False
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(3):
            step()
        fence()
        dt = (time.perf_counter() - t1) / 3
        ops.GEMM_SPLIT = True
        f32_only = {"value": n / dt, "unit": "img/s", "ms_per_step": dt * 1e3, "steps": 3,
                    "note": "same step, Winograd-domain GEMMs on v_mfma_f32_32x32x2_f32 instead of bf16x3"}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "train-step images/sec (G+D fwd+bwd), 8x 32->256 bs=8",
            "value": n * world / (elapsed / args.steps),
            "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp32", "data": "synthetic",
            "arithmetic": ("fp32 storage and accumulation; the wide 3x3 layers run as Winograd F(4x4,3x3) GEMMs whose "
                           "fp32 products are formed exactly from 3-term bf16 operand splits (6 bf16 MFMA products, "
                           "error below one fp32 rounding; tests/test_gpu_conv.py::test_gemm_bf16x3_is_fp32_accurate)"
                           if ops.GEMM_SPLIT else "fp32 (v_mfma_f32_32x32x2_f32)"),
            "config": {"workload": "independent 8x 32->256, 19-class masks, bs=%d per GPU, fp32 G+D train step "
                                   "(BASELINE.json configs[1])" % n,
                       "global_batch": n * world, "parallelism": "dp%d" % world,
                       "losses": {k: float(v.detach()) for k, v in tm.get_latest_losses().items()}},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak,
                         # HBM bytes per launch of this kernel from PMC passes over this same command (not live:
                         # counters need rocprofv3), profiles/r01_pmc_gemm_bf16x3.md
                         "traffic": PMC_TRAFFIC_BYTES_PER_LAUNCH.get(dom),
                         "peak_note": ("fp32 work on the bf16 matrix cores: 2516.6 dense bf16 TFLOP/s / 6 products per "
                                       "fp32 multiply-add" if "bf16x3" in dom else "v_mfma_f32 dense peak"),
                         "launches_per_step": kd["launches"], "avg_launch_ms": kd["ms"] / kd["launches"],
                         "algorithmic_gb_per_launch": (ops.PROFILE_BYTES.get(dom, 0.0) / kd["launches"] / 1e9) or None,
                         "traffic_note": ("PMC passes over this command (profiles/r01_pmc_gemm_bf16x3.md): FETCH_SIZE x2 "
                                          "(gfx950 correction) + WRITE_SIZE, averaged over the step's launches of this "
                                          "kernel; 7.04 GB vs 6.06 GB algorithmic at the dominant shape"
                                          if "bf16x3" in dom else None),
                         "algorithmic_tflop_per_step": kd["tflop"],
                         "mfma_kernels_ms_per_step": mfma_ms,
                         "all_mfma_kernels": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                                  "tflops": round(v["tflop"] / (v["ms"] / 1e3), 2),
                                                  "frac": round(v["tflop"] / (v["ms"] / 1e3) / kernel_peak(k), 3)}
                                              for k, v in kernels.items()}},
        }
        if f32_only:
            out["f32_mfma_only"] = f32_only
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
