"""Smoke the other BASELINE configs at full size on one GPU: guided 8x 32->256 (config 4, bs=8) and independent
32x 16->512 (config 5, bs=1): a few G+D steps each, finite losses, timing."""
import random, sys, time, torch
sys.path.insert(0, ".")
from deepsee_amd.managers import TrainerManager
from deepsee_amd.options import make_opt
import bench


def run(name, preset, n, steps=3, **over):
    opt = make_opt(preset, batchSize=n, seed=0, **over)
    random.seed(1)
    tm = TrainerManager(opt)
    b = bench.synthetic_batch(opt, n, 7, "cuda")
    def step():
        tm.run_generator_one_step(b); tm.run_discriminator_one_step(b)
    step(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    losses = {k: round(float(v.detach()), 4) for k, v in tm.get_latest_losses().items()}
    assert all(l == l and abs(l) < 1e6 for l in losses.values()), losses
    print("%s: %.1f ms/step, %.2f img/s, peak mem %.1f GB, losses %s" % (name, dt * 1e3, n / dt, torch.cuda.max_memory_allocated() / 2**30, losses), flush=True)
    del tm; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()


run("config4 guided 8x 32->256 bs=8", "guided_8x_256", 8)
run("config5 independent 32x 16->512 bs=1", "independent_32x_512", 1)
run("config2 independent 8x 32->256 bs=8", "independent_8x_256", 8)
