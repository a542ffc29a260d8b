import os, sys, time, torch
sys.path.insert(0, ".")
from oracle import deepsee_oracle as O
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p): print(p, open(p).read().strip())
opt = O.make_opt(start_size=8, crop_size=64, load_size=64, batchSize=1)
st = O.init_state(opt, 0); b = O.synthetic_batch(opt, 1, seed=1)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    o = O.Oracle(opt, st)
    o.run_generator_one_step(b)
    t = time.perf_counter(); o.run_generator_one_step(b); o.run_discriminator_one_step(b)
    print("threads", nt, "8->64 bs=1 G+D: %.2f s" % (time.perf_counter() - t), flush=True)
