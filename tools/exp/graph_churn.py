"""Does the hipGraph path survive many managers / captures in ONE process?  (batch 6 / 8: a replay segfaults inside the HIP runtime
in tests/test_gpu_model.py::test_training_loop_with_device_loader_and_metrics, but only in the full suite.)
usage: graph_churn.py [rounds] [loader|plain] [close|keep]"""
import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deepsee_amd import data as D
from deepsee_amd.managers import TrainerManager
from deepsee_amd.options import make_opt

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
use_loader = (sys.argv[2] if len(sys.argv) > 2 else "loader") == "loader"
close = (sys.argv[3] if len(sys.argv) > 3 else "keep") == "close"
over = dict(start_size=8, crop_size=64, load_size=64, batchSize=2, ngf=8)
keep = []
for r in range(rounds):
    tm = TrainerManager(make_opt(seed=11 + r, **over))
    ds = D.SyntheticDataset(tm.opt, length=12, seed=4)
    loader = D.DeviceLoader(ds, tm.opt, shuffle=True, seed=9)
    for batch in loader:
        if not use_loader:
            batch = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        tm.run_generator_one_step(batch)
        tm.run_discriminator_one_step(batch)
    torch.cuda.synchronize()
    print("round %d: %s  mem %.1f GB" % (r, tm.graph_stats, torch.cuda.memory_reserved() / 1e9), flush=True)
    if close:
        tm.close()
    else:
        keep.append(tm)
print("survived", rounds, "rounds")
