"""hipGraph replay of the G / D steps vs the eager steps: same seeds -> same parameters after K iterations."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import warnings
warnings.simplefilter("ignore", RuntimeWarning)
from deepsee_amd.managers import TrainerManager
from deepsee_amd.options import make_opt
from deepsee_amd import networks as N
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench


def run(graphs, iters, over):
    opt = make_opt("independent_8x_256", seed=3, hip_graphs=graphs, **over)
    tm = TrainerManager(opt)
    batch = bench.synthetic_batch(opt, opt.batchSize, 77, torch.device("cuda"))
    hist = []
    for i in range(iters):
        tm.run_generator_one_step(batch)
        tm.run_discriminator_one_step(batch)
        hist.append({k: float(v) for k, v in tm.get_latest_losses().items()})
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1) for p in tm.sr_model.parameters()]).cpu()
    return hist, params, tm


if __name__ == "__main__":
    small = dict(start_size=4, crop_size=32, load_size=32, batchSize=2, ngf=8) if len(sys.argv) < 2 else {}
    iters = 10
    h0, p0, _ = run(False, iters, small)
    h1, p1, tm = run(True, iters, small)
    print("graphs captured:", sorted(tm._graphs))
    for a, b in zip(h0, h1):
        print({k: (round(a[k], 6), round(b[k], 6)) for k in a})
    print("params: max |eager - graph| =", float((p0 - p1).abs().max()), " rel", float((p0 - p1).norm() / p0.norm()))
