"""Wall time of the G step and the D step separately (bs=8, 8x 32->256), plus forward/backward split of the G step."""
import os, sys, random, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsee_amd.managers import TrainerManager
from deepsee_amd.options import make_opt
from bench import synthetic_batch
opt = make_opt("independent_8x_256", batchSize=8, seed=0)
random.seed(1234)
tm = TrainerManager(opt)
batch = synthetic_batch(opt, 8, 1234, "cuda")
def t(fn, n=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
tm.run_generator_one_step(batch); tm.run_discriminator_one_step(batch)
print("G step %.1f ms | D step %.1f ms" % (t(lambda: tm.run_generator_one_step(batch)), t(lambda: tm.run_discriminator_one_step(batch))))
m = tm.sr_model
d = m._native(tm.preprocess_input({k: v.clone() for k, v in batch.items()}))
with torch.no_grad():
    print("G forward only (no grad) %.1f ms" % t(lambda: m.generate_fake(d)))
    fake, _ = m.generate_fake(d)
    print("D forward only (no grad) %.1f ms" % t(lambda: m.discriminate(d["labels"], fake, d["image_hr"], train_d=True)))
    print("VGG forward (fake) %.1f ms" % t(lambda: m.vgg(fake)))
