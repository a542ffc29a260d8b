"""Socket power and shader clock while one kernel runs back to back (rocm-smi sampled from a thread every ~100 ms).
usage: power_probe.py <entry> [seconds]   (DSEE_LIB selects a measurement build)"""
import os, sys, subprocess, threading, time, re
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from deepsee_amd import lib as L
entry = sys.argv[1]; secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
res = int(os.environ.get("RES", "256")); zero = os.environ.get("ZERO") == "1"
t, k, n = 8 * (res // 4) ** 2, 512, 512
M = 36 * t
pk = "f16p" in entry
gen = (lambda sz: torch.zeros(sz, device="cuda", dtype=torch.int16)) if zero else (lambda sz: (torch.randn(sz, device="cuda") * 1000).half().view(torch.int16))
real = os.environ.get("REAL") == "1"
am = torch.zeros(64 * 32, device="cuda"); am[0] = 4.0
amb = am
if real and not pk:
    # operands as the step makes them: V2 = the split transform of a LeakyReLU'd Gaussian activation (scale from the a-priori bound
    # 100 max|x|), U2 = the transform of Gaussian 3x3 weights (scale from max|w|)
    x = torch.nn.functional.leaky_relu(torch.randn(8, res, res, k, device="cuda"), 0.2)
    am = torch.zeros(64 * 32, device="cuda"); L.call("absmax", x, x.numel(), am)
    a2 = torch.empty(M * k * 2, dtype=torch.int16, device="cuda")
    L.call("wino43_input_f16x2", x, a2, 8, res, res, k, am, 100.0)
    w = torch.randn(n, k, 3, 3, device="cuda") * 0.02
    amb = torch.zeros(64 * 32, device="cuda"); L.call("absmax", w, w.numel(), amb)
    b2 = torch.empty(36 * n * k * 2, dtype=torch.int16, device="cuda")
    L.call("wino43_weights", w, b2, n, k, 0, 2, amb)
    del x
else:
    a2, b2 = gen(M * k * 2), gen(36 * n * k * 2)
c = torch.empty(M, n, device="cuda", dtype=torch.float16 if pk else torch.float32)
cs = torch.zeros(64 * 32, device="cuda")
run = lambda: L.call(entry, a2, b2, c, M, n, k, t, n, am, 100.0, amb, *((cs,) if pk else ()))
samples, stop = [], False
def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "-P", "-c", "--showmemuse"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Power \(W\):\s*([\d.]+)", o); s = re.search(r"sclk clock level:.*\((\d+)Mhz\)", o)
            m = re.search(r"mclk clock level:.*\((\d+)Mhz\)", o)
            samples.append((float(p.group(1)) if p else -1, int(s.group(1)) if s else -1, int(m.group(1)) if m else -1))
        except Exception as ex:
            samples.append((-2, -2, -2))
for _ in range(5): run()
torch.cuda.synchronize()
th = threading.Thread(target=poll); th.start()
t0 = time.time(); it = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50): run()
    it += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
ms = e0.elapsed_time(e1) / it
good = [s for s in samples if s[0] > 0]
tail = good[len(good) // 2:] or [(-1, -1, -1)]
print("%s %s%s: %.3f ms/launch over %d launches | socket power W (second half of %d samples): mean %.0f max %.0f | sclk MHz mean %.0f min %d max %d | mclk %d"
      % (os.environ.get("TAG", ""), entry, " zeros" if zero else (" real" if real else ""), ms, it, len(good), sum(s[0] for s in tail) / len(tail), max(s[0] for s in tail),
         sum(s[1] for s in tail) / len(tail), min(s[1] for s in tail), max(s[1] for s in tail), tail[-1][2]))
