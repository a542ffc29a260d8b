"""HBM traffic per launch of the GEMM kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over bench.py.
Usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/r02_pmc_traffic.json
FETCH_SIZE is doubled (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md); both counters are in KB."""
import csv, json, sys, collections, datetime

def load(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            k = r["Kernel_Name"]
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt

f, fc = load(sys.argv[1], "FETCH_SIZE")
w, wc = load(sys.argv[2], "WRITE_SIZE")
def targ(k, i):
    return k.split("<")[1].split(">")[0].split(",")[i].strip()

groups = {"winograd_gemm_f16x2": lambda k: "gemm3a_kernel" in k and targ(k, 4) == "2",
          "winograd_gemm_bf16x3": lambda k: "gemm3a_kernel" in k and targ(k, 4) == "3",
          "winograd_gemm_f16_1term": lambda k: "gemm3a_kernel" in k and targ(k, 4) == "1",
          "winograd_wgrad_f16x2": lambda k: "gemm3t_kernel" in k and targ(k, 5) == "2",
          "spade_modulate_fused": lambda k: "wino43_output_modulate" in k}
out = {}
for name, pred in groups.items():
    ks = [k for k in f if pred(k)]
    if not ks:
        continue
    n = sum(fc[k] for k in ks)
    fetch = 2.0 * sum(f[k] for k in ks) * 1024 / n
    write = sum(w.get(k, 0.0) for k in ks) * 1024 / max(1, sum(wc.get(k, 0) for k in ks))
    out[name] = {"bytes_per_launch": fetch + write, "fetch_bytes_per_launch_x2_corrected": fetch,
                 "write_bytes_per_launch": write, "launches_in_trace": n,
                 "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) over `bench.py --steps 2 "
                           "--warmup 1 --no-cpu-baseline --no-f32-run`, %s" % datetime.date.today().isoformat()}
print(json.dumps(out, indent=1))
