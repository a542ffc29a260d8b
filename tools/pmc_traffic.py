"""HBM traffic per launch of the GEMM kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over bench.py.
Usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/r02_pmc_traffic.json
FETCH_SIZE is doubled (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md); both counters are in KB."""
import csv, json, sys, collections, datetime

def load(path, counter):
    tot, cnt, per = collections.defaultdict(float), collections.defaultdict(int), collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            k = r["Kernel_Name"]
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
            per[k].append(float(r["Counter_Value"]))
    return tot, cnt, per

f, fc, fper = load(sys.argv[1], "FETCH_SIZE")
w, wc, wper = load(sys.argv[2], "WRITE_SIZE")
def targ(k, i):
    return k.split("<")[1].split(">")[0].split(",")[i].strip()

def pk(k, i):      # the PK (packed one-term, 16-bit storage mode) template flag
    a = k.split("<")[1].split(">")[0].split(",")
    return len(a) > i and a[i].strip() == "true"

groups = {"winograd_gemm_f16x2": lambda k: ("gemm3a_kernel" in k and targ(k, 4) == "2" and not pk(k, 7)) or "gemm_w4_kernel<false" in k,
          "winograd_gemm_f16_1term_packed": lambda k: ("gemm3a_kernel" in k and pk(k, 7)) or "gemm_w4_kernel<true" in k,
          "winograd_wgrad_f16_1term_packed": lambda k: "gemm3t_kernel" in k and pk(k, 8),
          "winograd_gemm_bf16x3": lambda k: "gemm3a_kernel" in k and targ(k, 4) == "3",
          "winograd_gemm_f16_1term": lambda k: "gemm3a_kernel" in k and targ(k, 4) == "1",
          "winograd_wgrad_f16x2": lambda k: "gemm3t_kernel" in k and targ(k, 5) == "2" and not pk(k, 8),
          "spade_modulate_fused": lambda k: "wino43_output_modulate" in k,
          "spade_fused": lambda k: "spade_fused_fwd_kernel" in k}
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
# the fused SPADE/SEAN forward at its largest shape (N = 8, 256 x 256, per-image tables): the launches of the same kernel run
# in the same order in both passes, so dispatch i of one pass pairs with dispatch i of the other
ks = [k for k in fper if "spade_fused_fwd_kernel" in k]
if ks:
    pairs = []
    for k in ks:
        for a_, b_ in zip(fper[k], wper.get(k, [])):
            pairs.append(2.0 * a_ * 1024 + b_ * 1024)
    pairs.sort()
    top = pairs[-max(1, len(pairs) // 8):]
    out["spade_fused"]["largest_launches_bytes"] = sum(top) / len(top)
    out["spade_fused"]["largest_launches_note"] = ("mean over the top eighth of the launches by bytes = the norms at 256 x 256 "
                                                   "(N = 8, C = 512, K = 160); SURVEY 8(d) algorithmic bytes: 3.49 GB")
print(json.dumps(out, indent=1))
