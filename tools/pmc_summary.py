"""Mean per-dispatch PMC values per kernel from rocprofv3 counter_collection.csv files (any number of passes).
Usage: python tools/pmc_summary.py <csv> [<csv> ...]"""
import csv, os, sys, collections, re
MATCH = os.environ.get("PMC_KERNEL", "gemm3")      # substring of the kernel names to report
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
        if MATCH not in k:
            continue
        key = (k, r["Counter_Name"])
        tot[key] += float(r["Counter_Value"]); cnt[key] += 1
print("| kernel | counter | mean per launch | launches |\n|---|---|---|---|")
for (k, c) in sorted(tot):
    print("| `%s` | %s | %.4g | %d |" % (k, c, tot[(k, c)] / cnt[(k, c)], cnt[(k, c)]))
