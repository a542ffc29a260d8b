"""Shader clock per kernel launch inside the step: GRBM_GUI_ACTIVE (busy cycles) / duration from one rocprofv3 PMC pass
(counter_collection.csv carries the dispatch's start / end timestamps).  Usage: python tools/pmc_clock.py <csv> [min_us]"""
import csv, sys, collections, re
rows = collections.defaultdict(list)
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") != "GRBM_GUI_ACTIVE":
        continue
    dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3   # us
    if dur < min_us:
        continue
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
    rows[k].append((dur, float(r["Counter_Value"])))
print("| kernel (launches >= %.0f us) | launches | mean us | busy cycles / us = MHz (mean) | MHz of the longest launch |" % min_us)
print("|---|---|---|---|---|")
for k, v in sorted(rows.items(), key=lambda kv: -sum(d for d, _ in kv[1]))[:40]:
    tot_d, tot_c = sum(d for d, _ in v), sum(c for _, c in v)
    big = max(v)
    print("| `%s` | %d | %.0f | %.0f | %.0f (%.0f us) |" % (k[:90], len(v), tot_d / len(v), tot_c / tot_d, big[1] / big[0], big[0]))
