"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per launch and per wave)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
keys = sys.argv[2:] or ["tri_scan_grouped", "tri_scan_ws", "chamfer_nn"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    for key in keys:
        if key in r["Kernel_Name"]:
            agg[key + "/" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    n = len(next(iter(v.values())))
    w = sum(v.get("SQ_WAVES", [1])) / n
    print(k, "launches", n, "VGPR", rows[0].get("VGPR_Count"))
    for c, vals in sorted(v.items()):
        m = sum(vals) / len(vals)
        print("   %-22s %14.0f   per wave %10.1f" % (c, m, m / w))
