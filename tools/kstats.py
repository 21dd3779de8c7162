"""Dev helper: per-kernel average of the LAST n dispatches from a rocprofv3 kernel_trace.csv."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 5
agg = collections.defaultdict(list)
for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    agg[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1][-last:]) / min(len(kv[1]), last)):
    print(f"{n:66s} n={len(v):4d} avg(last {last}) {sum(v[-last:]) / min(len(v), last):8.1f} us")
