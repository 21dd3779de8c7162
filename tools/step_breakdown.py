"""Dev helper: per-step kernel breakdown from a rocprofv3 kernel trace CSV of bench.py."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'tri_prep_kernel' in r['Kernel_Name']]
lo, hi = idx[20], idx[40]
win = rows[lo:hi]
print("wall per step us", (int(rows[hi]['Start_Timestamp']) - int(win[0]['Start_Timestamp'])) / 20 / 1e3)
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')[:74]
    agg[n][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); agg[n][1] += 1
print("kernel-time per step us", sum(v[0] for v in agg.values()) / 20 / 1e3, "launches/step", sum(v[1] for v in agg.values()) / 20)
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{n:76s} {v[1]/20:5.1f}x {v[0]/20/1e3:8.1f} us/step")
