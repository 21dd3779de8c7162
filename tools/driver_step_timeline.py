"""One replay of the captured driver step (bench.driver_step_times) from its rocprofv3 kernel trace, aggregated per kernel:
   python tools/driver_step_timeline.py <trace dir> [out.txt]
A step = the launches between two consecutive adam_kernel GROUPS of the graph replays (the last complete one)."""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
# the Adam launches of a step are consecutive (168 tensors = 3 launches): a step ends with the last one of a run
ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] != i + 1]
a, b = ends[-2], ends[-1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", n)[:86]


agg = collections.OrderedDict()
total = 0.0
for r in rows[a + 1:b + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total += d
    k = short(r["Kernel_Name"])
    n, t = agg.get(k, (0, 0.0))
    agg[k] = (n + 1, t + d)
wall = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3
lines = ["# one replay of the captured driver step (bench.driver_step_times: batch 16, 482 vertices, three deformation blocks, three",
         "# poolings, three surface losses, regularisers, Adam) under rocprofv3 --kernel-trace: kernel, launches per step, us per step, share",
         "# %d launches, kernel time %.1f us, wall %.1f us" % (b - a, total, wall)]
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append("%-88s %4d %9.1f  %5.1f %%" % (k, n, t, 100 * t / total))
# the three largest items against their roofs (batch 16 x 482 vertices = 7712 rows; peaks: 157.3 TFLOP/s fp32 MFMA, 8 TB/s HBM)
def per(prefix):
    hit = [(n, t) for k, (n, t) in agg.items() if k.startswith(prefix)]
    n, t = sum(h[0] for h in hit), sum(h[1] for h in hit)
    return n, (t / n if n else 0.0)


rows_, maps_bytes = 16 * 482, 16 * (64 * 56 * 56 + 128 * 28 * 28 + 256 * 14 * 14 + 512 * 7 * 7) * 4
lines.append("#")
lines.append("# against the roofs (algorithmic work / average launch; 7712 rows):")
n, us = per("Cijk_Ailk_Bljk_SB_MT64x32x32")
if n:
    fl = 2.0 * rows_ * 192 * 192
    lines.append("#   hidden-layer forward product 7712 x 192 x 192 (library, %d launches): %.1f us = %.1f TFLOP/s = %.2f of the fp32 MFMA peak"
                 % (n, us, fl / us / 1e6, fl / us / 1e6 / 157.3))
n, us = per("zn_aggregate_ell_kernel<0, false")
if n:
    by = rows_ * 192 * 4 * 2 + 482 * 8 * 8
    lines.append("#   zn_aggregate_ell_kernel forward (%d launches): %.1f us for %.1f MB = %.2f TB/s = %.2f of HBM (a latency chain at 7712 rows: "
                 "two dependent gather round trips + launch + drain)" % (n, us, by / 1e6, by / us / 1e6, by / us / 1e6 / 8.0))
n, us = per("db_fwd_kernel<true>")
if n:
    fl, by = 2.0 * rows_ * 192 * 192, rows_ * 192 * 4 * 4 + 192 * 192 * 4
    lines.append("#   db_fwd_kernel<true> (aggregation + BatchNorm1d(verts) + ReLU + residual + next product in one launch, %d launches): %.1f us; "
                 "the product alone = %.1f TFLOP/s = %.2f of the fp32 MFMA peak; 23.7 MB of tensors (S in, Z / X' / S' out) = %.2f TB/s"
                 % (n, us, fl / us / 1e6, fl / us / 1e6 / 157.3, by / us / 1e6))
n, us = per("db_bwd_kernel<true>")
if n:
    fl = 2.0 * rows_ * 192 * 192
    lines.append("#   db_bwd_kernel<true> (aggregation backward + input-gradient product + BatchNorm backward, %d launches): %.1f us; the "
                 "product alone = %.1f TFLOP/s = %.2f of the fp32 MFMA peak" % (n, us, fl / us / 1e6, fl / us / 1e6 / 157.3))
n, us = per("db_fwd_chain_kernel")
if n:
    fl, by = 12 * 2.0 * rows_ * 192 * 192, 13 * rows_ * 192 * 4 * 3 + 12 * 192 * 192 * 4
    lines.append("#   db_fwd_chain_kernel (the 13 hidden layers of a block forward -- aggregation + BatchNorm1d(verts) + ReLU + residual + next "
                 "product each -- in ONE launch, %d launches): %.1f us = %.1f per layer; the 12 products alone = %.1f TFLOP/s = %.2f of the "
                 "fp32 MFMA peak; %.0f MB of tensors = %.2f TB/s -- a chain of 13 neighbour hand-offs, not a throughput kernel"
                 % (n, us, us / 13, fl / us / 1e6, fl / us / 1e6 / 157.3, by / 1e6, by / us / 1e6))
n, us = per("db_bwd_chain_kernel")
if n:
    fl = 12 * 2.0 * rows_ * 192 * 192
    lines.append("#   db_bwd_chain_kernel (the 13 backward layers of a block in ONE launch, %d launches): %.1f us = %.1f per layer; the 12 "
                 "input-gradient products alone = %.1f TFLOP/s = %.2f of the fp32 MFMA peak" % (n, us, us / 13, fl / us / 1e6, fl / us / 1e6 / 157.3))
n, us = per("pool_bwd_lists_kernel")
if n:
    by = maps_bytes + rows_ * 960 * 4
    lines.append("#   pool_bwd_lists_kernel (%d launches; texel lists beside the vertex tiles' sums): %.1f us for %.1f MB (the four maps + the "
                 "upstream gradient, each once) = %.2f TB/s = %.2f of HBM -- VALU issue and gather latency, not bytes (LAB_NOTES 12.3)"
                 % (n, us, by / 1e6, by / us / 1e6, by / us / 1e6 / 8.0))
n, us = per("pool_bwd_grads_kernel")
if n:
    by = maps_bytes + rows_ * 960 * 4
    lines.append("#   pool_bwd_grads_kernel (%d launches; map gradient from the lists beside the vertex gradient): %.1f us for %.1f MB (the "
                 "gradient rows in, the four maps' gradients out) = %.2f TB/s = %.2f of HBM"
                 % (n, us, by / 1e6, by / us / 1e6, by / us / 1e6 / 8.0))
text = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text)
