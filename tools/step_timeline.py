"""One step of `python bench.py --steps N --warmup W --no-cpu-baseline --steps-only` in launch order, from the rocprofv3
kernel trace of that command:   python tools/step_timeline.py <trace dir> [out.txt]
A step = the launches between two consecutive step-closing dispatches of the graph replay (the last complete one)."""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last launch of a step: the stand-alone Adam launch, or -- when the optimiser step rides in the end-of-pass reduction
# (single-process step since round 3) -- that reduction launch
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
if len(adam) < 2:
    adam = [i for i, r in enumerate(rows) if "dense_reduce_kernel" in r["Kernel_Name"]]
a, b = adam[-2], adam[-1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*$", "", n)
    return n[:74]


lines = ["# one step of `python bench.py --steps 100 --warmup 10 --no-cpu-baseline --steps-only` under rocprofv3 --kernel-trace",
         "# (HIP-graph replay, 8 meshes): kernel, duration (us)"]
total = 0.0
for r in rows[a + 1:b + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total += d
    lines.append("%-76s %6.1f" % (short(r["Kernel_Name"]), d))
wall = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3
lines.append("# %d launches, kernel time %.1f us, wall %.1f us" % (b - a, total, wall))
text = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text)
