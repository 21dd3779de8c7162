"""One replay of the captured driver step from its rocprofv3 kernel trace, launch by launch IN ORDER (name, grid, us):
   python tools/driver_step_sequence.py <trace dir> [out.txt]    -- to see which eager torch launches sit between which kernels."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] != i + 1]
a, b = ends[-2], ends[-1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    return n[:150]


out = []
for i, r in enumerate(rows[a + 1:b + 1]):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    grid = "%sx%sx%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
    out.append("%4d %7.1f us  grid %-14s %s" % (i, d, grid, short(r["Kernel_Name"])))
text = "\n".join(out) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
else:
    print(text)
