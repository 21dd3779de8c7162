"""rocprofv3 bench_kernel_stats.csv -> the compact form kept under profiles/: `(anonymous namespace)::` and `void `
dropped, argument lists dropped, library GEMM names truncated to 60 characters."""
import csv
import re
import sys

rows = list(csv.reader(open(sys.argv[1])))
out = csv.writer(open(sys.argv[2], "w", newline=""))
out.writerow(rows[0])
for r in rows[1:]:
    name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name).strip()
    r[0] = name[:60]
    out.writerow(r)
