"""rocprofv3 counter_collection CSVs of the driver step (FETCH_SIZE pass, WRITE_SIZE pass) -> per kernel name: launches seen,
average fetched / written MB per launch (the counters are in KB as rocprofv3 reports them for gfx950; see
/opt/skills/guides/MI355X_MICROARCH.md for the caveats -- 16 B/lane reads are counted once here, not doubled)."""
import csv, glob, os, sys
from collections import defaultdict


def load(root, counter):
    acc, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(")[0][:90]
            acc[name] += float(r["Counter_Value"])
            n[name] += 1
    return {k: (acc[k] / n[k], n[k]) for k in acc}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0))[0] + write.get(k, (0, 0))[0]) * max(fetch.get(k, (0, 1))[1], 1)):
    f, nf = fetch.get(k, (0.0, 0))
    w, _ = write.get(k, (0.0, 0))
    rows.append("%-92s %6d launches  fetch %9.2f MB  write %9.2f MB per launch" % (k, nf, f / 1024.0, w / 1024.0))
text = "# HBM-side traffic per launch of the driver step's kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, KB as reported / 1024)\n" + "\n".join(rows) + "\n"
open(sys.argv[3], "w").write(text)
print(text[:4500])
