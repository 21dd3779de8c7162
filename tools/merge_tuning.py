"""Fold the selections of a partial TunableOp result file into the shipped one (same Validator header required):
    python tools/merge_tuning.py gpurun_out/tunableop_482.csv geometrics_amd/tuning/tunableop_gfx950.csv"""
import sys

new, dst = sys.argv[1], sys.argv[2]
head = lambda rows: [r for r in rows if r.startswith("Validator")]
body = lambda rows: [r for r in rows if r.strip() and not r.startswith("Validator")]
a, b = open(dst).read().splitlines(), open(new).read().splitlines()
if head(a) != head(b):
    raise SystemExit("library versions differ:\n%s\nvs\n%s" % ("\n".join(head(a)), "\n".join(head(b))))
key = lambda r: ",".join(r.split(",")[:2])
merged = {key(r): r for r in body(a)}
merged.update({key(r): r for r in body(b)})
open(dst, "w").write("\n".join(head(a) + sorted(merged.values())) + "\n")
print("merged: %d selections (+%d)" % (len(merged), len(merged) - len(body(a))))
