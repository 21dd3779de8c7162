#!/bin/bash
# VALU / SALU / LDS instruction counts and wait cycles of the culled vs the brute-force Chamfer scan
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/cp; rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d /tmp/cp -o p -- python $R/tools/time_culled_nn.py > /dev/null 2>&1
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("/tmp/cp/p_counter_collection.csv")))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "nn_" not in k: continue
    if int(r["Grid_Size"]) < 700 * 512: continue        # the BASELINE-size launches only
    acc[("culled" if "culled" in k else "prep" if "prep" in k else "brute") + ("<fma>" if "<true>" in k else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
