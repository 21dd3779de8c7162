cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dpt; rocprofv3 --kernel-trace --output-format csv -d /tmp/dpt -o t -- python $GRAFT_REPO_ROOT/tools/time_force_dp.py > /tmp/dpt.log 2>&1
grep force_dp /tmp/dpt.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open("/tmp/dpt/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 60 kernels = the tail of the force_dp=True run
tail = rows[-44:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    print("%8.1f %6.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:70]))
PY
