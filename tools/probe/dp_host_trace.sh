# host API calls and kernels of the last data-parallel steps on one time axis (rocprofv3 --kernel-trace --hip-trace)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dph; rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/dph -o t -- python $GRAFT_REPO_ROOT/tools/time_force_dp.py > /tmp/dph.log 2>&1
grep force_dp /tmp/dph.log
ls /tmp/dph
python - <<'PY'
import csv, glob
k = list(csv.DictReader(open(glob.glob("/tmp/dph/*kernel_trace.csv")[0])))
h = list(csv.DictReader(open(glob.glob("/tmp/dph/*hip_api_trace.csv")[0])))
k.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(k[-1]["End_Timestamp"])
t0 = t_end - 1_200_000          # the last 1.2 ms
ev = []
for r in k:
    s = int(r["Start_Timestamp"])
    if s >= t0:
        ev.append((s, "GPU  %-52s %6.1f" % (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:52], (int(r["End_Timestamp"]) - s) / 1e3)))
for r in h:
    s = int(r["Start_Timestamp"])
    if t0 - 600_000 <= s <= t_end and not r["Function"].startswith(("hipGetLastError", "hipPeekAtLastError", "hipGetDevice", "hipSetDevice", "__hip")):
        ev.append((s, "HOST %-40s %6.1f" % (r["Function"][:40], (int(r["End_Timestamp"]) - s) / 1e3)))
ev.sort()
for s, line in ev:
    print("%9.1f %s" % ((s - t0) / 1e3, line))
PY
