cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_fused -- python $R/tools/time_fused_layer.py > $R/gpurun_out/kt_fused.log 2>&1
python - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
f=glob.glob(os.path.join(R,"gpurun_out","kt_fused","**","*kernel_trace.csv"),recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    v=sorted(v); print("%-72s n=%4d median %7.2f us  p10 %7.2f" % (k, len(v), v[len(v)//2], v[len(v)//10]))
PY
