#!/bin/bash
# Report of the culled Chamfer scan against the brute-force scan (run on the GPU box from the repo root):
#   bash tools/probe/cull_report.sh > gpurun_out/r03_culled_chamfer.txt
# needs tools/probe/libcullstats.so (chamfer_nn.hip built with -DNN_CULL_STATS, see tools/probe/run_probes.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
echo "# culled vs brute-force Chamfer scan, 8 meshes x (3000 gt points vs 3000 sampled points on the level-4 icosphere)"
echo "## correctness (tools/time_culled_nn.py: bitwise comparison with the brute-force scan, 2 orders x 2 arithmetics + adversarial inputs)"
rm -rf /tmp/c1; rocprofv3 --kernel-trace --output-format csv -d /tmp/c1 -o c1 -- python $R/tools/time_culled_nn.py > /tmp/c1.log 2>&1
echo "identical: $(grep -c identical /tmp/c1.log)   different: $(grep -c DIFFERENT /tmp/c1.log)"
echo "## kernel durations at this size (rocprofv3 --kernel-trace, launches of >= 700 workgroups), us: mean / min / max / launches"
python - <<'PY'
import csv, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/c1/c1_kernel_trace.csv")):
    k = r["Kernel_Name"]
    if "nn_" not in k: continue
    wg = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * int(r["Grid_Size_Y"])
    if "index" in k:
        if wg < 90: continue
    elif wg < 700: continue
    name = ("nn_cull_index_kernel (one cloud)" if "index" in k else "chamfer_nn_culled_kernel" if "culled" in k else "chamfer_nn_scalar_kernel (brute force)") + (" <fma>" if "<true>" in k else "")
    acc[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    print("%-48s %6.1f / %6.1f / %6.1f / %d" % (k, sum(v) / len(v), min(v), max(v), len(v)))
PY
echo "## counters per launch (rocprofv3 --pmc, same program)"
bash $R/tools/probe/cull_pmc.sh 2>/dev/null | grep -E "^brute|^culled"
echo "## what the levels of the test let through, per 64-query tile (of 187 runs of 16 targets; 8 seed runs are always evaluated)"
python $R/tools/probe/stats_tool.py 2>/dev/null | grep tiles
echo "## kernel time with parts switched off (ablation build), total ns of 101 launches"
bash $R/tools/probe/cull_knobs.sh
echo "## larger clouds (python wall time per call incl. the index of both clouds; orders given)"
python $R/tools/probe/cull_scale.py 2>/dev/null
