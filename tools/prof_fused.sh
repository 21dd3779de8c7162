cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_fused -- python $R/tools/profile_fused.py > $R/gpurun_out/pmc_fused.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_fused2 -- python $R/tools/profile_fused.py >> $R/gpurun_out/pmc_fused.log 2>&1
python - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
t = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_fused", "pmc_fused2"):
    for f in glob.glob(os.path.join(R, "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "zs_layer" in r["Kernel_Name"]:
                t[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in t.items():
    print(k)
    for n, v in sorted(c.items()):
        print("   %-28s %14.0f" % (n, sum(v) / len(v)))
PY
