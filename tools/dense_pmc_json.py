"""Fold the two rocprofv3 --pmc passes of tools/prof_dense.sh (eager launches of the matrix-core products at the BASELINE
shard, tools/profile_dense.py) into one tracked table:   python tools/dense_pmc_json.py gpurun_out profiles/r04_dense_pmc.json
Per kernel: mean counter values per launch + the ratios that are read off them (MFMA pipe busy, waves waiting on an
instruction, LDS bank conflicts per LDS instruction cycle).  Stamped with the kernel-source digest like the step's table."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", name).strip()


def load(path, table):
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not (k.startswith("dense_") or k.startswith("Cijk")):
                continue
            table[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            table[k[:70]]["_vgpr"] = [int(r["VGPR_Count"])]
            table[k[:70]]["_lds"] = [int(r["LDS_Block_Size"])]
            table[k[:70]]["_grid"] = [int(r["Grid_Size"])]


def main(root, dst):
    raw = collections.defaultdict(lambda: collections.defaultdict(list))
    load(os.path.join(root, "pmc_dense"), raw)
    load(os.path.join(root, "pmc_dense2"), raw)
    out = {}
    for k, counters in sorted(raw.items()):
        row = {c + "_per_launch": round(sum(v) / len(v), 1) for c, v in counters.items() if not c.startswith("_")}
        row.update(vgpr=counters["_vgpr"][0], lds_bytes=counters["_lds"][0], grid=counters["_grid"][0])
        get = lambda c: row.get(c + "_per_launch")
        if get("SQ_VALU_MFMA_BUSY_CYCLES") and get("SQ_BUSY_CYCLES"):
            row["mfma_busy_over_sq_busy"] = round(get("SQ_VALU_MFMA_BUSY_CYCLES") / get("SQ_BUSY_CYCLES"), 4)
        if get("SQ_WAIT_INST_ANY") and get("SQ_WAVE_CYCLES"):
            row["wait_inst_over_wave_cycles"] = round(get("SQ_WAIT_INST_ANY") / get("SQ_WAVE_CYCLES"), 4)
        if get("SQ_LDS_BANK_CONFLICT") is not None and get("SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict_over_lds_active"] = round(get("SQ_LDS_BANK_CONFLICT") / get("SQ_LDS_IDX_ACTIVE"), 5)
        out[k] = row
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from geometrics_amd import build as hip_build
    out["_meta"] = {"source_sha256": hip_build.source_digest(), "shapes": "rows 20496 (8 meshes x 2562), cin 963 and 192, c 192",
                    "command": "bash tools/prof_dense.sh && python tools/dense_pmc_json.py gpurun_out <this file>",
                    "note": "two separate --pmc passes (kernel trace only); SQ counters are summed over the chip's shader engines"}
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", dst, len(out) - 1, "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
