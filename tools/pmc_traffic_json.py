"""Fold the three rocprofv3 PMC passes of tools/pmc_traffic.sh into one per-kernel table (JSON).

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; the per-launch mean of the LAST 5 dispatches of every
kernel is kept (the first ones include cold caches and the GEMM tuner's probes).  SQ counters: mean over all
dispatches of the kernel.  Kernel names: `(anonymous namespace)::` dropped, arguments dropped."""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", name).strip()


def load(path):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(os.path.join(path, "p_counter_collection.csv"))):
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]),
                   "workgroup": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"])}
    return per, meta


def main(src, dst):
    table = {}
    fetch, meta = load(os.path.join(src, "fetch"))
    write, _ = load(os.path.join(src, "write"))
    valu, _ = load(os.path.join(src, "valu"))
    for k in sorted(fetch):
        if not any(tag in k for tag in ("_kernel", "Cijk")):
            continue
        f = fetch[k]["FETCH_SIZE"][-5:]
        w = write.get(k, {}).get("WRITE_SIZE", [0.0])[-5:]
        row = {"launches": len(fetch[k]["FETCH_SIZE"]), "FETCH_SIZE_KB_per_launch": round(sum(f) / len(f), 1),
               "WRITE_SIZE_KB_per_launch": round(sum(w) / len(w), 1)}
        row.update(meta[k])
        for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU",
                  "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"):
            vals = valu.get(k, {}).get(c)
            if vals:
                row[c + "_per_launch"] = round(sum(vals) / len(vals), 1)
        table[k[:60]] = row
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from geometrics_amd import build as hip_build
    table["_meta"] = {"source_sha256": hip_build.source_digest(),
                      "note": "counters are valid for exactly these kernel sources (geometrics_amd.build.source_digest); "
                              "bench.py ignores the file when the digest differs",
                      "command": "bash tools/pmc_traffic.sh && python tools/pmc_traffic_json.py gpurun_out/pmc <this file>"}
    with open(dst, "w") as fh:
        json.dump(table, fh, indent=1, sort_keys=True)
    print("wrote", dst, len(table), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
