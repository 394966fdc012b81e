#!/usr/bin/env python3
"""Summarise the two rocprofv3 PMC passes of tools/gpu_pmc_mfma.sh into profiles/pmc_mfma.json (matrix-pipe utilisation of k_mmq_mfma).

    tools/pmc_mfma_summary.py <dir with pmc_mfma/ and pmc_mfma2/> <key> <stamp> [out.json]

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x elapsed cycles), elapsed = GRBM_GUI_ACTIVE summed over the 8 XCDs / 8;
valu_insts_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA. Averages per k_mmq_mfma launch. The stamp is bench.prefill_source_stamp() of the
build the passes ran on: bench.py refuses a quote taken on another build of prefill.hip."""
import csv, glob, json, os, sys
from collections import defaultdict


KERNEL = "k_mmq_fast"   # the timed default of sequence mode (prefill_fast.hip); pass a fifth argument for another kernel name


def collect(d):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    global KERNEL
    root, key, stamp = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_mfma.json"
    if len(sys.argv) > 5:
        KERNEL = sys.argv[5]
    a, b = collect(os.path.join(root, "pmc_mfma")), collect(os.path.join(root, "pmc_mfma2"))
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in a or "GRBM_GUI_ACTIVE" not in b or "SQ_INSTS_MFMA" not in b:
        sys.exit("counters missing: " + str(sorted(a)) + " / " + str(sorted(b)))
    busy, valu = a["SQ_VALU_MFMA_BUSY_CYCLES"][0], a["SQ_INSTS_VALU"][0]
    grbm, mfma = b["GRBM_GUI_ACTIVE"][0], b["SQ_INSTS_MFMA"][0]
    e = {"kernel": KERNEL, "launches_sampled": a["SQ_VALU_MFMA_BUSY_CYCLES"][1], "SQ_INSTS_MFMA_per_launch": mfma,
         "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": busy, "SQ_INSTS_VALU_per_launch": valu, "SQ_WAVE_CYCLES_per_launch": a.get("SQ_WAVE_CYCLES", (0, 0))[0],
         "GRBM_GUI_ACTIVE_per_launch_sum_over_8_XCDs": grbm, "mfma_util": busy / (1024.0 * grbm / 8.0), "valu_insts_per_mfma": valu / mfma,
         "prefill_source_stamp": stamp,
         "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU / --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 "
                   "SQ_INSTS_MFMA (two passes, tools/gpu_pmc_mfma.sh), averages per launch of that kernel over the 1024-token pass"}
    d = json.load(open(out)) if os.path.exists(out) else {}
    d[key] = e
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(e))


if __name__ == "__main__":
    main()
