#!/usr/bin/env python3
"""profiles/pmc_match_mfma.json from a counter pass over the matcher's MFMA kernel:
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA --output-format csv \
            -d gpurun_out/pmc_mfma -- python3 tools/prof_stereo.py 6
  python tools/pmc_match.py"""
import csv, glob, json, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_mfma", "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_match_mfma" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
n = len(next(iter(acc.values())))
res = {"kernel": "uvo::k_match_mfma (v_mfma_f32_32x32x16_bf16 on bf16 hi/lo splits: shortlist of the brute-force matcher, fixed grid of 768 workgroups walking the tiles), "
                 "C3 stereo pair, both calls averaged",
       "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA -- python3 tools/prof_stereo.py 6",
       "launches_sampled": n, "counters_avg_per_launch": avg}
if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CU_CYCLES" in avg:
    res["mfma_busy_over_cu_busy_x4_simd"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * avg["SQ_BUSY_CU_CYCLES"])
    res["note"] = ("SQ_INSTS_MFMA x 32 cycles = the MFMA-busy figure; the ratio is the share of a busy CU's four matrix pipes that is in use; "
                   "the end-to-end figure against the 157.3 TFLOP/s f32 MFMA peak is bench.py's roofline_match")
json.dump(res, open(os.path.join(ROOT, "profiles", "pmc_match_mfma.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
