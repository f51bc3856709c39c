#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc SQ counter CSVs (one or more passes) per kernel family.
usage: pmc_mix.py OUT.json DIR [DIR ...]
Derived: matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), instructions per MFMA,
LDS bank-conflict share, wave-time split (SQ_WAIT_ANY: parked at s_waitcnt / barrier; SQ_WAIT_INST_ANY: issue stall;
SQ_ACTIVE_INST_ANY: issuing) -- MI355X_MICROARCH.md, rocprofv3 PMC slots."""
import csv
import glob
import json
import os
import sys

# substrings of the (demangled) kernel names in rocprofv3's counter CSV; first match wins
FAMILIES = {"hwgrad_kernel": "hwgrad_kernel", "hwgrad_s2_kernel": "hwgrad_s2_kernel", "hwgrad_fc_kernel": "hwgrad_fc_kernel",
            "hconv_s2_fwd_kernel": "hconv_s2_fwd_kernel", "hconv_s2_bwd_pair_kernel": "hconv_s2_bwd_pair_kernel", "hconv_s2_kernel": "hconv_s2_kernel", "hconv_kernel": "hconv_kernel", "wino_wgrad_kernel": "wino_wgrad_kernel", "wino_kernel": "wino_kernel", "igemm_s2_fwd": "igemm_kernel<2,",
            "igemm_s2_bwd": "igemm_kernel<1, 0,", "igemm_s1": "igemm_kernel<1, -1,", "wgrad_direct": "::wgrad_kernel<"}
csv.field_size_limit(1 << 30)


def main():
    dst, dirs = sys.argv[1], sys.argv[2:]
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"]
                for fam, pat in FAMILIES.items():
                    if pat in name:
                        a = acc.setdefault(fam, {"_ids": {}})
                        a[row["Counter_Name"]] = a.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                        a["_ids"].setdefault(row["Counter_Name"], set()).add(row["Dispatch_Id"])
                        break
    out = {}
    for fam, a in acc.items():
        ids = a.pop("_ids")
        r = dict(a)
        r["launches"] = max(len(v) for v in ids.values())
        g = r.get
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CU_CYCLES"):
            r["matrix_pipe_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * g("SQ_BUSY_CU_CYCLES"))
        if g("SQ_INSTS_MFMA"):
            for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"):
                if g(k):
                    r[k.lower().replace("sq_insts_", "") + "_per_mfma"] = g(k) / g("SQ_INSTS_MFMA")
        if g("SQ_LDS_BANK_CONFLICT") and g("SQ_LDS_IDX_ACTIVE"):
            r["lds_conflict_share"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
        if g("SQ_WAVE_CYCLES"):
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if g(k):
                    r[k.lower() + "_share_of_wave_cycles"] = g(k) / g("SQ_WAVE_CYCLES")
        out[fam] = r
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if not kk.startswith("SQ_")} for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
