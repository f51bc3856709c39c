#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (separate passes) into per-kernel HBM bytes per
launch.  usage: pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters
are in KiB, and FETCH_SIZE counts wide coalesced reads at half their bytes on gfx950 (x2 column)."""
import csv
import glob
import json
import os
import sys

KERNELS = {"hwgrad": "hwgrad_kernel", "hconv_s2": "hconv_s2_kernel", "hconv": "hconv_kernel", "igemm_s1": "wino_kernel", "wgrad": "wino_wgrad_kernel", "igemm_s2": "igemm_kernel<2,",
           "igemm_bwd_s2": "igemm_kernel<1, 0,", "wgrad_direct": "::wgrad_kernel<"}


def by_name(d, counter):
    """per exact kernel name (template arguments kept, argument lists cut): Counter_Value sum and distinct dispatches"""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            s = out.setdefault(name, {"sum": 0.0, "ids": set()})
            s["sum"] += float(row["Counter_Value"])
            s["ids"].add(row.get("Dispatch_Id"))
    return out


def collect(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            for key, pat in KERNELS.items():
                if pat in name:
                    s = out.setdefault(key, {"sum": 0.0, "ids": set()})
                    s["sum"] += float(row["Counter_Value"])
                    s["ids"].add(row.get("Dispatch_Id"))
    return out


def main():
    fd, wd, dst = sys.argv[1:4]
    f, w = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    res = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1` "
                    "(2 steps); counter unit KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their bytes "
                    "(MI355X_MICROARCH.md, HBM): x2 column"}
    for k in f:
        n = len(f[k]["ids"])
        res[k] = {"launches_profiled": n, "FETCH_SIZE_KiB_sum": f[k]["sum"], "WRITE_SIZE_KiB_sum": w.get(k, {}).get("sum", 0.0),
                  "fetch_bytes_per_launch_raw": f[k]["sum"] * 1024 / n,
                  "fetch_bytes_per_launch_x2_gfx950": 2 * f[k]["sum"] * 1024 / n,
                  "write_bytes_per_launch": w.get(k, {}).get("sum", 0.0) * 1024 / n}
    # the same per exact kernel name (so that e.g. hwgrad_kernel is not averaged with its reduce kernel), largest first
    fn, wn = by_name(fd, "FETCH_SIZE"), by_name(wd, "WRITE_SIZE")
    names = sorted(fn, key=lambda k: -fn[k]["sum"])[:24]
    res["by_kernel_name"] = {k: {"launches_profiled": len(fn[k]["ids"]),
                                 "fetch_GB_per_launch_x2_gfx950": 2 * fn[k]["sum"] * 1024 / len(fn[k]["ids"]) / 1e9,
                                 "write_GB_per_launch": wn.get(k, {}).get("sum", 0.0) * 1024 / max(1, len(wn.get(k, {}).get("ids", []))) / 1e9,
                                 "fetch_GB_total_x2": 2 * fn[k]["sum"] * 1024 / 1e9} for k in names}
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
