"""Summarise rocprofv3 PMC passes (one directory per pass) into a small JSON for profiles/.
    python pmc_summary.py <kernel-substring> <out.json> <pass_dir>...
Per kernel whose name contains the substring: mean of every counter over its dispatches (first dispatch
dropped as warm-up when there are >2). HBM traffic per launch follows MI355X_MICROARCH.md §HBM:
  bytes = 2 * FETCH_SIZE * 1024  (gfx950: FETCH_SIZE reads exactly half of a wide coalesced stream;
                                  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB)
        + WRITE_SIZE * 1024      (uncalibrated on gfx950; reported separately as well)
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    sub, out = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for d in sys.argv[3:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"]
                if sub not in n:
                    continue
                key = n[:120]
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[key] = {"vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]),
                             "sgpr": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                             "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"])}
    res = {}
    for k, cs in agg.items():
        m = {}
        for c, v in cs.items():
            v = v[1:] if len(v) > 2 else v
            m[c] = sum(v) / len(v)
        e = {"counters": m, "meta": meta[k]}
        if "FETCH_SIZE" in m:
            e["hbm_read_bytes_per_launch"] = 2.0 * m["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in m:
            e["hbm_write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["hbm_traffic_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            # MFMA_BUSY is summed over all SIMDs (1024), GUI_ACTIVE over the 8 XCDs
            e["mfma_busy_frac"] = (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (m["GRBM_GUI_ACTIVE"] / 8.0)
        res[k] = e
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
