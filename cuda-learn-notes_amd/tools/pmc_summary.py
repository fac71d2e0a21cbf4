"""Summarise rocprofv3 PMC passes (one directory per pass) into a small JSON for profiles/.
    python pmc_summary.py <kernel-substring> <out.json> <pass_dir>...
Per kernel whose name contains the substring: mean of every counter over its dispatches (first dispatch
dropped as warm-up when there are >2). L2<->fabric traffic per launch (an UPPER BOUND of the HBM bytes: Infinity-Cache
hits are inside FETCH_SIZE, MI355X_MICROARCH.md §HBM / §Infinity Cache; rounds 1-3 wrote these keys as hbm_*):
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


def dynamic_lds(kernel_name):
    """Dynamic LDS bytes the launcher asks for (csrc/*: Cfg::LDS_BYTES / GeoSplit::LDS_BYTES / Geo::LDS_BYTES)."""
    table = (("hgemm_pp_kernel", 2 * (256 + 256) * 64 * 2), ("hgemm_w4_kernel", 2 * (256 + 256) * 64 * 2), ("hgemm_pp32_kernel", 4 * (256 + 256) * 32 * 2),
             ("fa2_fwd_dsplit_kernelILi64E", 2 * 2 * 128 * 128), ("fa2_fwd_dsplit_kernelILi128E", 8 * 32 * (256 + 16)),
             ("fa2_fwd_m16_pair_kernelILi2ELb0E", 8 * 32 * (512 + 16)), ("fa2_fwd_m16_pair_kernelILi2ELb1E", 2 * 2 * 32 * 1024 + 8 * 4096),
             ("fa2_fwd_pair2_kernel", 2 * 2 * 32 * 1024 + 4 * 4096 + 4 * 128),
             ("fa2_fwd_m16_kernelILi64E", 2 * 2 * 128 * 128), ("fa2_fwd_m16_kernelILi128E", 2 * 2 * 128 * 256),
             ("fa2_fwd_m16x_kernelILi64ELi32E", 2 * 2 * 128 * 128), ("fa2_fwd_m16x_kernelILi64ELi64E", 8 * 64 * (128 + 16)), ("fa2_fwd_m16x_kernelILi128E", 2 * 2 * 128 * 256),
             ("fa2_fwd_dring_kernelILi1024E", 4 * 16 * 2048 + 8 * 2048), ("fa2_fwd_dring_kernelILi768E", 4 * 16 * 1536 + 8 * 2048), ("fa2_fwd_dring_kernelILi640E", 4 * 16 * 1280 + 8 * 2048),
             ("fa2_fwd_dsplit_kernelILi256E", 8 * 32 * (512 + 16)), ("fa2_fwd_dsplit_kernelILi512E", 2 * 2 * 32 * 1024 + 8 * 4096))
    for sub, b in table:
        if sub in kernel_name:
            return b
    return None


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
                # rocprofv3 on gfx950 reports VGPR_Count in allocation units of 2 registers (108 for a 213-register
                # kernel) and LDS_Block_Size WITHOUT the dynamic segment (0 for kernels whose LDS is all dynamic): the
                # launch-side numbers are added from the table below (csrc Geo structs), the compiler's own from
                # tools/kernel_resources.py
                meta[key] = {"vgpr_rocprof_units_of_2": int(r["VGPR_Count"]), "vgpr_allocated": 2 * int(r["VGPR_Count"]),
                             "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                             "lds_static_bytes": int(r["LDS_Block_Size"]), "lds_dynamic_bytes": dynamic_lds(n),
                             "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"])}
    res = {}
    for k, cs in agg.items():
        m = {}
        for c, v in cs.items():
            v = v[1:] if len(v) > 2 else v
            m[c] = sum(v) / len(v)
        e = {"counters": m, "meta": meta[k]}
        if "FETCH_SIZE" in m:
            e["l2_fabric_read_bytes_per_launch"] = 2.0 * m["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in m:
            e["l2_fabric_write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["l2_fabric_traffic_bytes_per_launch"] = e["l2_fabric_read_bytes_per_launch"] + e["l2_fabric_write_bytes_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            # MFMA_BUSY is summed over all SIMDs (1024), GUI_ACTIVE over the 8 XCDs
            e["mfma_busy_frac"] = (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (m["GRBM_GUI_ACTIVE"] / 8.0)
        res[k] = e
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
