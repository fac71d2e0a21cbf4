"""rocprofv3 kernel trace of bw_prof_target.py -> achieved GB/s per kernel (device time from the trace, algorithmic
bytes from SURVEY 8(d)).   python bw_prof_summary.py <kernel_trace.csv> <order.json> <out.json>"""
import csv
import json
import sys

trace, order, out = sys.argv[1:4]
rows = sorted((r for r in csv.DictReader(open(trace))), key=lambda r: int(r["Start_Timestamp"]))
ours = [r for r in rows if not r["Kernel_Name"].startswith("void at::")]
res, pos = [], 0
for item in json.load(open(order)):
    grp = ours[pos:pos + item["launches"]]
    pos += item["launches"]
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp][3:]  # drop the first three launches
    avg = sum(durs) / len(durs)
    res.append({"kernel": item["tag"], "device_kernel": grp[0]["Kernel_Name"][:90], "avg_us": round(avg / 1e3, 2),
                "algorithmic_MB": round(item["bytes"] / 1e6, 1), "achieved_GBps": round(item["bytes"] / avg, 1),
                "frac_of_8TBps": round(item["bytes"] / avg / 8000.0, 3)})
json.dump(res, open(out, "w"), indent=1)
for r in res:
    print("%-58s %8.2f us %8.1f GB/s (%.0f%% of 8 TB/s)" % (r["kernel"], r["avg_us"], r["achieved_GBps"], 100 * r["frac_of_8TBps"]))
