"""rocprofv3 kernel trace of fa_trace_target.py -> one row per shape: the traced kernel, the number of timed launches, their
average / min device duration and 4*B*H*N^2*D / avg.   python fa_trace_summary.py <kernel_trace.csv> <order.json> <out.csv>"""
import csv
import json
import sys

trace, order, out = sys.argv[1:4]
rows = sorted((r for r in csv.DictReader(open(trace)) if "fa2" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
pos = 0
with open(out, "w") as f:
    f.write("shape,kernel,calls,avg_us,min_us,tflops_4bhn2d\n")
    for item in json.load(open(order)):
        n = item["warm"] + item["launches"]
        grp = rows[pos:pos + n][item["warm"]:]
        pos += n
        assert len(grp) == item["launches"] and len({r["Kernel_Name"] for r in grp}) == 1, (item, len(grp))
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp]
        B, H, N, D = item["shape"]
        avg = sum(durs) / len(durs)
        f.write("%dx%dx%dx%d,%s,%d,%.2f,%.2f,%.1f\n" % (B, H, N, D, grp[0]["Kernel_Name"][:70].replace(",", ";"), len(durs), avg, min(durs),
                                                     4.0 * B * H * N * N * D / avg * 1e-6))
assert pos == len(rows), (pos, len(rows))
print(open(out).read())
