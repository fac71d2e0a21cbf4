"""Every reference-style bench script (cuda-learn-notes_amd/kernels/<topic>/<topic>.py: the reference's shapes, rows and timing protocol), run whole,
and per section the fastest of our rungs against the script's own torch row (`*_th*`).  Finds shapes where a family is behind torch.
python scripts_vs_torch.py [topic ...]   (GPU box)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TOPICS = ["elementwise/elementwise", "reduce/block_all_reduce", "softmax/softmax", "layer-norm/layer_norm", "rms-norm/rms_norm", "rope/rope",
          "embedding/embedding", "relu/relu", "gelu/gelu", "elu/elu", "sigmoid/sigmoid", "swish/swish", "hardswish/hardswish", "hardshrink/hardshrink",
          "dot-product/dot_product", "sgemv/sgemv", "hgemv/hgemv", "mat-transpose/mat_transpose"]
want = sys.argv[1:]
worst = []
for topic in TOPICS:
    if want and not any(w in topic for w in want):
        continue
    path = os.path.join(ROOT, "cuda-learn-notes_amd", "kernels", topic + ".py")
    try:
        out = subprocess.run([sys.executable, path], capture_output=True, text=True, timeout=280).stdout
    except subprocess.TimeoutExpired:
        print("SVT %-28s timed out" % topic, flush=True)
        continue
    header, ours = "", []
    for line in out.splitlines():
        m = re.match(r"\s*out_(\S+)\s*:.*time:\s*([\d.]+)\s*ms", line)
        if not m:
            if line.strip() and not set(line.strip()) <= {"-"}:
                header = line.strip()[:40]
            continue
        tag, ms = m.group(1), float(m.group(2))
        if "_th" in tag:
            if ours:
                best = min(ours, key=lambda r: r[1])
                ratio = best[1] / ms
                worst.append((ratio, topic, header, best[0], best[1], ms))
                print("SVT %-24s %-36s best %-24s %8.2f us | %-14s %8.2f us | ours/torch %.2f" % (
                    topic.split("/")[1], header, best[0], best[1] * 1e3, tag, ms * 1e3, ratio), flush=True)
            ours = []
        else:
            ours.append((tag, ms))
worst.sort(reverse=True)
print("SVTSUM sections %d, behind torch (ours/torch > 1.05): %d" % (len(worst), sum(r[0] > 1.05 for r in worst)))
for r in worst[:12]:
    print("SVTSUM worst %.2f  %s  %s  %s %.2f us vs %.2f us" % (r[0], r[1].split("/")[1], r[2], r[3], r[4] * 1e3, r[5] * 1e3))
