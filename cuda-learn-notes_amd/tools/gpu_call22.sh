#!/bin/bash
# the driver's bench command (20 steps, 5 warm-up) as the FIRST GPU work on a fresh box: graph and eager launch modes
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
for i in 1 2; do for mode in graph eager; do
  timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --launch $mode > $OUT/c22_${mode}_$i.json 2> $OUT/c22_${mode}_$i.err || { echo "bench $mode $i failed"; tail -5 $OUT/c22_${mode}_$i.err; exit 7; }
  python - <<PY
import json
d=json.loads(open("$OUT/c22_${mode}_$i.json").read().strip().splitlines()[-1])
c=d["config"]
print("$mode $i value", d["value"], "ev-based", d["roofline"]["achieved"], "ms/step", d["ms_per_step"], "enqueue", c["host_enqueue_ms_per_step"], "launch:", c["launch"], "settle", c["settle_ms_per_step"])
PY
done; done
