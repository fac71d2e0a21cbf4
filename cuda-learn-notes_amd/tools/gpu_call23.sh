#!/bin/bash
# final bench lines of the round: the driver's command (20 steps, 5 warm-up) first on a fresh box, then the default run
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r02_bench_20steps.json 2> $OUT/r02_bench_20steps.err || { echo bench20 failed; tail -5 $OUT/r02_bench_20steps.err; exit 7; }
timeout 200 python bench.py > $OUT/r02_bench_default.json 2> $OUT/r02_bench_default.err || { echo bench failed; exit 8; }
python - <<PY
import json
for f in ("r02_bench_20steps.json","r02_bench_default.json"):
    d=json.loads(open("$OUT/"+f).read().strip().splitlines()[-1]); c=d["config"]
    print(f, d["value"], d["roofline"]["frac"], c["settle_ms_per_step"], c["host_enqueue_ms_per_step"], "| C4", d["roofline_fa2_c4_d64"]["achieved"], "d128", d["roofline_fa2_d128"]["achieved"], "c5", d["roofline_fa2_c5_d512"]["achieved"])
    print("   extras", json.dumps(d["extras"])[:900])
PY
