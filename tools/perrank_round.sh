#!/bin/bash
# per-rank passes of the multi-GPU configurations on one GPU (no separators / all-reduces): tools/perrank_round.sh TAG
set -u
TAG=$1; ROOT=$PWD; OUT=$ROOT/gpurun_out/pr_$TAG; mkdir -p $OUT
for spec in "cfg4 2500" "cfg5 6250"; do
  set -- $spec
  python bench.py --workload $1 --frames $2 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_$1_$2.json 2> $OUT/bench_$1_$2.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_$1_$2.json"))
print("$1 $2 frames: ms_per_step %.4f (min %.4f max %.4f) D=%d" % (d["ms_per_step"], d["timing"]["ms_per_step_min"], d["timing"]["ms_per_step_max"], d["config"]["reduced_dim"]))
for k, v in sorted(d["kernels_in_loop"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print("   %-34s avg %8.1f us  per step %8.1f us  %s" % (k, 1e3 * v["avg_ms"], 1e3 * v["ms_per_step"], v["stream"][:1]))
PY
done
