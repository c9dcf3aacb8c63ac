#!/bin/bash
# per-rank passes of the multi-GPU configurations on one GPU (no separators / all-reduces): tools/perrank_round.sh TAG
set -u
TAG=$1; ROOT=$PWD; OUT=$ROOT/gpurun_out/pr_$TAG; mkdir -p $OUT
# (cfg3 / 2: 1000 frames through the SHARDED code path -- split k_reduced / k_final around the two all-reduces, which go through the
#  library's RCCL communicator with its one rank; without a peer there is no separator frame: D = 29, not the 38 of a real two-rank run)
for spec in "cfg3 1000 1" "cfg4 2500 0" "cfg5 6250 0"; do
  set -- $spec
  VICALIB_AMD_FORCE_SHARD_PATH=$3 python bench.py --workload $1 --frames $2 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_$1_$2.json 2> $OUT/bench_$1_$2.err
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_$1_$2.json") if l.startswith("{")][-1])      # (RCCL prints a banner to stdout ahead of the line)
print("$1 $2 frames: ms_per_step %.4f (min %.4f max %.4f) D=%d" % (d["ms_per_step"], d["timing"]["ms_per_step_min"], d["timing"]["ms_per_step_max"], d["config"]["reduced_dim"]))
for k, v in sorted(d["kernels_in_loop"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print("   %-34s avg %8.1f us  per step %8.1f us  %s" % (k, 1e3 * v["avg_ms"], 1e3 * v["ms_per_step"], v["stream"][:1]))
PY
done
