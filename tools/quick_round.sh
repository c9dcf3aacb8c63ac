#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/${1:-quick}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_flag_sync_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
show() { python - <<PY
import json
d = json.load(open("$1"))
print("$2 ms_per_step %.4f (min %.4f max %.4f)" % (d["ms_per_step"], d["timing"]["ms_per_step_min"], d["timing"]["ms_per_step_max"]))
for k, v in sorted(d["kernels_in_loop"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print("   %-34s avg %8.1f us  per step %8.1f us  %s" % (k, 1e3 * v["avg_ms"], 1e3 * v["ms_per_step"], v["stream"][:1]))
PY
}
python bench.py --workload cfg3 --no-cpu-baseline --no-secondary > $O/bench_cfg3.json 2> $O/bench_cfg3.err; show $O/bench_cfg3.json cfg3 | head -8
bash tools/timeline_round.sh cfg3 k_final > $O/pass_timeline_cfg3.txt 2>&1; tail -25 $O/pass_timeline_cfg3.txt
python bench.py --workload cfg4 --frames 2500 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_cfg4_2500.json 2> $O/bench_cfg4_2500.err; show $O/bench_cfg4_2500.json cfg4/2500
python bench.py --workload cfg5 --frames 6250 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_cfg5_6250.json 2> $O/bench_cfg5_6250.err; show $O/bench_cfg5_6250.json cfg5/6250
