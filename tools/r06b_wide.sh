#!/bin/bash
# wide reduced systems: parity tests that go through the blocked solve + per-rank benches + phase stamps
set -u
R=$PWD; O=$R/gpurun_out/${1:-r06b_w}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sharding.py -x -q -m gpu -k "traces or reduced or shard or rank" > $O/tests.log 2>&1; tail -3 $O/tests.log
for spec in "cfg4 2500" "cfg5 6250"; do
  set -- $spec
  python bench.py --workload $1 --frames $2 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
  python - <<PY
import json
d = json.load(open("$O/bench_$1_$2.json"))
print("$1/$2 ms_per_step %.4f" % d["ms_per_step"], " k_reduced %.1f us" % (1e3 * d["kernels_in_loop"]["k_reduced"]["avg_ms"]))
PY
  echo "== reduced stamps $spec"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_rstamps.so python tools/reduced_stamps.py $spec 2>&1 | tail -9
done
