#!/bin/bash
# register-tiled reduced factorisation (D > 32): parity tests of the wide / sharded systems, per-rank benches alternating with the panel form
# (tools/probe/libvicalib_amd_single.so: -DVC_REDUCED_TILED=0), phase stamps
set -u
R=$PWD; O=$R/gpurun_out/${1:-r06c_tiled}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_sharding.py tests/test_reduced_side_jobs_gpu.py tests/test_baseline_full_size.py -x -q -m gpu -k "traces or reduced or shard or rank or full or cfg5 or cfg4 or covariance" > $O/tests.log 2>&1; tail -3 $O/tests.log
for rep in 1 2; do
for lib in vicalib_amd/libvicalib_amd.so tools/probe/libvicalib_amd_single.so; do
  for spec in "cfg4 2500" "cfg5 6250"; do
    set -- $spec
    VICALIB_AMD_LIB=$R/$lib python bench.py --workload $1 --frames $2 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
    python - <<PY
import json
d = json.loads([l for l in open("$O/bench_$1_$2.json") if l.startswith("{")][-1])
k = d["kernels_in_loop"]
print("$1/$2 [$lib] ms_per_step %.4f  k_reduced %.1f us" % (d["ms_per_step"], 1e3 * k["k_reduced"]["avg_ms"]))
PY
  done
done
done
for spec in "cfg4 2500" "cfg5 6250"; do echo "== k_reduced, $spec"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_rstamps.so python tools/reduced_stamps.py $spec; done 2>&1 | tee $O/reduced_stamps.txt
