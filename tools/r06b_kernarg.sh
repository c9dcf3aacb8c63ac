#!/bin/bash
# where the kernel arguments live: HIP_FORCE_DEV_KERNARG = 0 / 1 / unset, cfg3 bench (same box, alternating)
set -u
R=$PWD; O=$R/gpurun_out/${1:-r06b_ka}; mkdir -p $O
for rep in 1 2; do
for ka in unset 0 1; do
  if [ $ka = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$ka; fi
  python bench.py --workload cfg3 --no-cpu-baseline --no-secondary > $O/bench_$ka.json 2> $O/bench_$ka.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/bench_$ka.json") if l.startswith("{")][-1])
print("HIP_FORCE_DEV_KERNARG=$ka  ms_per_step %.4f (min %.4f max %.4f)" % (d["ms_per_step"], d["timing"]["ms_per_step_min"], d["timing"]["ms_per_step_max"]))
PY
done
done
