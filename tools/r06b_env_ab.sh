#!/bin/bash
# same-box A/B of an environment switch on the cfg3 bench: tools/r06b_env_ab.sh VAR=VALUE
set -u
for rep in 1 2 3; do
  for mode in base "$1"; do
    if [ "$mode" = base ]; then r=$(python bench.py --workload cfg3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1); else r=$(env "$mode" python bench.py --workload cfg3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1); fi
    echo "$mode $(echo "$r" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4f" % d["ms_per_step"], " ".join("%s=%.1f" % (k[:14], 1e3*v["avg_ms"]) for k, v in d["kernels_in_loop"].items() if "trial" in k))')"
  done
done
