#!/bin/bash
# A/B of the N = 1 bench loop on ONE box: tools/ab_bench.sh "VAR=VALUE ..." "VAR=VALUE ..." [rounds]   (each quoted set of assignments is a variant)
cd "$(dirname "$0")/.."
A="$1"; B="$2"; R=${3:-3}
for r in $(seq $R); do
  for V in "$A" "$B"; do
    ms=$(env $V python bench.py --no-cpu-baseline --no-secondary --repeats 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f (min %.4f)' % (d['ms_per_step'], d['timing']['ms_per_step_min']))")
    echo "round $r  [${V:-default}]  ms_per_step $ms"
  done
done
