#!/bin/bash
# A/B of two builds of the library on ONE box: tools/ab_libs.sh libA.so libB.so [rounds] -- the N = 1 bench loop alternating, then a pass timeline of each
cd "$(dirname "$0")/.."
A="$1"; B="$2"; R=${3:-3}
for r in $(seq $R); do
  for L in "$A" "$B"; do
    ms=$(VICALIB_AMD_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-secondary --repeats 5 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%.4f (min %.4f)' % (d['ms_per_step'], d['timing']['ms_per_step_min']))")
    echo "round $r  [$L]  ms_per_step $ms"
  done
done
for L in "$A" "$B"; do echo "== timeline, $L"; VICALIB_AMD_LIB=$PWD/$L bash tools/timeline_round.sh cfg3 k_final 2>&1 | tail -17; done
