#!/bin/bash
# round 6: the split-row Gram blocks of the radial models (fov / linear / poly2 / poly3): parity tests, then A/B against the build without
# (tools/probe/libvicalib_amd_nosplit.so: -DVC_JAC_SPLIT_ROWS=0) on one box
set -u
R=$PWD; O=$R/gpurun_out/r06_split; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linearisation or cfg1 or cfg2 or mixed or traces or cfg4 or fixed_intrinsics or ragged or widest or pnp or eight or full_size" > $O/tests.log 2>&1; tail -3 $O/tests.log
run() {   # workload, frames (0: all), label
  FR=""; [ "$2" != "0" ] && FR="--frames $2"
  python bench.py --workload $1 $FR --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k=d['kernels_in_loop']; n=[x for x in ('k_reproj_jac(trial)','k_trial') if x in k][0]; print('$3 $1 $2', 'ms_per_step %.4f' % d['ms_per_step'], n, '%.1f us' % (1e3*k[n]['avg_ms']))"
}
for spec in "cfg4 2500" "cfg5 6250" "cfg2 0"; do
  set -- $spec
  for rep in 1 2; do
    VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_nosplit.so run $1 $2 "four-column / both rows per step"
    run $1 $2 "split rows                      "
  done
done
