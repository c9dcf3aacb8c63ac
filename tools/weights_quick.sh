#!/bin/bash
# quick round for k_imu_weights: parity tests + phase stamps (+ kernel stats when asked).  tools/weights_quick.sh TAG [stats]
set -u
TAG=$1; ROOT=$PWD; OUT=$ROOT/gpurun_out/w_$TAG; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "weight or imu_blocks or visual_inertial" 2>&1 | tail -4 | tee $OUT/tests.txt
VICALIB_AMD_LIB=$ROOT/tools/probe/libvicalib_amd_wstamps.so python tools/weights_stamps.py 2>&1 | tee $OUT/stamps.txt
if [ "${2:-}" = "stats" ]; then
  cd /tmp && export TMPDIR=/tmp
  for WL in cfg3 cfg4; do
    B="python $ROOT/bench.py --workload $WL --steps 40 --warmup 5 --no-cpu-baseline --no-secondary"
    rocprofv3 --kernel-trace --stats -d $OUT/trace_$WL -o t -- $B > $OUT/bench_$WL.json 2> $OUT/trace_$WL.err
    python $ROOT/tools/rocpd_stats.py $(ls $OUT/trace_$WL/*results.db | head -1) > $OUT/kernel_stats_$WL.txt
    rm -rf $OUT/trace_$WL
    grep "kernel \|weights\|imu_delta\|imu_block" $OUT/kernel_stats_$WL.txt
  done
fi
