#!/bin/bash
# k_imu_weights on the GPU box: parity tests of the weight update, kernel times at cfg3 / cfg4, executed VALU instructions.
#   tools/weights_round.sh TAG [cfg4]      -> gpurun_out/w_TAG/*
set -u
TAG=$1; BIG=${2:-}
ROOT=$PWD; OUT=$ROOT/gpurun_out/w_$TAG; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "weight or imu_blocks or visual_inertial" > $OUT/tests.txt 2>&1; tail -3 $OUT/tests.txt
cd /tmp && export TMPDIR=/tmp
for WL in cfg3 $BIG; do
  B="python $ROOT/bench.py --workload $WL --steps 40 --warmup 5 --no-cpu-baseline --no-secondary"
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$WL -o t -- $B > $OUT/bench_$WL.json 2> $OUT/trace_$WL.err
  python $ROOT/tools/rocpd_stats.py $(ls $OUT/trace_$WL/*results.db | head -1) > $OUT/kernel_stats_$WL.txt
  rm -rf $OUT/trace_$WL
  head -12 $OUT/kernel_stats_$WL.txt
done
B="python $ROOT/bench.py --workload cfg3 --steps 20 --warmup 2 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/pmc -o p -- $B > /dev/null 2> $OUT/pmc.err
python $ROOT/tools/rocpd_pmc.py $(ls $OUT/pmc/*results.db | head -1) > $OUT/pmc_insts.txt 2>&1
rm -rf $OUT/pmc
grep -i "weights\|kernel " $OUT/pmc_insts.txt
# phase stamps of the first wavefront (a -DVC_W_STAMPS build, prepared before the push: tools/probe/libvicalib_amd_wstamps.so)
if [ -f $ROOT/tools/probe/libvicalib_amd_wstamps.so ]; then
  cd $ROOT && VICALIB_AMD_LIB=$ROOT/tools/probe/libvicalib_amd_wstamps.so python tools/weights_stamps.py > $OUT/stamps.txt 2>&1; cat $OUT/stamps.txt
fi
