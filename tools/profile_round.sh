#!/bin/bash
# One round's profiling evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02 cfg3      -> gpurun_out/prof_r02_cfg3/{kernel_stats.txt, pmc_traffic.txt, pmc_traffic.json, bench.json}
# kernel trace and the two PMC passes are separate rocprofv3 runs (counters never together with tracing).
set -u
TAG=$1; WL=$2
OUT=$PWD/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
ROOT=$PWD
BENCH="python $ROOT/bench.py --workload $WL --steps 40 --warmup 5 --no-cpu-baseline --no-secondary"
cd /tmp && export TMPDIR=/tmp
# (kernel statistics also with the event hand-overs: per-kernel durations do not depend on the hand-over mode, except that with the
#  flags k_final / k_imu_block contain their waits; the pass timeline -- tools/timeline_round.sh -- is taken with the flags)
VICALIB_AMD_FLAG_SYNC=0 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $BENCH > $OUT/bench_traced.json 2> $OUT/trace.err
# counter passes serialise the kernels of all queues: a kernel that waits for a flag of the other stream would sit there
# until its bound runs out -- the counter passes therefore run with the event hand-overs (the bytes a kernel moves do not depend on them)
VICALIB_AMD_FLAG_SYNC=0 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f -- $BENCH > /dev/null 2> $OUT/fetch.err
VICALIB_AMD_FLAG_SYNC=0 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w -- $BENCH > /dev/null 2> $OUT/write.err
cd $ROOT
python tools/rocpd_stats.py $(ls $OUT/trace/*results.db | head -1) > $OUT/kernel_stats.txt
python tools/pmc_traffic.py $(ls $OUT/fetch/*results.db | head -1) $(ls $OUT/write/*results.db | head -1) $WL $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
rm -rf $OUT/trace $OUT/fetch $OUT/write
head -30 $OUT/kernel_stats.txt; cat $OUT/pmc_traffic.txt
