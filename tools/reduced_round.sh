#!/bin/bash
# k_reduced work round: parity subset, phase stamps at the three sizes, per-rank passes, cfg3 A/B of the k_imu_jac placement
set -u
R=$PWD; O=$R/gpurun_out/red_$1; mkdir -p $O
timeout 800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu  > $O/tests.log 2>&1; tail -3 $O/tests.log
(for spec in "cfg3" "cfg4 2500" "cfg5 6250"; do VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_rstamps.so python tools/reduced_stamps.py $spec; done) > $O/reduced_stamps.txt 2>&1
grep -c . $O/reduced_stamps.txt; grep "total\|solve done" $O/reduced_stamps.txt
tools/perrank_round.sh $1 2>&1 | grep -v "^   k_\(part\|final\|imu_jac \|reproj_jac \|imu_delta+k_imu_block \)"
for j in 1 0; do
  VICALIB_AMD_JAC_STREAM2=$j python bench.py --steps 40 --warmup 5 --repeats 5 --no-cpu-baseline --no-secondary > $O/bench_cfg3_j$j.json 2> $O/bench_cfg3_j$j.err
  python -c "
import json; d=json.load(open('$O/bench_cfg3_j$j.json')); print('cfg3 jac_stream2=$j', d['ms_per_step'], d['timing'])"
done
