#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/red_$1; mkdir -p $O
timeout 800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
(for spec in "cfg3" "cfg4 2500" "cfg5 6250"; do VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_rstamps.so python tools/reduced_stamps.py $spec; done) > $O/reduced_stamps.txt 2>&1
cat $O/reduced_stamps.txt | grep -v "^$" | tail -50
tools/perrank_round.sh $1 2>&1 | grep -v "^   k_\(final\|imu_jac \|reproj_jac \|imu_delta+k_imu_block \)"
bash tools/timeline_round.sh cfg4 k_final 2500 > $O/timeline_cfg4_2500.txt 2>&1
python bench.py --steps 40 --warmup 5 --repeats 5 --no-cpu-baseline --no-secondary > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print('cfg3', d['ms_per_step'], d['timing'])"
