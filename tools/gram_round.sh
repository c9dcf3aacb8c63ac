#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/gram_$1; mkdir -p $O
timeout 800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
for s in "cfg3" "cfg4 2500" "cfg5 6250"; do echo $s; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_gstamps.so python tools/gram_stamps.py $s; done
tools/perrank_round.sh $1 2>&1 | grep -v "^   k_\(imu_jac \|reproj_jac \|imu_delta+k_imu_block \)"
python bench.py --steps 40 --warmup 5 --repeats 5 --no-cpu-baseline --no-secondary > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print('cfg3', d['ms_per_step'], d['timing']); print({k:round(1e3*v['avg_ms'],1) for k,v in d['kernels_in_loop'].items()})"
