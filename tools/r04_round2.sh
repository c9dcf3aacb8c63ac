#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/${1:-r04d}; mkdir -p $O
timeout 900 python -m pytest tests/test_flag_sync_gpu.py -x -q -m gpu -s > $O/tests_sync.log 2>&1; tail -15 $O/tests_sync.log
timeout 900 python -m pytest tests/test_sharding.py -x -q -m gpu -k "two_processes or two_rank_sharded_visual" > $O/tests_shard.log 2>&1; tail -5 $O/tests_shard.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests_parity.log 2>&1; tail -3 $O/tests_parity.log
VICALIB_AMD_TIMING=1 python bench.py --workload cfg3 --no-cpu-baseline --no-secondary > $O/bench_cfg3.json 2> $O/bench_cfg3.err
grep -c "run_iterations" $O/bench_cfg3.err; grep "run_iterations" $O/bench_cfg3.err | tail -8
python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print(d['ms_per_step'], d['timing'])"
