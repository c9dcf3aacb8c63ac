#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/${1:-r04k}; mkdir -p $O
timeout 900 python -m pytest tests/test_sharding.py -x -q -m gpu -k "flag_handovers or two_rank" > $O/tests_shard.log 2>&1; tail -5 $O/tests_shard.log
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_sharding.py > $O/tests.log 2>&1; tail -3 $O/tests.log
VICALIB_AMD_TIMING=1 python bench.py --workload cfg3 --no-cpu-baseline --no-secondary > $O/bench_cfg3.json 2> $O/bench_cfg3.err
grep -i "stage" $O/bench_cfg3.err | tail -4
python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print(d['ms_per_step'], d['timing'], d.get('complete_calibration'))"
VICALIB_AMD_FORCE_SHARD_PATH=1 python bench.py --workload cfg3 --no-cpu-baseline --no-secondary > $O/bench_cfg3_shardpath.json 2> $O/bench_cfg3_shardpath.err
VICALIB_AMD_FORCE_SHARD_PATH=1 VICALIB_AMD_SHARD_FLAG_SYNC=1 python bench.py --workload cfg3 --no-cpu-baseline --no-secondary > $O/bench_cfg3_shardflags.json 2> $O/bench_cfg3_shardflags.err
python -c "
import json
for n in ('shardpath','shardflags'):
    d=json.load(open('$O/bench_cfg3_%s.json'%n)); print(n, d['ms_per_step'], d['timing']['ms_per_step_min'])"
