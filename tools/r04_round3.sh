#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/${1:-r04f}; mkdir -p $O
timeout 900 python -m pytest tests/test_flag_sync_gpu.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
VICALIB_AMD_TIMING=1 python bench.py --workload cfg3 --no-cpu-baseline --no-secondary > $O/bench_cfg3.json 2> $O/bench_cfg3.err
grep "run_iterations" $O/bench_cfg3.err | tail -4
python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print(d['ms_per_step'], d['timing'], d.get('complete_calibration'))"
bash tools/boundary_round.sh cfg3 > $O/boundary.txt 2>&1; head -30 $O/boundary.txt
