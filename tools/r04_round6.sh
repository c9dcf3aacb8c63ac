#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/${1:-r04j}; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
VICALIB_AMD_TIMING=1 python bench.py --workload cfg3 --no-cpu-baseline --no-secondary > $O/bench_cfg3.json 2> $O/bench_cfg3.err
grep "upload:" $O/bench_cfg3.err | tail -5; grep -i "stage\|solve:" $O/bench_cfg3.err | head -12
python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print(d['ms_per_step'], d['timing'], d.get('complete_calibration'))"
