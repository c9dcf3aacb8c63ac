#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/vis_$1; mkdir -p $O
timeout 800 python -m pytest tests/test_gpu_parity.py tests/test_robustness_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
python bench.py --workload cfg2 --steps 200 --warmup 10 --repeats 5 --no-cpu-baseline --no-secondary > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python -c "
import json; d=json.load(open('$O/bench_cfg2.json')); print('cfg2', d['ms_per_step'], d['timing']); print({k:round(1e3*v['avg_ms'],1) for k,v in d['kernels_in_loop'].items()})"
python bench.py --workload cfg2 --frames 5000 --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_cfg2x10.json 2> $O/bench_cfg2x10.err
python -c "
import json; d=json.load(open('$O/bench_cfg2x10.json')); print('cfg2x10', d['ms_per_step'], d['timing']); print({k:round(1e3*v['avg_ms'],1) for k,v in d['kernels_in_loop'].items()})"
