#!/bin/bash
# A/B round for the vision sweep: parity, vision-only loop at cfg2 and 10 x cfg2, per-rank passes, cfg3
set -u
R=$PWD; O=$R/gpurun_out/sweep_$1; mkdir -p $O
timeout 800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for spec in "cfg2 500 200" "cfg2 5000 100"; do
  set -- $spec
  python bench.py --workload $1 --frames $2 --steps $3 --warmup 10 --repeats 5 --no-cpu-baseline --no-secondary > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
  python -c "
import json; d=json.load(open('$O/bench_$1_$2.json')); print('$1 $2', d['ms_per_step']); print({k:round(1e3*v['avg_ms'],1) for k,v in d['kernels_in_loop'].items()})"
done
tools/perrank_round.sh sw 2>&1 | grep "frames:\|k_reproj_jac"
python bench.py --steps 40 --warmup 5 --repeats 5 --no-cpu-baseline --no-secondary > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print('cfg3', d['ms_per_step'], d['timing']); print({k:round(1e3*v['avg_ms'],1) for k,v in d['kernels_in_loop'].items()})"
