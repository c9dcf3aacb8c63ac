#!/bin/bash
# A/B of the cross-stream hand-overs (device flags vs event records): parity, cfg3 bench both ways, timeline, per-rank passes
set -u
R=$PWD; O=$R/gpurun_out/sync_$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cli.py tests/test_robustness_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
for f in 1 0; do
  VICALIB_AMD_FLAG_SYNC=$f timeout 300 python bench.py --steps 40 --warmup 5 --repeats 5 --no-cpu-baseline --no-secondary > $O/bench_cfg3_f$f.json 2> $O/bench_cfg3_f$f.err
  python -c "
import json; d=json.load(open('$O/bench_cfg3_f$f.json')); print('cfg3 flag_sync=$f', d['ms_per_step'], d['timing'], d['complete_calibration']['seconds'])"
done
bash tools/timeline_round.sh cfg3 k_final > $O/timeline_cfg3.txt 2>&1; head -21 $O/timeline_cfg3.txt
tools/perrank_round.sh sync 2>&1 | grep "frames:"
