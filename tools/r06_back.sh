#!/bin/bash
# round 6: the generalised one-launch back-substitution (wide borders, sharded passes): tests + per-rank passes
set -u
R=$PWD; O=$R/gpurun_out/r06_back; mkdir -p $O
timeout 1700 python -m pytest tests/test_back_path_gpu.py tests/test_sharding.py tests/test_baseline_full_size.py -x -q -m gpu -k "${1:-}" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "traces or chain_elimination" > $O/tests2.log 2>&1; tail -3 $O/tests2.log
tools/perrank_round.sh r06_back > $O/perrank.txt 2>&1; grep -v "^Traceback\|^  File\|^    \|^json" $O/perrank.txt
