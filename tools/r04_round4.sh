#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/${1:-r04g}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_flag_sync_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for tb in 0 1; do for wb in 0 1; do
VICALIB_AMD_CHAIN_TWO_BOTTOM=$tb VICALIB_AMD_WEIGHTS_BEHIND_L0=$wb python bench.py --workload cfg3 --no-cpu-baseline --no-secondary > $O/bench_cfg3_$tb$wb.json 2> $O/bench_cfg3_$tb$wb.err
python -c "
import json; d=json.load(open('$O/bench_cfg3_$tb$wb.json')); print('two_bottom=$tb weights_behind=$wb', d['ms_per_step'], d['timing']['ms_per_step_min'], 'fwd', d['kernels_in_loop']['k_chain_fwd']['avg_ms'], 'back', d['kernels_in_loop']['k_chain_back']['avg_ms'])"
done; done
bash tools/timeline_round.sh cfg3 k_final > $O/pass_timeline_cfg3.txt 2>&1; tail -26 $O/pass_timeline_cfg3.txt
