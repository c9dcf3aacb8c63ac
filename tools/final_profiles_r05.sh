#!/bin/bash
# round 5 evidence on the GPU box (one gpurun call): kernel statistics + PMC traffic (cfg3, cfg4), pass timeline, solve boundary,
# per-rank passes, default bench line.  Outputs under gpurun_out/final_r05/; copy what is to be judged into profiles/.
set -u
R=$PWD; O=$R/gpurun_out/final_r05; mkdir -p $O
tools/profile_round.sh r05 cfg3 > $O/prof_cfg3.log 2>&1
tools/profile_round.sh r05 cfg4 > $O/prof_cfg4.log 2>&1
bash tools/timeline_round.sh cfg3 k_final > $O/pass_timeline_cfg3.txt 2>&1
bash tools/boundary_round.sh cfg3 > $O/solve_boundary_cfg3.txt 2>&1
tools/perrank_round.sh final_r05 > $O/perrank.txt 2>&1
python bench.py --workload cfg5 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_cfg5_full.json 2> $O/bench_cfg5_full.err
cd /tmp && export TMPDIR=/tmp
VICALIB_AMD_FLAG_SYNC=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU -d $O/pmc2 -o p -- python $R/bench.py --workload cfg4 --frames 2500 --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc2.err
python $R/tools/rocpd_pmc.py $(ls $O/pmc2/*results.db | head -1) > $O/sq_sweep_cfg4_2500.txt 2>&1; rm -rf $O/pmc2
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/prof_cfg3.log; grep "frames:" $O/perrank.txt; python -c "
import json
txt=open('$O/bench_default.json').read(); d=json.loads([l for l in txt.splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d['timing'], d['roofline']['kernel'], d['roofline']['frac'], d.get('complete_calibration'))"
# round 5 extras: phase stamps of the kernels this round changed (k_chain_init, k_final, k_reduced) and the multi-rank bench branch on one GPU
cd $R
for spec in "cfg3" "cfg4 2500" "cfg5 6250"; do echo "== k_chain_init, $spec"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_istamps.so python tools/init_stamps.py $spec; done > $O/init_stamps.txt 2>&1
( echo "== k_chain_l0 (chain assembly folded into the bottom level), cfg3"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_l0stamps.so python tools/l0_stamps.py ) > $O/l0_stamps.txt 2>&1
( echo "== k_final, cfg3"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_fstamps.so python tools/final_stamps.py; echo "== k_reduced, cfg3"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_rstamps.so python tools/reduced_stamps.py cfg3 ) > $O/final_reduced_stamps.txt 2>&1
python bench.py --gpus 2 --transport gloo --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline > $O/bench_two_ranks_gloo_one_gpu.json 2> $O/bench_two_ranks_gloo_one_gpu.err
tools/ab_bench.sh "VICALIB_AMD_EARLY_GRAM=1" "VICALIB_AMD_EARLY_GRAM=0" 2 > $O/ab_early_gram.txt 2>&1
tools/ab_bench.sh "VICALIB_AMD_FOLD_L0=1" "VICALIB_AMD_FOLD_L0=0" 3 > $O/ab_fold_l0.txt 2>&1
tools/diag_fold.sh > $O/fold_vs_apart.txt 2>&1
