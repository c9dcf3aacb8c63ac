#!/bin/bash
# round 6 evidence on the GPU box (one gpurun call): kernel statistics + PMC traffic (cfg3, cfg4), pass timeline, solve boundary,
# per-rank passes (cfg3 / 2 through the sharded path, cfg4 / 4, cfg5 / 8), SQ counters, k_imu_block / k_imu_jac phase stamps,
# default bench line.  Outputs under gpurun_out/final_r06d/; copy what is to be judged into profiles/.
set -u
R=$PWD; O=$R/gpurun_out/final_r06d; mkdir -p $O
tools/profile_round.sh r06d cfg3 > $O/prof_cfg3.log 2>&1
tools/profile_round.sh r06d cfg4 > $O/prof_cfg4.log 2>&1
bash tools/timeline_round.sh cfg3 k_final > $O/pass_timeline_cfg3.txt 2>&1
bash tools/boundary_round.sh cfg3 > $O/solve_boundary_cfg3.txt 2>&1
tools/perrank_round.sh final_r06d > $O/perrank.txt 2>&1
python bench.py --workload cfg5 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_cfg5_full.json 2> $O/bench_cfg5_full.err
python bench.py --workload cfg4 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary > $O/bench_cfg4_full.json 2> $O/bench_cfg4_full.err
cd /tmp && export TMPDIR=/tmp
for spec in "cfg3 0 sq_cfg3" "cfg4 2500 sq_sweep_cfg4_2500"; do
  set -- $spec
  FR=""; [ "$2" != "0" ] && FR="--frames $2"
  VICALIB_AMD_FLAG_SYNC=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU -d $O/pmc_$3 -o p -- python $R/bench.py --workload $1 $FR --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_$3.err
  python $R/tools/rocpd_pmc.py $(ls $O/pmc_$3/*results.db | head -1) > $O/$3.txt 2>&1; rm -rf $O/pmc_$3
done
cd $R
( echo "== k_imu_block, cfg3"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_ibstamps.so python tools/ib_stamps.py cfg3; echo "== k_imu_jac, cfg3"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_ijstamps.so python tools/ij_stamps.py cfg3 ) > $O/imu_stamps.txt 2>&1
python bench.py --gpus 2 --transport gloo --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline > $O/bench_two_ranks_gloo_one_gpu.json 2> $O/bench_two_ranks_gloo_one_gpu.err
( echo "== k_chain_fwd2, cfg3"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_f2stamps.so python tools/f2_stamps.py ) > $O/f2_stamps.txt 2>&1
( for spec in "cfg3" "cfg4 2500" "cfg5 6250"; do echo "== k_reduced, $spec"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_rstamps.so python tools/reduced_stamps.py $spec; done ) > $O/reduced_stamps.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/prof_cfg3.log; grep "frames:" $O/perrank.txt; python -c "
import json
txt=open('$O/bench_default.json').read(); d=json.loads([l for l in txt.splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d['timing'], d['roofline']['kernel'], d['roofline']['frac'], d.get('complete_calibration'))"
