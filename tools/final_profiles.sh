set -u
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/tests_gpu.log 2>&1; tail -2 $O/tests_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
tools/profile_round.sh r03 cfg3 > $O/prof_cfg3.log 2>&1
tools/profile_round.sh r03 cfg4 > $O/prof_cfg4.log 2>&1
bash tools/timeline_round.sh cfg3 k_final > $O/pass_timeline_cfg3.txt 2>&1
VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_wstamps.so python tools/weights_stamps.py > $O/weights_stamps.txt 2>&1
VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_f2stamps.so python tools/f2_stamps.py > $O/fwd2_stamps.txt 2>&1
(VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_rstamps.so python tools/reduced_stamps.py cfg3; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_rstamps.so python tools/reduced_stamps.py cfg4 2500; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_rstamps.so python tools/reduced_stamps.py cfg5 6250) > $O/reduced_stamps.txt 2>&1
(for spec in "cfg3" "cfg4 2500" "cfg5 6250"; do echo "== $spec"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_gstamps.so python tools/gram_stamps.py $spec; done) > $O/gram_stamps.txt 2>&1
(for spec in "cfg3" "cfg4 2500" "cfg5 6250"; do echo "== $spec"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_bstamps.so python tools/back_stamps.py $spec; done) > $O/back_stamps.txt 2>&1
(for spec in "cfg3" "cfg4 2500"; do echo "== $spec"; VICALIB_AMD_LIB=$R/tools/probe/libvicalib_amd_jstamps.so python tools/jac_stamps.py $spec; done) > $O/jac_stamps.txt 2>&1
tools/perrank_round.sh final > $O/perrank.txt 2>&1
cd /tmp && export TMPDIR=/tmp
VICALIB_AMD_FLAG_SYNC=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU -d $O/pmc -o p -- python $R/bench.py --workload cfg3 --steps 20 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc.err
python $R/tools/rocpd_pmc.py $(ls $O/pmc/*results.db | head -1) > $O/sq_insts_cfg3.txt 2>&1; rm -rf $O/pmc
# the vision sweep's pipes at the per-rank size of cfg4 (one PMC pass, no tracing beside it)
VICALIB_AMD_FLAG_SYNC=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU -d $O/pmc2 -o p -- python $R/bench.py --workload cfg4 --frames 2500 --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc2.err
python $R/tools/rocpd_pmc.py $(ls $O/pmc2/*results.db | head -1) > $O/sq_sweep_cfg4_2500.txt 2>&1; rm -rf $O/pmc2
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/prof_cfg3.log; cat $O/perrank.txt | grep "frames:"; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['ms_per_step'], d['timing'], d['roofline']['kernel'], d['roofline']['frac'])"
