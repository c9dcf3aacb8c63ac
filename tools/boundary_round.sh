ROOT=$PWD; mkdir -p gpurun_out/tl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $ROOT/gpurun_out/tl/trace -o t -- python $ROOT/tools/trace_pass.py ${1:-cfg3} > $ROOT/gpurun_out/tl/out.txt 2> $ROOT/gpurun_out/tl/err.txt
cd $ROOT
python tools/solve_boundary.py $(ls gpurun_out/tl/trace/*results.db | head -1) 1
python tools/pass_timeline.py $(ls gpurun_out/tl/trace/*results.db | head -1) 4 k_final
rm -rf gpurun_out/tl/trace
