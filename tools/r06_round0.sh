#!/bin/bash
# round 6, first call: the whole GPU suite on the round's starting code (+ D = 32 fix), default bench, pass timeline
set -u
R=$PWD; O=$R/gpurun_out/r06_0; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --no-cpu-baseline --no-secondary > $O/bench_cfg3.json 2> $O/bench_cfg3.err; python - <<PY
import json
d = json.loads([l for l in open("$O/bench_cfg3.json") if l.startswith("{")][-1])
print("cfg3 ms_per_step", d["ms_per_step"], d["timing"])
PY
bash tools/timeline_round.sh cfg3 k_final > $O/pass_timeline_cfg3.txt 2>&1; tail -25 $O/pass_timeline_cfg3.txt
