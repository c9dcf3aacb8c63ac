#!/bin/bash
# round 6 quick look: IMU / trace parity tests, cfg3 bench, pass timeline.  tools/r06_quick.sh TAG [pytest -k expression]
set -u
R=$PWD; O=$R/gpurun_out/${1:-r06_q}; mkdir -p $O
K=${2:-"imu or traces or visual_inertial or chain"}
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --no-cpu-baseline --no-secondary > $O/bench_cfg3.json 2> $O/bench_cfg3.err; python - <<PY
import json
d = json.loads([l for l in open("$O/bench_cfg3.json") if l.startswith("{")][-1])
print("cfg3 ms_per_step", d["ms_per_step"], d["timing"])
for k, v in sorted(d["kernels_in_loop"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print("   %-34s avg %8.1f us  per step %8.1f us  %s" % (k, 1e3 * v["avg_ms"], 1e3 * v["ms_per_step"], v["stream"][:1]))
PY
bash tools/timeline_round.sh cfg3 k_final > $O/pass_timeline_cfg3.txt 2>&1; tail -17 $O/pass_timeline_cfg3.txt
