"""A handful of plain LM passes (no in-loop timing events) of a BASELINE workload's final stage, for tools/pass_timeline.py:
    rocprofv3 --kernel-trace -d out -o t -- python tools/trace_pass.py cfg3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
p = synth.generate_native(synth.BASELINE_CONFIGS[name])
cal = ViCalibrator(0).load_problem(p)
if os.environ.get("VICALIB_AMD_FORCE_SHARD_PATH") == "1":
    cal.set_shard_rccl(0, 1)        # the sharded code path (split kernels, all-reduces) with one rank
if p.imu_t is not None:
    cal.SetStageLimit(3); cal.Solve()
else:
    cal.SetCalibrateImu(False)
cal.prepare()
cal.run_iterations(5)
cal.run_iterations(20)
