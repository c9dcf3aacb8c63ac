"""A handful of plain LM passes (no in-loop timing events) of a BASELINE workload's final stage, for tools/pass_timeline.py:
    rocprofv3 --kernel-trace -d out -o t -- python tools/trace_pass.py cfg3 [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = synth.BASELINE_CONFIGS[name]
if len(sys.argv) > 2:        # a frame count other than the configuration's (per-rank size of a sharded run)
    import dataclasses
    cfg = dataclasses.replace(cfg, n_frames=int(sys.argv[2]))
p = synth.generate_native(cfg)
cal = ViCalibrator(0).load_problem(p)
if os.environ.get("VICALIB_AMD_FORCE_SHARD_PATH") == "1":
    cal.set_shard_rccl(0, 1)        # the sharded code path (split kernels, all-reduces) with one rank
if p.imu_t is not None:
    cal.SetStageLimit(3); cal.Solve()
else:
    cal.SetCalibrateImu(False)
cal.prepare()
cal.run_iterations(5)
cal.run_iterations(30)
