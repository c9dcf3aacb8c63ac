"""Does the host program's own stream usage slow the visual-inertial pass?  HIP multiplexes the streams of a process onto
GPU_MAX_HW_QUEUES (4) hardware queues per priority level; torch with an eagerly created NCCL communicator
(init_process_group(..., device_id=...)) brings enough streams of its own that the library's two streams can share one queue,
which serialises the pass.  The library's second stream therefore has the lowest priority (own queue pool).
    python tools/stream_alias_probe.py plain|dist|dist_lazy            # one process, cfg3, ms per LM iteration
    VICALIB_AMD_FORCE_SHARD_PATH=1 python tools/stream_alias_probe.py shard|dist_shard
    VICALIB_AMD_STREAM2_PRIORITY=default python tools/stream_alias_probe.py dist      # the old behaviour: 0.46 instead of 0.30 ms
    GPU_MAX_HW_QUEUES=1 python tools/stream_alias_probe.py plain"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from vicalib_amd import synth
from vicalib_amd.lib import ViCalibrator
mode = sys.argv[1]
torch.cuda.set_device(0)
if mode in ("dist", "dist_shard", "dist_lazy"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    if mode == "dist_lazy":
        dist.init_process_group("nccl", rank=0, world_size=1)
    else:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
p = synth.generate_native(synth.BASELINE_CONFIGS["cfg3"])
cal = ViCalibrator(0).load_problem(p)
if mode in ("shard", "dist_shard"):
    cal.set_shard_rccl(0, 1)
cal.SetStageLimit(3); cal.Solve(); cal.prepare()
cal.run_iterations(10)
torch.cuda.synchronize(); t0 = time.perf_counter(); n, _, _ = cal.run_iterations(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(mode, "ms per iteration %.4f" % (1e3 * dt / n), flush=True)
os._exit(0)
