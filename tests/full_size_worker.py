"""Worker of the frame-sharded full-size tests (launched with torch.distributed.run, all ranks on the one GPU of the test
box; gloo carries the all-reduces of device tensors because RCCL refuses several ranks per device).

  full_size_worker.py <cfg name> <reference npz>

Rank r generates frames [r N / P, (r + 1) N / P) of BASELINE <cfg> itself (native generator, nothing is shipped between the
ranks), solves the sharded problem and compares with the single-process solve the parent test stored in the npz: cost of
every LM iteration and the accept / reject sequence at 1e-7, shared parameters at 1e-7, its own frames' poses."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402
from vicalib_amd import synth  # noqa: E402
from vicalib_amd.lib import ViCalibrator  # noqa: E402
from vicalib_amd.parallel import FrameShardComm, frame_shard  # noqa: E402


def main():
    name, ref_path = sys.argv[1], sys.argv[2]
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base = synth.BASELINE_CONFIGS[name]
    n_total = int(round(base.n_frames * scale))
    lo, hi = frame_shard(n_total, rank, world)
    shard = synth.generate_native(synth.Config(models=base.models, grid=base.grid, n_frames=hi - lo, imu=base.imu, first_frame=lo, extrinsics_prior=base.extrinsics_prior))
    cal = ViCalibrator(0).load_problem(shard)
    cal.SetCalibrateImu(bool(base.imu))
    comm = FrameShardComm(device="cuda:0", stream_ptr=cal.stream())
    cal.set_shard(rank, world, comm)
    cal.Solve()
    ref = np.load(ref_path)
    tg, tr = cal.trace(), ref["trace"]
    assert cal.shared_dim() == int(ref["D"]) + (9 * (world - 1) if base.imu else 0)
    assert len(tg) == len(tr), (len(tg), len(tr))
    np.testing.assert_array_equal(tg[:, 8], tr[:, 8])                      # accept / reject sequence
    np.testing.assert_array_equal(tg[:, 9], tr[:, 9])                      # stages
    np.testing.assert_allclose(tg[:, 1], tr[:, 1], rtol=1e-7)              # cost of every iteration
    for c in range(len(base.models)):
        K, T = cal.GetCamera(c)
        np.testing.assert_allclose(K, ref["K%d" % c], rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(T, ref["T%d" % c], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(cal.GetCameraProjRMSE(), ref["rmse"], rtol=1e-7)
    if base.imu:
        np.testing.assert_allclose(cal.GetBiases(), ref["biases"], rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(cal.GetScaleFactor(), ref["scale"], rtol=1e-7)
        np.testing.assert_allclose(cal.GetGravity(), ref["gravity"], rtol=1e-7, atol=1e-10)
        assert abs(cal.time_offset() - float(ref["time_offset"])) < 1e-9
    sel = np.linspace(0, hi - lo - 1, 64).astype(int)                      # a sample of this rank's frames
    mine = np.array([cal.GetFrame(int(f))[0] for f in sel])
    np.testing.assert_allclose(mine, ref["frames"][lo + sel], rtol=1e-7, atol=1e-9)
    assert comm.calls > 0
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:      # noqa: BLE001
        import traceback
        print("WORKER-FAILURE rank %s: %s\n%s" % (os.environ.get("RANK"), type(e).__name__, (str(e) + "\n" + traceback.format_exc())[:6000]), flush=True)
        os._exit(1)
