"""world_size-2 tests of the frame-sharded path."""
import os
import socket
import subprocess
import sys
import pytest

from vicalib_amd.parallel import frame_shard

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(mode, timeout, nproc=2):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "dist_worker.py"), mode]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") >= nproc


def test_frame_partition_is_contiguous_and_complete():
    for n, p in [(500, 8), (7, 3), (10, 4), (1, 1), (50000, 8)]:
        got = []
        for r in range(p):
            lo, hi = frame_shard(n, r, p)
            assert lo <= hi
            got += list(range(lo, hi))
        assert got == list(range(n))


def test_two_rank_reduced_system_allreduce_gloo_cpu():
    _run("cpu", 600)


@pytest.mark.gpu
def test_two_rank_sharded_solve_gloo_on_one_gpu():
    _run("gpu", 900)


@pytest.mark.gpu
def test_two_rank_sharded_visual_inertial_solve_on_one_gpu():
    """IMU chain across the shard boundary: the second rank's first frame is a separator in the reduced system."""
    _run("gpu_imu", 900)


@pytest.mark.gpu
def test_three_rank_sharded_visual_inertial_solve_on_one_gpu():
    """The middle rank holds a separator (its first frame) and a ghost (the last rank's first frame)."""
    _run("gpu_imu", 900, nproc=3)
