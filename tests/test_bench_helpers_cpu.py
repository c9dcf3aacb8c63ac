"""bench.py's host-side choices (no GPU): which BASELINE configuration a GPU count runs, how its frames are cut into ranks, and that a
multi-GPU request without a launcher re-executes itself under torch.distributed.run with one rank per GPU."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpu_count_selects_the_baseline_configuration():
    b = _bench()
    assert [b.default_workload(n) for n in (1, 2, 4, 8)] == ["cfg3", "cfg3", "cfg4", "cfg5"]      # BASELINE.json configs 3, 3 (split), 4, 5
    assert set(b.WORKLOADS) == {"cfg2", "cfg3", "cfg4", "cfg5"}


@pytest.mark.parametrize("n_frames,world", [(2000, 1), (2000, 2), (10000, 4), (50000, 8), (17, 4), (9, 8)])
def test_frame_shards_are_contiguous_and_cover_every_frame(n_frames, world):
    from vicalib_amd.parallel import frame_shard
    cuts = [frame_shard(n_frames, r, world) for r in range(world)]
    assert cuts[0][0] == 0 and cuts[-1][1] == n_frames
    for (lo, hi), (lo2, _) in zip(cuts, cuts[1:]):
        assert hi == lo2 and hi > lo
    sizes = [hi - lo for lo, hi in cuts]
    assert max(sizes) - min(sizes) <= 1


def test_multi_gpu_request_relaunches_itself_with_one_rank_per_gpu(monkeypatch):
    b = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(b.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") is not None
