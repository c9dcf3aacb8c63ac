"""bench.py's multi-rank branch, executed end to end on ONE GPU (verdict r4 missing #3 / next #3): `--gpus 2 --transport gloo` starts
two ranks through torch.distributed.run, both on device 0, frames of BASELINE cfg3 split between them, the per-iteration all-reduces
through gloo and the FrameShardComm callback.  Asserts the contract of the one JSON line a driver would parse at N > 1.  This is a
functional check of the script's world > 1 code (barriers, MAX over ranks, per-rank gathers, comm object, tear-down order) -- not a
scaling measurement: no run on more than one GPU exists."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, from rank 0
    return json.loads(lines[0]), r.stderr


def test_two_ranks_through_gloo_on_one_gpu():
    out, err = _bench("--gpus", "2", "--steps", "4", "--warmup", "1", "--repeats", "2", "--no-cpu-baseline", "--transport", "gloo")
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1
    assert out["metric"] == "corner_residuals_per_sec" and out["value"] > 0 and out["ms_per_step"] > 0
    assert out["config"]["frames_total"] == 2000 and out["config"]["corners_total"] == 373493      # BASELINE cfg3, both shards together
    assert out["config"]["reduced_dim"] == 29 + 9                                                    # + one separator frame
    comm = out["comm"]
    assert comm["transport"] == "gloo" and comm["communicator_size"] == 2 and comm["world_size_env"] == 2
    assert comm["ranks_agree"] is True and comm["rccl_ranks_seen"] is None      # (torch.distributed's group: every rank at its own rank)
    assert len(comm["per_rank_ms_per_step"]) == 2 and all(x > 0 for x in comm["per_rank_ms_per_step"])
    assert set(comm["allreduce_ms_per_step"]) == {"allreduce(S)", "allreduce(step scalars)"}
    assert all(v > 0 for v in comm["allreduce_ms_per_step"].values())
    assert out["roofline"] is not None and "cpu_baseline" not in out and "secondary" not in out
    # the sharded solve is the single-process solve (tests/test_sharding.py holds it to 1e-7): the complete calibration ran all stages
    assert out["complete_calibration"]["stages"] == 4 and abs(out["final_rmse_px"][0] - 0.1) < 0.01


def test_single_rank_line_is_unchanged_by_the_transport_switch():
    out, _ = _bench("--steps", "4", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--no-secondary")
    assert out["n_gpus"] == 1 and out["comm"] is None and out["config"]["reduced_dim"] == 29
    assert out["roofline"]["kernel"] and 0 < out["roofline"]["frac"] < 1


def test_one_rank_through_the_library_rccl_communicator_reports_what_rccl_saw():
    """VICALIB_AMD_FORCE_SHARD_PATH=1: one rank through the sharded code path with the library's own RCCL communicator -- the
    `comm` object carries what RCCL itself reports (ncclCommCount / ncclCommUserRank through vc_shard_info), not WORLD_SIZE."""
    env_keep = os.environ.get("VICALIB_AMD_FORCE_SHARD_PATH")
    os.environ["VICALIB_AMD_FORCE_SHARD_PATH"] = "1"
    try:
        out, _ = _bench("--steps", "4", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--no-secondary", "--workload", "cfg3", "--frames", "200")
    finally:
        if env_keep is None:
            os.environ.pop("VICALIB_AMD_FORCE_SHARD_PATH", None)
        else:
            os.environ["VICALIB_AMD_FORCE_SHARD_PATH"] = env_keep
    comm = out["comm"]
    assert comm["transport"] == "rccl" and comm["rccl_error"] is None
    assert comm["communicator_size"] == 1 and comm["rccl_ranks_seen"] == [1, 1] and comm["ranks_agree"] is True
    assert comm["allreduce_calls"] > 0
