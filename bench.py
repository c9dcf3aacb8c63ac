#!/usr/bin/env python
"""bench.py -- LM-iteration throughput of the calibration hot path on MI355X.

Workload per GPU count (BASELINE.json `configs`; the metric is not quoted on one of them, so N = 1 runs the largest
single-GPU configuration):
  N = 1   cfg3: mono kb4 + IMU (biases, scale factors, time offset), small grid, 2000 frames
  N = 2   cfg3, its frames sharded over the two ranks
  N = 4   cfg4: 4 x poly3 + IMU, large grid (900 dots), 10 000 frames sharded over 4 ranks
  N = 8   cfg5: 8 cameras fov/kb4 + IMU, small grid, 50 000 frames sharded over 8 ranks
(--workload cfg2|cfg3|cfg4|cfg5 overrides; cfg2 = stereo fov, 500 frames, no IMU -- last round's headline.)

A "step" is one Levenberg-Marquardt iteration of the real solver in the FINAL stage of the reference's schedule
(vicalibrator.h:977-1000: camera-to-IMU transform, intrinsics, gravity, biases, scale factors and time offset all free):
IMU weight update (UpdateImuWeights) + Jacobian sweep over all corners + IMU Jacobians + elimination of the frame chain +
reduced solve + back-substitution + manifold update + residual sweeps of the trial point + accept/reject.  The stages before
it run first (untimed); complete final-stage solves are then run back to back from that state until exactly K iterations are
done.  N > 1: one process per GPU, frames sharded, one all-reduce of the reduced system and one of the step scalars per
iteration over RCCL.  `python bench.py --gpus N` launches its own N ranks when it is not already running under
torch.distributed.run.  Prints ONE JSON line on rank 0.
`--transport gloo` (or VICALIB_AMD_BENCH_TRANSPORT=gloo) runs the same multi-rank branch with the ranks sharing the visible device(s)
and the all-reduces through torch.distributed's gloo backend: what tests/test_bench_multirank_gpu.py executes on a one-GPU box.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    "cfg2": dict(name="BASELINE cfg2: stereo fov,fov, small grid 19x10, 500 frames, intrinsics+extrinsics, no IMU"),
    "cfg3": dict(name="BASELINE cfg3: mono kb4 + IMU (biases, scale, time offset), small grid 19x10, 2000 frames"),
    "cfg4": dict(name="BASELINE cfg4: 4-camera poly3 rig + IMU, large grid 25x36, 10000 frames"),
    "cfg5": dict(name="BASELINE cfg5: 8-camera fov/kb4 rig + IMU, small grid 19x10, 50000 frames"),
}
# HBM bytes per launch from rocprofv3 --pmc passes on the N = 1 default workload (profiles/, see README there)
PMC_TRAFFIC = {}
TRAFFIC_SOURCE = ("profiles/pmc_traffic_latest.json: HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc passes of this command "
                  "taken on an earlier run and committed) -- NOT measured in this run")
try:
    with open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")) as _f:
        PMC_TRAFFIC = json.load(_f)
except (OSError, ValueError):
    pass


def default_workload(n):
    return "cfg3" if n <= 2 else ("cfg4" if n <= 4 else "cfg5")


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="auto", choices=["auto"] + sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="override the workload's total frame count")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary (cfg2 / at-scale) measurements")
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps iterations each; ms_per_step is their median")
    ap.add_argument("--transport", default=os.environ.get("VICALIB_AMD_BENCH_TRANSPORT", "auto"), choices=["auto", "rccl", "gloo"],
                    help="multi-rank runs: rccl = one GPU per rank, the library's own RCCL communicator (auto: the same); gloo = torch.distributed "
                         "gloo + the FrameShardComm callback, ranks share the visible device(s) -- executes the whole multi-rank branch of this "
                         "script on a one-GPU box (a functional check, not a scaling measurement)")
    args = ap.parse_args()
    gloo = args.transport == "gloo"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: one rank per GPU over RCCL
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    from vicalib_amd import synth
    from vicalib_amd.lib import ViCalibrator
    from vicalib_amd.parallel import frame_shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # gloo transport: the ranks share whatever devices there are (all of them device 0 on a one-GPU box)
    device = local_rank % torch.cuda.device_count() if gloo else local_rank
    torch.cuda.set_device(device)
    # VICALIB_AMD_FORCE_SHARD_PATH=1 (test hook): one rank, but through the sharded code path (split kernels + RCCL)
    force_shard = world == 1 and os.environ.get("VICALIB_AMD_FORCE_SHARD_PATH") == "1"
    if world > 1 or force_shard:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    # the bench's own small collectives (agreement on the transport, MAX of the block times, gathers for the JSON line)
    coll_dev = "cpu" if gloo else "cuda"

    wl = default_workload(world) if args.workload == "auto" else args.workload
    base = synth.BASELINE_CONFIGS[wl]
    n_total = args.frames or base.n_frames
    lo, hi = frame_shard(n_total, rank, world)
    cfg = synth.Config(models=base.models, grid=base.grid, n_frames=hi - lo, imu=base.imu, first_frame=lo, extrinsics_prior=base.extrinsics_prior)
    prob = synth.generate_native(cfg)
    vi = bool(base.imu)

    rccl_errors = []      # native RCCL path failed on this rank: the texts go into the JSON line's comm object
    # ONE library communicator per process, lent to the three calibrators below (a first 8-GPU run must not also be the debut of
    # three ncclCommInitRank / two ncclCommDestroy in a row whose relative order across the ranks nobody has seen)
    shared = {"comm": None, "kind": None}

    def transport():
        """Decided once, collectively: 'rccl' (shared library communicator), 'torch' (callback on torch.distributed) or 'none'."""
        if shared["kind"] is not None:
            return shared["kind"]
        if world == 1 and not force_shard:
            shared["kind"] = "none"
            return "none"
        kind = "torch"
        if not gloo and os.environ.get("VICALIB_AMD_SHARD_COMM", "rccl") == "rccl":
            from vicalib_amd.lib import ShardComm
            ok = 1
            try:
                shared["comm"] = ShardComm(device, rank, world)
            except Exception as e:      # noqa: BLE001
                # the exception text carries vc_last_error(): which RCCL call failed and RCCL's own error string
                ok = 0
                rccl_errors.append(str(e))
            # every rank must end up on the same transport: one failed rank sends all of them to the torch.distributed callback
            if world > 1:
                flag = torch.tensor([ok], device=coll_dev, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                all_ok = int(flag.item())
            else:
                all_ok = ok
            if all_ok:
                kind = "rccl"
            else:
                if ok:
                    rccl_errors.append("native RCCL communicator dropped: another rank failed to create its own")
                    shared["comm"].close(); shared["comm"] = None
                else:
                    print("bench[rank %d]: native RCCL path unavailable (%s); using the torch.distributed callback" % (rank, rccl_errors[-1]), file=sys.stderr)
        shared["kind"] = kind
        return kind

    def attach(c):
        kind = transport()
        if kind == "rccl":
            c.set_shard_comm(shared["comm"])
        elif kind == "torch":
            from vicalib_amd.parallel import FrameShardComm
            c.set_shard(rank, world, FrameShardComm(device="cuda:%d" % device, stream_ptr=c.stream()))
        return "gloo" if (kind == "torch" and gloo) else kind

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- complete calibrations: the metric's "final RMS reproj err" and the wall time of the whole schedule.  The first solve of a
    # process also pays for code-object loading and first-touch allocations (~20 ms); the second one is what a caller sees from
    # then on.
    full = None
    for rep in range(2):
        cal2 = ViCalibrator(device).load_problem(prob)
        cal2.SetCalibrateImu(vi)
        attach(cal2)
        barrier(); t0 = time.perf_counter()
        cal2.Solve()
        barrier(); t_full = time.perf_counter() - t0
        if rep == 0:
            t_first = t_full
            continue
        rmse = [float(x) for x in cal2.GetCameraProjRMSE()]
        full_trace = cal2.trace()
        full = {"seconds": t_full, "seconds_first_in_process": t_first, "lm_iterations": int(np.sum(full_trace[:, 0] > 0)),
                "stages": int(full_trace[-1, 9]) + 1 if len(full_trace) else 0}
        if vi:
            gt = prob.imu_gt
            full["time_offset_error_s"] = abs(cal2.time_offset() - gt["time_offset"])
            full["gyro_bias_error"] = float(np.abs(cal2.GetBiases()[:3] - gt["bg"]).max())
    cal2.close()
    del cal2

    # ---- the timed loop: LM iterations of the final stage --------------------------------------------------------
    cal = ViCalibrator(device).load_problem(prob)
    cal.SetCalibrateImu(vi)
    comm_kind = attach(cal)
    if vi:
        cal.SetStageLimit(3)        # stages A (vision), B (rotation), C (+ translation, gravity, biases) run; D is set up
        cal.Solve()
    cal.prepare()
    n_obs_local = cal.num_observations()
    cal.run_iterations(args.warmup)
    # the timed region: --repeats blocks of exactly --steps LM iterations, every block bracketed by barrier + synchronize on both
    # sides and reduced with MAX over the ranks; ms_per_step / value are the MEDIAN block's (one block is only a few ms: a single
    # one is at the mercy of the host), min and max are printed next to it
    block_dt = []; block_local = []
    done = jac_sweeps = res_sweeps = 0
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        done, jac_sweeps, res_sweeps = cal.run_iterations(args.steps)
        barrier()
        d = time.perf_counter() - t0
        block_local.append(d)
        if world > 1:
            tt = torch.tensor([d], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        block_dt.append(d)
    dt = float(np.median(block_dt))
    per_rank_ms = None
    if world > 1:
        mine = torch.tensor([1e3 * float(np.median(block_local)) / max(done, 1)], device=coll_dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [float(x.item()) for x in allr]
        no = torch.tensor([float(n_obs_local)], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(no)
        n_obs_total = int(no.item())
    else:
        n_obs_total = n_obs_local
    allreduce_calls = cal.allreduce_calls() if comm_kind == "rccl" else None
    # what the transport itself says about its size: RCCL's ncclCommCount / ncclCommUserRank (vc_shard_info) or torch.distributed's group
    shard_info = cal.shard_info()
    if comm_kind == "rccl":
        ranks_seen, rank_seen = shard_info["rccl_ranks"], shard_info["rccl_rank"]
    elif world > 1:
        ranks_seen, rank_seen = dist.get_world_size(), dist.get_rank()
    else:
        ranks_seen, rank_seen = (1, 0) if comm_kind != "none" else (None, None)
    if world > 1:
        seen = torch.tensor([float(ranks_seen), float(rank_seen == rank)], device=coll_dev, dtype=torch.float64)
        lo_hi = [seen.clone(), seen.clone()]
        dist.all_reduce(lo_hi[0], op=dist.ReduceOp.MIN); dist.all_reduce(lo_hi[1], op=dist.ReduceOp.MAX)
        ranks_seen_min, ranks_seen_max, ranks_agree = int(lo_hi[0][0].item()), int(lo_hi[1][0].item()), bool(lo_hi[0][1].item() == 1.0)
    else:
        ranks_seen_min = ranks_seen_max = ranks_seen
        ranks_agree = rank_seen == rank if rank_seen is not None else None

    # ---- in-loop kernel durations (HIP events on the calibrator's stream around every launch group of the same loop) ----
    cal.set_kernel_timing(True)
    cal.run_iterations(args.steps)
    kt = cal.kernel_timing()
    cal.set_kernel_timing(False)
    n_tiles = cal.num_tiles()
    n_frames_local = len(prob.frame_time)
    n_imu_blocks = max(n_frames_local - 1, 0)
    nk = {0: 5, 1: 6, 2: 7, 3: 8, 4: 4, 5: 10}
    kc_mean = float(np.mean([nk[m] for m in prob.cam_model]))
    # Jacobian sweep 18 B/corner + per tile [64 B in + 8*(21 + 6 + 6*S_c + 1) B out], S_c = 6 + K_c; ~1.05 kflop/corner
    bytes_jac = 18.0 * n_obs_local + n_tiles * (64 + 8 * (28 + 6 * (6 + kc_mean)))
    flops_jac = 1050.0 * n_obs_local
    bytes_res = 18.0 * n_obs_local + n_tiles * (64 + 8)
    imu_samples = len(prob.imu_t) if vi else 0
    # IMU sweep in delta form, counted as the kernels run it:
    #   k_imu_block: per block 7 parameter directions (6 gyro parameters + the time offset; the accelerometer partials are sums of the stage
    #                rotations' columns) x one RK4 step from the identity state per sample interval with CLOSED-FORM partials (round 6:
    #                values + one rotation tangent, 612 fp64 instructions per interval in the kernel's ISA, ~1100 flop with FMA = 2; rounds
    #                2-5 ran dual numbers, ~1700 flop) + the ordered product of the interval deltas (pairwise append, then a three-level scan
    #                over eight lanes: ~4 appends of ~280 flop per lane); in 56 B per sample + the IMU parameters, out one 154-double
    #                record per block
    #   k_imu_jac:   per block 30 lanes put the delta on the start state and run the residual's tail (~900 flop each), then the 33 x 33
    #                weighted J^T J (9 x 33 x 33 x 2 flop); in the record, two states, the 9 x 9 weight; out 33 x 33 + 33 + 1 doubles
    n_meas = imu_samples / max(n_imu_blocks, 1) + 2
    flops_delta = n_imu_blocks * 7 * ((n_meas - 1) * 1100.0 + 8 * 4 * 280.0)
    bytes_delta = imu_samples * 56.0 + n_imu_blocks * (2 * 56 + 15 * 8 + 154 * 8)
    flops_imu = n_imu_blocks * (30 * 900.0 + 9 * 33 * 33 * 2.0)
    bytes_imu = n_imu_blocks * (154 * 8 + 2 * 12 * 8 + 81 * 8 + (33 * 33 + 34) * 8.0)
    # weight update: per RK4 step 16 sensitivity columns through 4 stages (~9 kflop) + Sigma <- F Sigma F^T + G R G^T (~5.2 kflop)
    flops_w = n_imu_blocks * n_meas * 14200.0
    bytes_w = n_imu_blocks * (n_meas * 56 + 160 + 2 * 81 * 8)
    # The weight update, per RK4 step as the reference's hand-derived chain does it -- 16 sensitivity columns through 4 stages
    # (~9 kflop) + Sigma <- F Sigma F^T + G R G^T dense (~5.2 kflop) -- is the unit the roofline fraction is quoted on (same unit
    # as round 2).  The kernel itself does less: only the 7 columns that pass through the quaternion blocks are propagated and the
    # maps keep their block form (~8.5 kflop per interval + ~3 kflop per block for the projection): `executed_flops_estimate`.
    D = cal.shared_dim()
    N = n_frames_local
    # chain kernels, per pass: every frame is eliminated once -- 9 x 9 Cholesky (9^3 / 3) + forward solves of its
    # D + 19 border / coupling columns (81 each) + the rank-18 update of D + 28 columns (2 x 9 x 18 each); image in + out
    flops_fwd = N * (243.0 + 81.0 * (D + 19) + 324.0 * (D + 28)); bytes_fwd = N * 2.0 * 9 * (D + 28) * 8
    flops_back = N * (18.0 * D + 405.0); bytes_back = N * 9.0 * (D + 28) * 8 + N * 9 * 8
    flops_gram = N * 9.0 * (D + 1) * (D + 2); bytes_gram = N * 9.0 * (D + 1) * 8
    flops_init = n_tiles * 2500.0 + N * 600.0; bytes_init = n_tiles * 272.0 * 8 + N * (738.0 * 8 + 9.0 * (D + 28) * 8)
    flops_red = D ** 3 / 3.0 + 4.0 * D * D; bytes_red = 2.0 * (D * D + 3 * D) * 8
    algo = {"k_reproj_jac": ("mfma", flops_jac, bytes_jac), "k_trial": ("mfma", flops_jac, bytes_jac + n_tiles * 8 * 96 + n_frames_local * 8 * 48),
            "k_imu_jac": ("fp64-valu", flops_imu, bytes_imu), "k_imu_weights": ("fp64-valu", flops_w, bytes_w),
            "k_imu_block": ("fp64-valu", flops_delta, bytes_delta),
            "k_chain_init": ("hbm", flops_init, bytes_init), "k_chain_fwd": ("fp64-valu", flops_fwd, bytes_fwd),
            "k_chain_back": ("fp64-valu", flops_back, bytes_back), "k_chain_gram": ("mfma", flops_gram, bytes_gram),
            "k_reduced": ("fp64-valu", flops_red, bytes_red)}
    if vi and "k_chain_gram" not in kt and "k_chain_fwd" in kt:
        # early Gram: the Gram sums ride in the top level's launch -- the group k_chain_fwd carries their flops and bytes
        algo["k_chain_fwd"] = ("fp64-valu", flops_fwd + flops_gram, bytes_fwd + bytes_gram)
    if vi and "k_chain_init" not in kt and "k_chain_fwd" in kt:
        # the chain assembly folded into the bottom level (k_chain_l0): its flops and the bytes it reads ride in the group as well; the
        # unsolved images no longer travel (9 (D + 28) doubles per frame written and read back)
        _, fl, by = algo["k_chain_fwd"]
        algo["k_chain_fwd"] = ("fp64-valu", fl + flops_init, by + bytes_init - N * 2.0 * 9 * (D + 28) * 8)
    # launch groups that run on the second stream next to the critical path (vc_pass.cpp: enqueue_pass)
    overlapped = {"k_imu_weights", "k_imu_block(trial)", "k_imu_block", "k_imu_jac"} if vi else set()
    if vi and os.environ.get("VICALIB_AMD_JAC_STREAM2", "1") != "0" and os.environ.get("VICALIB_AMD_OVERLAP_WEIGHTS", "1") != "0":
        overlapped.add("k_imu_jac(trial)")      # beside the vision sweep of the trial point (round 3)
    kernels = {}
    for name, (cnt, avg_ms) in kt.items():
        e = {"launch_groups": cnt, "avg_ms": avg_ms, "ms_per_step": avg_ms * cnt / max(done, 1),
             "stream": "B (overlapped with the chain solve)" if name in overlapped else "A (critical path)"}
        base_name = name.replace("(trial)", "")
        if base_name in algo:
            bound, fl, by = algo[base_name]
            e.update({"bound": bound, "algorithmic_flops": fl, "algorithmic_bytes": by, "tflops": fl / (avg_ms * 1e-3) / 1e12,
                      "fp64_frac": fl / (avg_ms * 1e-3) / 78.6e12, "hbm_gbs": by / (avg_ms * 1e-3) / 1e9, "hbm_frac": by / (avg_ms * 1e-3) / 8e12})
            if base_name == "k_imu_weights":
                e["executed_flops_estimate"] = n_imu_blocks * (n_meas * 8500.0 + 3000.0)
        kernels[name] = e
    def pmc_traffic(group):
        """HBM bytes of one launch group from the committed PMC passes (profiles/pmc_traffic_latest.json): a single kernel's
        per-launch figure, or -- the chain elimination's levels -- everything its kernels moved per LM pass."""
        t = PMC_TRAFFIC.get(wl, {})
        base = group.replace("(trial)", "")
        if base == "k_chain_fwd":
            parts = [t.get("k_chain_fwd@pass"), t.get("k_chain_fwd2@pass"), t.get("k_chain_top_gram@pass", t.get("k_chain_top_gram")),
                     t.get("k_chain_l0@pass", t.get("k_chain_l0"))]
            return sum(x for x in parts if x) if any(parts) else t.get(base)
        if base == "k_chain_back":
            return t.get("k_chain_back@pass", t.get(base))
        return t.get(base)

    # the dominant kernel = the launch group with the largest share of the step among ALL groups that carry a model
    # (a group of several launches -- the levels of k_chain_fwd -- counts as one: per pass its flops / its time)
    modelled = [k for k in kernels if k.replace("(trial)", "") in algo]
    dom = max(modelled, key=lambda k: kernels[k]["ms_per_step"]) if modelled else None
    roofline = None
    if dom:
        bound, fl, by = algo[dom.replace("(trial)", "")]
        e = kernels[dom]
        hbm = bound == "hbm"
        roofline = {"kernel": dom, "bound": bound, "achieved": e["hbm_gbs"] if hbm else e["tflops"], "peak": 8000.0 if hbm else 78.6,
                    "unit": "GB/s" if hbm else "TFLOP/s", "frac": e["hbm_frac"] if hbm else e["fp64_frac"],
                    "traffic": pmc_traffic(dom) if world == 1 and not args.frames else None, "traffic_source": TRAFFIC_SOURCE, "avg_ms": e["avg_ms"],
                    "algorithmic_flops": fl, "algorithmic_bytes": by, "hbm_gbs": e["hbm_gbs"], "hbm_frac": e["hbm_frac"],
                    "launches_in_group": ("one launch per level of the partitioned elimination" + ("; the top level's launch also forms the chain's Gram sums" if "chain_fwd" in dom and "k_chain_gram" not in kt else "")) if "chain_fwd" in dom or "chain_back" in dom else 1,
                    "timing": "HIP events around every launch group of this kernel inside the timed LM loop (decisions live)"}
    # the step's critical path: the groups of the main stream in launch order, from the same in-loop events
    crit = [(k, e["ms_per_step"]) for k, e in kernels.items() if e["stream"].startswith("A")]
    crit_sum = sum(x[1] for x in crit) or 1.0
    critical_path = {"entries": [{"kernel": k, "us_per_step": 1e3 * v, "share": v / crit_sum} for k, v in sorted(crit, key=lambda x: -x[1])],
                     "sum_us_per_step": 1e3 * crit_sum,
                     "note": "main-stream launch groups, from an instrumented repeat of the loop (the timing events are barriers of their own on the "
                             "main stream: with the device-flag hand-overs the sum can exceed ms_per_step); not in the sum: the first pass of "
                             "every solve (both sweeps at the accepted state) and the host's feeding of the passes"}
    nm = "k_reproj_jac(trial)" if "k_reproj_jac(trial)" in kernels else ("k_trial" if "k_trial" in kernels else None)
    roofline_sweep = None
    if nm:
        e = kernels[nm]
        roofline_sweep = {"kernel": nm + " (residual + Jacobian + tile normal equations sweep)", "avg_ms": e["avg_ms"], "tflops": e["tflops"],
                          "fp64_frac": e["fp64_frac"], "hbm_gbs": e["hbm_gbs"], "hbm_frac": e["hbm_frac"],
                          "traffic": PMC_TRAFFIC.get(wl, {}).get(nm.replace("(trial)", "")) if world == 1 and not args.frames else None,
                          "traffic_source": TRAFFIC_SOURCE}
    # the residual-only sweep (k_reproj_res; replaces Problem::Evaluate at vicalibrator.h:959-971, :873-898).  The LM
    # loop no longer runs it (residual_sweeps: 0 -- the trial point is judged by the Jacobian sweeps themselves); the RMSE and outlier
    # passes do.  Timed stand-alone, launches back to back on the calibrator's stream (vc_time_kernels).
    roofline_res = None
    if world == 1 and n_tiles > 0:
        jac_alone_ms, res_ms = cal.time_kernels(50)
        roofline_res = {"kernel": "k_reproj_res (residual-only sweep: RMSE / outlier passes)", "bound": "hbm", "avg_ms": res_ms,
                        "algorithmic_bytes": bytes_res, "achieved": bytes_res / (res_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": bytes_res / (res_ms * 1e-3) / 8e12, "corners_per_sec": n_obs_local / (res_ms * 1e-3),
                        "traffic": PMC_TRAFFIC.get(wl, {}).get("k_reproj_res") if not args.frames else None, "traffic_source": TRAFFIC_SOURCE,
                        "timing": "50 stand-alone launches back to back between two HIP events on the calibrator's stream",
                        "jacobian_sweep_alone_ms": jac_alone_ms, "jacobian_sweep_alone_fp64_frac": flops_jac / (jac_alone_ms * 1e-3) / 78.6e12}
    comm = None
    if world > 1 or force_shard:
        ar = {k: kernels[k] for k in kernels if k.startswith("allreduce")}
        # communicator_size: the smallest size any rank's transport reports for itself (== WORLD_SIZE when every rank is in ONE
        # communicator of all ranks); rccl_ranks_seen: [min, max] over the ranks of ncclCommCount; ranks_agree: every rank sits at the
        # rank it was launched as
        comm = {"transport": comm_kind, "communicator_size": ranks_seen_min, "world_size_env": world,
                "rccl_ranks_seen": [ranks_seen_min, ranks_seen_max] if comm_kind == "rccl" else None, "ranks_agree": ranks_agree,
                "allreduce_calls": allreduce_calls, "per_rank_ms_per_step": per_rank_ms,
                "allreduce_ms_per_step": {k: v["ms_per_step"] for k, v in ar.items()},
                "payload_doubles": {"allreduce(S)": D * D + 3 * D + 2, "allreduce(step scalars)": world * 8},
                "rccl_error": rccl_errors[0] if rccl_errors else None}

    out = None
    if rank == 0:
        iters_per_s = done / dt
        value = n_obs_total * iters_per_s
        out = {
            "metric": "corner_residuals_per_sec", "value": value, "unit": "corner-residuals/s (LM iterations x corners)",
            "n_gpus": world, "steps": done, "warmup": args.warmup, "ms_per_step": 1e3 * dt / done, "higher_is_better": True,
            "timing": {"blocks": len(block_dt), "steps_per_block": done, "ms_per_step_median": 1e3 * dt / done,
                       "ms_per_step_min": 1e3 * min(block_dt) / done, "ms_per_step_max": 1e3 * max(block_dt) / done},
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": WORKLOADS[wl]["name"] + (" [frames overridden: %d]" % n_total if args.frames else ""),
                       "stage": "final stage of the schedule (all camera / IMU parameters free)" if vi else "vision-only solve",
                       "frames_total": n_total, "corners_total": n_obs_total, "tiles_rank0": n_tiles, "imu_samples_rank0": imu_samples,
                       "reduced_dim": cal.shared_dim(),
                       "parallelism": "frames sharded x%d, all-reduce of reduced system per LM iteration (%s)" % (world, comm_kind)},
            "lm_iters_per_sec": iters_per_s, "jacobian_sweeps": jac_sweeps, "residual_sweeps": res_sweeps, "allreduce_calls": allreduce_calls,
            "final_rmse_px": rmse, "complete_calibration": full, "roofline": roofline, "roofline_jacobian_sweep": roofline_sweep, "roofline_residual_sweep": roofline_res,
            "critical_path": critical_path, "comm": comm, "kernels_in_loop": kernels,
        }
        if not args.no_secondary and world == 1 and not force_shard:
            out["secondary"] = secondary(device)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl, prob)
    if rank == 0:
        # the one JSON line comes last: RCCL writes a version banner to the C-level stdout, flush that first
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:      # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    if world > 1 or force_shard:
        # leave together, and without the interpreter's teardown order deciding which of the two RCCL users (torch's process
        # group, the library's one communicator) is torn down first: calibrators, then the shared communicator, then the barrier
        try:
            cal.close()
            if shared["comm"] is not None:
                shared["comm"].close()
            dist.barrier()
        except Exception:      # noqa: BLE001
            pass
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


def secondary(device):
    """(a) BASELINE cfg4 at full size on this one GPU (4 x poly3 + IMU, 900-dot grid, 10 000 frames, 21.7 M corners): the same loop
    where the chip is filled -- what the per-kernel roofline fractions look like away from cfg3's launch-latency regime;
    (b) last round's vision-only headline (cfg2) and the same rig with 10x the frames.  Measured the same way (complete solves
    back to back, in-loop kernel timing).  Informational: `value` above is the N = 1 workload's number."""
    from vicalib_amd import synth
    from vicalib_amd.lib import ViCalibrator
    out = {}
    base = synth.BASELINE_CONFIGS["cfg4"]
    p = synth.generate_native(base)
    cal = ViCalibrator(device).load_problem(p)
    cal.SetStageLimit(3); cal.Solve(); cal.prepare()
    n = cal.num_observations()
    cal.run_iterations(3)
    t0 = time.perf_counter(); done, _, _ = cal.run_iterations(12); dt = time.perf_counter() - t0
    cal.set_kernel_timing(True); cal.run_iterations(12); kt = cal.kernel_timing(); cal.set_kernel_timing(False)
    jac = kt.get("k_reproj_jac(trial)", (0, float("nan")))[1]
    jac_alone_ms, res_ms = cal.time_kernels(10)
    n_tiles4 = cal.num_tiles()
    bytes_res4 = 18.0 * n + n_tiles4 * (64 + 8)
    out["cfg4_one_gpu"] = {"roofline_residual_sweep": {"kernel": "k_reproj_res", "avg_ms": res_ms, "algorithmic_bytes": bytes_res4, "hbm_gbs": bytes_res4 / (res_ms * 1e-3) / 1e9,
                                                       "hbm_frac": bytes_res4 / (res_ms * 1e-3) / 8e12, "traffic": PMC_TRAFFIC.get("cfg4", {}).get("k_reproj_res"),
                                                       "traffic_source": TRAFFIC_SOURCE, "jacobian_sweep_alone_ms": jac_alone_ms,
                                                       "jacobian_sweep_alone_fp64_frac": 1050.0 * n / (jac_alone_ms * 1e-3) / 78.6e12},
                           "frames": len(p.frame_time), "corners": n, "reduced_dim": cal.shared_dim(), "ms_per_lm_iteration": 1e3 * dt / done,
                           "corner_residuals_per_sec": n * done / dt, "kernels_in_loop_us": {k: 1e3 * v[1] for k, v in kt.items()},
                           "jacobian_sweep_tflops": 1050.0 * n / (jac * 1e-3) / 1e12, "jacobian_sweep_fp64_frac": 1050.0 * n / (jac * 1e-3) / 78.6e12,
                           "jacobian_sweep_hbm_frac": 18.0 * n / (jac * 1e-3) / 8e12}
    out["cfg4_one_gpu"] = dict(sorted(out["cfg4_one_gpu"].items()))
    del cal, p
    # (c) the image front-end (dot detection ahead of the solver; vc_detect.hip): images per second through the C ABI --
    # host image in, dot centres out (upload, integral image, adaptive threshold, labelling, statistics, one conic fit per dot, download).
    # Untuned kernels of a few hundred wavefronts each; informational.
    try:
        from vicalib_amd.lib import ConicDetector
        det_out = {}
        for (w, h, nx, ny, r) in ((640, 480, 13, 9, 9.0), (1280, 960, 26, 18, 9.0)):
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.full((h, w), 230.0)
            for j in range(ny):
                for i in range(nx):
                    cx, cy = (i + 1.0) * w / (nx + 1.0), (j + 1.0) * h / (ny + 1.0)
                    rr = r if (i * 7 + j * 3) % 3 else 0.66 * r
                    x0, x1, y0, y1 = int(cx - rr - 2), int(cx + rr + 3), int(cy - rr - 2), int(cy + rr + 3)
                    d2 = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2
                    img[y0:y1, x0:x1] = np.where(d2 < rr * rr, 25.0, img[y0:y1, x0:x1])
            img8 = img.astype(np.uint8)
            det = ConicDetector(w, h, device); det.set_params()
            found = det.find(img8)
            reps = 40
            t0 = time.perf_counter()
            for _ in range(reps):
                det.find(img8)
            dt = time.perf_counter() - t0
            det_out["%dx%d" % (w, h)] = {"dots_rendered": nx * ny, "dots_found": int(len(found)), "images_per_sec": reps / dt, "ms_per_image": 1e3 * dt / reps}
            det.close()
        out["detector"] = det_out
    except Exception as e:      # noqa: BLE001
        out["detector"] = {"error": str(e)}
    for tag, frames in (("cfg2", 500), ("cfg2_x10", 5000)):
        p = synth.generate_native(synth.Config(models=("fov", "fov"), grid="small", n_frames=frames, imu=False))
        cal = ViCalibrator(device).load_problem(p); cal.SetCalibrateImu(False); cal.prepare()
        n = cal.num_observations()
        cal.run_iterations(20)
        t0 = time.perf_counter(); done, _, _ = cal.run_iterations(200); dt = time.perf_counter() - t0
        cal.set_kernel_timing(True); cal.run_iterations(100); kt = cal.kernel_timing(); cal.set_kernel_timing(False)
        tr = kt.get("k_trial", (0, float("nan")))[1]
        out[tag] = {"frames": frames, "corners": n, "us_per_lm_iteration": 1e6 * dt / done, "corner_residuals_per_sec": n * done / dt,
                    "kernels_in_loop_us": {k: 1e3 * v[1] for k, v in kt.items()},
                    "k_trial_tflops": 1050.0 * n / (tr * 1e-3) / 1e12, "k_trial_fp64_frac": 1050.0 * n / (tr * 1e-3) / 78.6e12}
    return out


def cpu_baseline(wl, prob):
    """The CPU oracle (a port of the reference's autodiff + solver path: Ceres cannot be built here) timed on this host on a
    bounded sample of the same workload: complete LM-iteration work units (weight update, dual-number Jacobian sweep, IMU
    blocks, block solve, cost sweep) of the same stage on the first frames of the same problem, 4 threads (the reference's
    num_threads, vicalibrator.h:141)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    vi = prob.imu_t is not None
    threads = 4
    n_sub = min(200 if vi else 100, len(prob.frame_time))
    tf, tc, off, ids, pix = prob.flat

    def build(nthreads, fast=False):
        orc = ol.Oracle(fast=fast)       # fast: libvco_fast.so (closed-form Jacobians borrowed from the product; timed only)
        for c, m in enumerate(prob.cam_model):
            orc.add_camera(m, prob.cam_K_gt[c], prob.cam_T_ck_gt[c])
        for n in range(n_sub):
            orc.add_frame(prob.frame_T_wk_gt[n], prob.frame_time[n])
        nobs = 0
        for k in range(len(tf)):
            if tf[k] < n_sub:
                sl = slice(off[k], off[k + 1])
                orc.add_observations(int(tf[k]), int(tc[k]), prob.grid_points[ids[sl]], pix[sl]); nobs += int(off[k + 1] - off[k])
        if vi:
            sel = prob.imu_t <= prob.frame_time[n_sub - 1] + 0.1
            orc.add_imu(prob.imu_gyro[sel], prob.imu_accel[sel], prob.imu_t[sel])
            orc.set_options(calibrate_imu=True, num_threads=nthreads)
            orc.set_flags(True, True, False, True)          # final stage: everything free
            orc.prepare(vis_mult=4, imu_mult=3)
        else:
            orc.set_options(calibrate_imu=False, num_threads=nthreads)
            orc.prepare(vis_mult=1)
        return orc, nobs

    def run(orc, budget):
        orc.time_iterations(1)
        iters = 0; t = 0.0
        while t < budget and iters < 400:
            t += orc.time_iterations(2); iters += 2
        return iters, t

    orc, nobs = build(threads)
    iters, t = run(orc, 12.0)
    # "sweeps also timed alone" -- the residual-only sweep (Problem::Evaluate: every residual block, reprojection and IMU, no
    # Jacobians) and the residual + Jacobian sweep (dual numbers, normal-equation blocks), same sample, same threads
    def sweep_alone(fn, budget=2.0):
        fn(); k = 0; t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget and k < 50:
            fn(); k += 1
        return k / (time.perf_counter() - t0)
    res_rate = sweep_alone(orc.evaluate_cost)
    jac_rate = sweep_alone(orc.linearize)
    sweeps_alone = {"residual_only_corner_residuals_per_sec": nobs * res_rate, "residual_jacobian_corner_residuals_per_sec": nobs * jac_rate, "cores": threads,
                    "what": "one sweep over all residual blocks of the sample (reprojection + IMU); the Jacobian leg forms the dual-number blocks and the normal-equation "
                            "blocks (and copies them out: oracle linearize())"}
    ncpu = os.cpu_count() or threads
    all_cores = best = None
    if ncpu > threads:
        nt = min(ncpu, 64)
        orc2, _ = build(nt)
        it2, t2 = run(orc2, 5.0)
        all_cores = {"value": nobs * it2 / t2, "cores": nt}
        # the same work with closed-form reprojection Jacobians (oracle/vco_fast.h) instead of forward duals
        orc3, _ = build(nt, fast=True); orc3.set_closed_form(True)
        it3, t3 = run(orc3, 5.0)
        best = {"value": nobs * it3 / t3, "cores": nt, "what": "closed-form reprojection Jacobians (libvco_fast.so), dual-number IMU blocks, block elimination; threads over frames / IMU blocks",
                "caveat": "not a tuned CPU solver: the oracle's iteration has serial sections (chain solve and weight update on one thread), which is why 4 -> 64 threads buys 1.2x"}
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return {"value": nobs * iters / t, "unit": "corner-residuals/s (LM iterations x corners)", "cores": threads, "kind": "port",
            "sample": "%d LM-iteration work units of the %s (IMU weight update, dual-number Jacobian sweep, IMU blocks, block solve, cost sweep) on the "
                      "first %d of %d frames (%d corners) of %s, %.1f s" % (iters, "final stage" if vi else "vision-only solve", n_sub, len(prob.frame_time), nobs, wl, t),
            "lm_iters_per_sec_extrapolated_to_full": (nobs * iters / t) / prob.n_obs,
            "sweeps_alone": sweeps_alone, "all_cores": all_cores, "closed_form_cpu": best, "host": {"nproc": ncpu, "cpu": cpu_model}}


if __name__ == "__main__":
    main()
