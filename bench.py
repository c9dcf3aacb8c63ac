#!/usr/bin/env python
"""bench.py -- LM-iteration throughput of the calibration hot path on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): stereo fov,fov rig, small
grid (19x10 dots), 500 synthetic frames per GPU, intrinsics + extrinsics, no IMU.  A "step" is one
Levenberg-Marquardt iteration of the real solver (Jacobian sweep over all corners + frame-block Schur
elimination + reduced solve + manifold update + residual sweep of the trial point + accept/reject);
complete solves are run back to back from the same initial state until exactly K iterations are done.
N > 1: one process per GPU, frames sharded (weak scaling: 500 frames per rank), one all-reduce of the
reduced system + one of the step scalars per iteration over RCCL (torch.distributed "nccl").
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# HBM bytes per launch at the default workload, measured with rocprofv3 --pmc (profiles/r01_v10_pmc_traffic_cfg2.txt)
PMC_TRAFFIC = {"k_trial": 6035060.0, "k_reproj_res": 3344207.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=500, help="frames per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-at-scale", action="store_true", help="skip the extra kernel timing on the 10x larger problem")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from vicalib_amd import synth
    from vicalib_amd.lib import ViCalibrator

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # VICALIB_AMD_FORCE_SHARD_PATH=1 (test hook): one rank, but through the sharded code path (split kernels + RCCL callbacks)
    force_shard = world == 1 and os.environ.get("VICALIB_AMD_FORCE_SHARD_PATH") == "1"
    if world > 1 or force_shard:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- this rank's frame shard of the N*frames problem ------------------------------------------
    cfg = synth.Config(models=("fov", "fov"), grid="small", n_frames=args.frames, imu=False, first_frame=rank * args.frames)
    prob = synth.generate(cfg)
    cal = ViCalibrator(local_rank).load_problem(prob)
    cal.SetCalibrateImu(False)

    # Per-iteration all-reduces: the library's own RCCL communicator, enqueued straight onto the calibrator's stream
    # (VICALIB_AMD_SHARD_COMM=torch selects the torch.distributed callback instead; it is also the fallback).
    def attach(c):
        if world == 1 and not force_shard:
            return "none"
        if os.environ.get("VICALIB_AMD_SHARD_COMM", "rccl") == "rccl":
            try:
                c.set_shard_rccl(rank, world)
                return "rccl"
            except Exception as e:      # noqa: BLE001
                print("bench: native RCCL path unavailable (%s); using the torch.distributed callback" % e, file=sys.stderr)
        from vicalib_amd.parallel import FrameShardComm
        c.set_shard(rank, world, FrameShardComm(device="cuda:%d" % local_rank, stream_ptr=c.stream()))
        return "torch"
    comm_kind = attach(cal)

    cal.prepare()
    n_obs_local = cal.num_observations()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    cal.run_iterations(args.warmup)
    barrier()
    t0 = time.perf_counter()
    done, jac_sweeps, res_sweeps = cal.run_iterations(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        no = torch.tensor([float(n_obs_local)], device="cuda", dtype=torch.float64)
        dist.all_reduce(no)
        n_obs_total = int(no.item())
    else:
        n_obs_total = n_obs_local

    # ---- final accuracy of one complete solve (the metric's "final RMS reproj err") --------------------
    cal2 = ViCalibrator(local_rank).load_problem(prob)
    cal2.SetCalibrateImu(False)
    attach(cal2)
    cal2.Solve()
    rmse = [float(x) for x in cal2.GetCameraProjRMSE()]

    # ---- kernel-level roofline of the dominant kernel (HIP events on the calibrator's stream) --------
    # Vision-only passes evaluate the Jacobian sweep at the trial point inside k_trial (one projection sweep per LM
    # iteration): k_trial<true> = back-substitution of the tile's frame + manifold update + Jacobian/Gram sweep of the tile.
    stages = cal.time_stages(50)                 # us per launch, each stage launched 50x back to back
    trial_ms = stages["trial"] * 1e-3
    jac_ms, res_ms = cal.time_kernels(50)        # stand-alone sweeps: k_reproj_jac (first pass of a solve), k_reproj_res (RMSE)
    n_tiles = cal.num_tiles()
    kc = 5   # fov
    # 18 B/corner + per tile [64 B in + 8*(21 + 6 + 6*S_c + 1) B out], S_c = 6 + K_c; the fused kernel also
    # reads the frame factor (48 doubles / frame) and Y (6 x 16 doubles / tile) for the back-substitution
    bytes_jac = 18.0 * n_obs_local + n_tiles * (64 + 8 * (28 + 6 * (6 + kc)))
    bytes_trial = bytes_jac + n_tiles * 8 * 96 + len(prob.frame_time) * 8 * 48
    bytes_res = 18.0 * n_obs_local + n_tiles * (64 + 8)
    flops_jac = 1050.0 * n_obs_local          # ~1.0-1.1 kflop per corner (fp64)
    ach = flops_jac / (trial_ms * 1e-3) / 1e12
    # traffic: HBM bytes per launch from rocprofv3 PMC passes on this exact workload (FETCH_SIZE and WRITE_SIZE in
    # separate runs, KB -> bytes, FETCH x2 per the gfx950 note in MI355X_MICROARCH.md): profiles/r01_v10_pmc_traffic_cfg2.txt
    base_cfg = (args.frames == 500 and world == 1)
    traffic_trial = PMC_TRAFFIC.get("k_trial") if base_cfg else None
    traffic_res = PMC_TRAFFIC.get("k_reproj_res") if base_cfg else None
    roofline = {"kernel": "k_trial<fused Jacobian sweep>", "bound": "mfma", "achieved": ach, "peak": 78.6, "unit": "TFLOP/s", "frac": ach / 78.6,
                "traffic": traffic_trial, "hbm_gbs": bytes_trial / (trial_ms * 1e-3) / 1e9, "hbm_frac": bytes_trial / (trial_ms * 1e-3) / 8e12,
                "avg_ms": trial_ms, "algorithmic_bytes": bytes_trial, "algorithmic_flops": flops_jac,
                "standalone_jacobian_sweep_ms": jac_ms, "stage_us": stages}
    roofline_res = {"kernel": "k_reproj_res", "bound": "hbm", "achieved": bytes_res / (res_ms * 1e-3) / 1e9, "peak": 8000.0,
                    "unit": "GB/s", "frac": bytes_res / (res_ms * 1e-3) / 8e12, "traffic": traffic_res, "avg_ms": res_ms,
                    "algorithmic_bytes": bytes_res}

    out = None
    if rank == 0:
        iters_per_s = done / dt
        value = n_obs_total * iters_per_s
        out = {
            "metric": "corner_residuals_per_sec", "value": value, "unit": "corner-residuals/s (LM iterations x corners)",
            "n_gpus": world, "steps": done, "warmup": args.warmup, "ms_per_step": 1e3 * dt / done, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE cfg2: stereo fov,fov, small grid 19x10, %d frames/GPU, intrinsics+extrinsics, no IMU" % args.frames,
                       "frames_total": args.frames * world, "corners_total": n_obs_total, "tiles_per_gpu": n_tiles,
                       "parallelism": "frames sharded x%d, all-reduce of reduced system per LM iteration (%s)" % (world, comm_kind)},
            "lm_iters_per_sec": iters_per_s, "jacobian_sweeps": jac_sweeps, "residual_sweeps": res_sweeps,
            "final_rmse_px": rmse, "roofline": roofline, "roofline_residual_sweep": roofline_res,
        }
        if not args.no_at_scale and world == 1 and not force_shard:
            out["roofline_at_scale"] = roofline_at_scale(local_rank, args.frames * 10)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(prob)
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
        # group, the library's own communicators) is torn down first
        try:
            dist.barrier()
        except Exception:      # noqa: BLE001
            pass
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


def roofline_at_scale(device, n_frames):
    """The same kernels on the same rig with 10x the frames (cfg2 is one wave per SIMD: latency-bound by construction);
    informational, the bench value above is the cfg2 number."""
    from vicalib_amd import synth
    from vicalib_amd.lib import ViCalibrator
    p = synth.generate(synth.Config(models=("fov", "fov"), grid="small", n_frames=n_frames, imu=False))
    cal = ViCalibrator(device).load_problem(p); cal.SetCalibrateImu(False); cal.prepare()
    st = cal.time_stages(20)
    n = cal.num_observations()
    tf = 1050.0 * n / (st["trial"] * 1e-6) / 1e12
    return {"workload": "stereo fov,fov, small grid, %d frames" % n_frames, "corners": n, "kernel": "k_trial<fused Jacobian sweep>",
            "avg_ms": st["trial"] * 1e-3, "achieved": tf, "peak": 78.6, "unit": "TFLOP/s", "frac": tf / 78.6,
            "corners_per_sec": n / (st["trial"] * 1e-6), "stage_us": st,
            "lm_iteration_us_sum_of_stages": st["frame_schur"] + st["reduced"] + st["trial"] + st["final"]}


def cpu_baseline(prob):
    """The CPU oracle (a port of the reference's autodiff + solver path, since Ceres cannot be built here)
    timed on this host on a bounded sample of the same workload: full LM-iteration work units on the
    first frames of the same problem, 4 threads (the reference's num_threads, vicalibrator.h:141)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    threads = 4
    n_sub = min(100, len(prob.frame_time))
    orc = ol.Oracle()
    for c, m in enumerate(prob.cam_model):
        orc.add_camera(m, prob.cam_K_init[c], prob.cam_T_ck_init[c])
    for n in range(n_sub):
        orc.add_frame(prob.frame_T_wk_init[n], prob.frame_time[n])
    nobs = 0
    for (f, c, ids, pix) in prob.tiles:
        if f < n_sub:
            orc.add_observations(f, c, prob.grid_points[ids], pix); nobs += len(ids)
    orc.set_options(calibrate_imu=False, num_threads=threads)
    orc.prepare(vis_mult=1)
    orc.time_iterations(1)
    iters = 0; t = 0.0
    while t < 10.0 and iters < 400:
        t += orc.time_iterations(4); iters += 4
    # the same work on every host core (SURVEY 8d asks for both; `value` stays the reference's 4-thread configuration)
    ncpu = os.cpu_count() or threads
    all_cores = None
    if ncpu > threads:
        orc.set_options(calibrate_imu=False, num_threads=ncpu)
        orc.time_iterations(1)
        it2 = 0; t2 = 0.0
        while t2 < 5.0 and it2 < 400:
            t2 += orc.time_iterations(4); it2 += 4
        all_cores = {"value": nobs * it2 / t2, "cores": ncpu}
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return {"value": nobs * iters / t, "unit": "corner-residuals/s (LM iterations x corners)", "cores": threads, "kind": "port",
            "sample": "%d LM-iteration work units (dual-number Jacobian sweep + block solve + cost sweep) on the first %d of %d frames (%d corners), %.1f s"
                      % (iters, n_sub, len(prob.frame_time), nobs, t),
            "lm_iters_per_sec_extrapolated_to_full": (nobs * iters / t) / prob.n_obs,
            "all_cores": all_cores, "host": {"nproc": ncpu, "cpu": cpu_model}}


if __name__ == "__main__":
    main()
