#pragma once
// vc_calibrator.hpp -- the host driver behind the C ABI (include/vicalib_amd.h): problem container and state of one calibrator.
//
// Mirrors visual_inertial_calibration::ViCalibrator (include/vicalib/vicalibrator.h).  The long member functions live in their own
// translation units: vc_upload.cpp (SetupProblem: layout + device upload), vc_pass.cpp (one LM pass as a launch graph over two
// streams), vc_solve.cpp (ceres::Solve replacement: feeding passes, the stage machine, RMSE, outliers, gravity); the C entry points
// are in vc_capi.cpp.
#include "vc_host.hpp"

struct vc_calibrator {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;        // the IMU weight update of a pass runs here, under the pass's Jacobian sweeps and chain solve
  hipEvent_t ev_state = nullptr, ev_weights = nullptr, ev_imujac = nullptr, ev_reduced = nullptr, ev_back = nullptr, ev_pre = nullptr;
  bool top_gram_launch = false;         // early Gram: the chain's top-level frames need a Gram launch of their own (upload)
  bool pre_weights_pending = false;     // solve_once has recorded ev_pre ahead of the weight update that precedes a solve
  bool pre_weights_fresh = false;       // ... and nothing has moved the state since: the first pass's own update would repeat it
  int wcur = 0;                         // weight buffer holding the current weight_sqrt_
  Packer pack;                          // staging image of a stage's small uploads
  bool flag_sync = false;               // hand-overs to the second stream through device flags instead of event records (set at creation)
  bool weights_behind_l0 = !(std::getenv("VICALIB_AMD_WEIGHTS_BEHIND_L0") && std::getenv("VICALIB_AMD_WEIGHTS_BEHIND_L0")[0] == '0');
  bool shard_flag_sync = std::getenv("VICALIB_AMD_SHARD_FLAG_SYNC") && std::getenv("VICALIB_AMD_SHARD_FLAG_SYNC")[0] == '1';
  long long sync_bound = 800000;        // polls before a flag wait gives up (~0.2 s); VICALIB_AMD_SYNC_BOUND (test hook: a tiny bound forces the time-out path)
  int sync_bound_from_pass = 0;         // VICALIB_AMD_SYNC_BOUND_FROM_PASS (test hook): the tiny bound only from this pass of the calibrator on -- a time-out in the middle of a solve
  int wr_ring[16] = {0};                // weight buffer read by pass (pass_seq & 15)
  int sync_timeouts = 0;                // flag hand-overs that ran into their bound (each one reported on stderr, the solve resumed with events)
  long long pass_seq = 0;               // passes enqueued (the value the flags carry)
  bool prev_pass_signals = false;       // the previous pass of this solve was enqueued with signalling kernels
  DBuf<long long> d_sync, d_part_ready;
  bool jac_on_stream2 = true;           // the trial point's k_imu_jac beside the vision sweep (VICALIB_AMD_JAC_STREAM2=0: after it, main stream)
  bool serial_weights = false;          // false: IMU Jacobians + weight update on the second stream (VICALIB_AMD_OVERLAP_WEIGHTS=0: in line); was: VICALIB_AMD_OVERLAP_WEIGHTS=1 moves it to a second
                                        // stream under the Jacobian sweeps / chain solve (measured: the two latency-bound kernels then
                                        // share the CUs and the pass gets 4 % slower on cfg3)
  hipGraphExec_t pass_graph[2] = {nullptr, nullptr};   // one captured LM pass per weight-buffer parity (single process)
  bool use_graphs = false;      // measured slower on ROCm 7.2 (cfg2: 65 vs 62 us / pass, instantiation ~10 ms per stage): opt-in via VICALIB_AMD_GRAPHS=1
  hipError_t last_hip_error = hipSuccess;
  // ---- problem (host copy) ---------------------------------------------------------------
  std::vector<HostCam> cams;
  std::vector<HostFrame> frames;
  std::vector<int> o_frame, o_cam, o_pid;     // o_pid: index into the table of distinct target points
  std::vector<double> o_pc;
  PointTable pts;                             // exact-bit de-duplication of the p_w the caller passes, done once at AddObservation
  std::vector<signed char> o_removed;       // RemoveOutliers: 1 = no copy left (dropped), 2 = one copy fewer than vis_mult (kObsOneLess)
  long n_one_less = 0;
  bool obs_dirty = true;          // the observation set (or its multiplicity bits) changed since the tile layout was built
  int n_points_dev = 0;
  std::vector<double> imu_w, imu_a, imu_t;
  double imu_end_time = -1.0;
  double g_dir[2] = {0, 0}, time_offset = 0, biases[6] = {0, 0, 0, 0, 0, 0}, scale[6] = {1, 1, 1, 1, 1, 1};
  double gyro_sigma = 5.3088444e-5, accel_sigma = 0.001883649;   // types.h:34-35
  // ---- flags: Clear() defaults, vicalibrator.h:232-249 -----------------------------------
  bool fix_intrinsics = false, is_bias_active = false, is_scale_active = false, is_inertial_active = false,
       is_visual_active = true, rotation_only = true, optimize_time_offset = true, is_finished = false,
       gravity_initialized = false, outliers_removed = false;
  int pnp_its = 0; double pnp_tol = 0.0;     // PosePnPRansac(..., robust_3pt_its = 0, robust_3pt_tol = 0, ...) at vicalib-task.cc:323-325
  int max_iters = 200;                       // FLAGS_max_iters
  double function_tolerance = 1e-6;          // vicalibrator.h:149
  double gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;   // Ceres defaults
  bool calibrate_imu = true, remove_outliers = false;
  double outlier_threshold = 2.0;
  int vis_mult = 0, imu_mult = 0;
  // ---- results ----------------------------------------------------------------------------
  std::vector<double> cam_rmse;
  double mse = 0;
  std::atomic<unsigned> num_iterations{0};
  std::vector<IterRecord> trace;
  int stage = 0;
  long jac_sweeps = 0, res_sweeps = 0;
  // ---- threading --------------------------------------------------------------------------
  std::thread worker;
  std::atomic<bool> is_running{false}, should_run{false};
  std::mutex result_mutex;
  // ---- sharding ---------------------------------------------------------------------------
  int rank = 0, world = 1;
  bool force_shard_path = false;   // VICALIB_AMD_FORCE_SHARD_PATH=1: run the sharded code path (split kernels + callbacks) with one rank (test hook)
  bool sharded() const { return world > 1 || (force_shard_path && (allreduce || rccl_comm)); }
  DBuf<double> d_halo, d_sep_strip, d_gath;
  long global_first = 0, global_total = 0;     // this rank's frame range in the sharded problem (known after gather_shard_info)
  vc_allreduce_fn allreduce = nullptr;
  void* allreduce_ctx = nullptr;
  void* rccl_comm = nullptr;        // RCCL communicator (vc_set_shard_rccl: own; vc_set_shard_comm: borrowed): all-reduces go straight onto `stream`
  bool rccl_comm_owned = false;     // this calibrator created it and destroys it
  void drop_comm() {
    if (rccl_comm && rccl_comm_owned && g_rccl.CommDestroy) { (void)hipStreamSynchronize(stream); (void)g_rccl.CommDestroy(rccl_comm); }
    rccl_comm = nullptr; rccl_comm_owned = false;
  }
  long rccl_calls = 0;
  // ---- device -----------------------------------------------------------------------------
  bool device_dirty = true;      // host problem changed since the last upload
  DevView dv{};
  int cur = 0;
  DBuf<double2> d_uv; DBuf<unsigned short> d_pt; DBuf<double> d_points;
  DBuf<TileHdr> d_tile_hdr;
  DBuf<int> d_tile_frame, d_tile_cam, d_tile_off, d_frame_tile_off, d_frame_cam_tile, d_cam_model, d_cam_flags, d_cam_col0,
      d_col_cam, d_col_local, d_flags;
  DBuf<double> d_wgpart;
  int kpass = 0;                  // passes enqueued since init_ctrl (merged mode: selects the control record and flag parity)
  DBuf<double> d_pose[2], d_cam[2], d_G[2], d_tile_cost[2], d_Y, d_fr, d_fdiag, d_fscale2, d_part, d_Sbuf, d_hadd, d_sdiag,
      d_sscale2, d_slam, d_delta_s, d_fpart, d_scal, d_tmp, d_pose_init, d_cam_init, d_tile_trial, d_trace, d_part_total;
  DBuf<Ctrl> d_ctrl;
  DBuf<double> d_vel[2], d_imus[2], d_imu_t, d_imu_w, d_imu_a, d_frame_time, d_wsqrt[2], d_seg[2], d_seg_cost[2],
      d_cW, d_cdelta, d_ct0, d_cg, d_clam, d_cdiag, d_cscale2, d_vel_init, d_imus_init, d_rX[2], d_grp_part, d_wg_trial, d_wg_imu_trial,
      d_imu_delta_blk, d_imu_grav;
  DBuf<long long> d_cready;
  size_t wsqrt_frames = 0;       // number of frames the device weight_sqrt_ array was initialised for
  size_t imu_uploaded = 0; double imu_uploaded_last = 0.0;      // sample count / last time stamp of the device copy of the IMU samples
  int trace_cap = 0;
  struct Pinned { Ctrl up; Ctrl down; Ctrl dev; double trace[64 * kTraceCols]; unsigned long long progress; };      // dev / trace / progress: written by the device
  Pinned* pin = nullptr;        // page-locked staging (async copies without a bounce buffer)
  long nres_global_cached = -1; int nres_mult_cached[2] = {-1, -1};      // sharded: the all-reduced residual count and the multiplicities it was formed with
  long solve_epoch = 0, nres_epoch_cached = -1;     // ... and the public solve call it was formed in (bumped by every rank at the same entry points)
  int expected_passes = 8;       // passes the previous solve needed: size of the first batch of the next one (batched schedule)
  int feed_ahead = 1;            // passes kept queued beyond the last decision seen (grows when the host is found late)
  bool feed_passes = std::getenv("VICALIB_AMD_BATCHED") == nullptr;   // single process: feed passes against the device's progress word
  DBuf<unsigned char> d_mask;
  std::vector<int> h_tile_frame, h_tile_cam, h_tile_off, h_obs_index;   // h_obs_index: device corner -> host observation
  std::vector<int> cam_flags, cam_col0;

  ~vc_calibrator() {
    stop();
    drop_comm();
    drop_graphs();
    kt_free();
    if (stream2) (void)hipStreamDestroy(stream2);
    if (ev_state) (void)hipEventDestroy(ev_state);
    if (ev_pre) (void)hipEventDestroy(ev_pre);
    if (ev_weights) (void)hipEventDestroy(ev_weights);
    if (ev_imujac) (void)hipEventDestroy(ev_imujac);
    if (ev_reduced) (void)hipEventDestroy(ev_reduced);
    if (ev_back) (void)hipEventDestroy(ev_back);
    if (stream) (void)hipStreamDestroy(stream);
    if (pin) (void)hipHostFree(pin);
  }
  void stop() {
    should_run = false;
    if (worker.joinable()) worker.join();
  }

  // ---- layout of the shared (non-frame) parameters: SetupProblem constancy rules -------------
  int build_layout(std::vector<int>& col_cam, std::vector<int>& col_local);
  bool imu_on() const { return calibrate_imu && is_inertial_active; }
  int imu_param_col[15];

  // sum a small host vector over the ranks through the caller's device all-reduce
  int host_allreduce_sum(std::vector<double>& v) {
    if (!sharded()) return VC_OK;
    HIP_OK(d_halo.upload(v, stream));
    int rc = do_allreduce(d_halo.p, (int)v.size(), 0); if (rc) return rc;
    HIP_OK(hipMemcpyAsync(v.data(), d_halo.p, v.size() * 8, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    return VC_OK;
  }
  // every rank's first frame (pose, velocity, time) and frame count: slot r of the gathered table
  int gather_shard_info(std::vector<double>* table) {
    table->assign((size_t)world * 12, 0.0);
    if (!frames.empty()) {
      double* o = table->data() + (size_t)rank * 12;
      std::memcpy(o, frames[0].T, 56); std::memcpy(o + 7, frames[0].v, 24); o[10] = frames[0].time;
    }
    (*table)[(size_t)rank * 12 + 11] = (double)frames.size();
    int rc = host_allreduce_sum(*table); if (rc) return rc;
    global_first = 0; global_total = 0;
    for (int r = 0; r < world; ++r) { if (r < rank) global_first += (long)(*table)[(size_t)r * 12 + 11]; global_total += (long)(*table)[(size_t)r * 12 + 11]; }
    return VC_OK;
  }

  int upload();
  // device-to-device reset of the state to what upload() put there (benchmark restarts)
  int reset_state();
  // copy the accepted device state back into the host problem
  int download_state();

  void drop_graphs() {
    for (int i = 0; i < 2; ++i) if (pass_graph[i]) { (void)hipGraphExecDestroy(pass_graph[i]); pass_graph[i] = nullptr; }
  }
  // A pass is a fixed sequence of launches (up to ~45 with the IMU chain): captured once per upload and replayed, the host
  // pays one graph launch per pass instead of one call per kernel.  Sharded runs keep direct launches (host callbacks).
  int launch_pass_graph();

  // ---- in-loop kernel timing (vc_set_kernel_timing): every launch group of a pass bracketed by a pair of events on the
  // calibrator's stream; durations are read back after the solve.  Off by default (an event record costs ~1 us of stream time).
  bool ktime_on = false;
  std::vector<hipEvent_t> kt_ev;            // pool: 2 per bracket
  std::vector<int> kt_label;                // label of bracket i
  size_t kt_used = 0;
  std::vector<std::string> kt_names;
  std::vector<double> kt_total_ms; std::vector<long> kt_count;
  int kt_label_id(const char* name) {
    for (size_t i = 0; i < kt_names.size(); ++i) if (kt_names[i] == name) return (int)i;
    kt_names.push_back(name); kt_total_ms.push_back(0.0); kt_count.push_back(0);
    return (int)kt_names.size() - 1;
  }
  hipStream_t kt_stream = nullptr;
  void kt_begin(const char* name, hipStream_t strm = nullptr) {
    kt_stream = strm ? strm : stream;
    if (kt_ev.size() < 2 * (kt_used + 1)) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { ktime_on = false; return; }
      kt_ev.push_back(a); kt_ev.push_back(b);
    }
    kt_label.resize(kt_used + 1); kt_label[kt_used] = kt_label_id(name);
    (void)hipEventRecord(kt_ev[2 * kt_used], kt_stream);
  }
  void kt_end() { (void)hipEventRecord(kt_ev[2 * kt_used + 1], kt_stream); ++kt_used; }
  void kt_collect() {          // after a stream synchronisation
    for (size_t i = 0; i < kt_used; ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, kt_ev[2 * i], kt_ev[2 * i + 1]) == hipSuccess) { kt_total_ms[kt_label[i]] += ms; kt_count[kt_label[i]] += 1; }
    }
    kt_used = 0;
  }
  void kt_free() { for (hipEvent_t e : kt_ev) (void)hipEventDestroy(e); kt_ev.clear(); kt_used = 0; }
#define KT(name, call) do { if (ktime_on) kt_begin(name); call; if (ktime_on) kt_end(); } while (0)
#define KT2(name, call) do { if (ktime_on) kt_begin(name, stream2); call; if (ktime_on) kt_end(); } while (0)

  // ---- one pass of the device pipeline (all asynchronous; the decision is taken on the device) ------
  int do_allreduce(double* p, int n, int op) {
    if (sharded() && rccl_comm) {
      ++rccl_calls;
      if (g_rccl.AllReduce(p, p, (size_t)n, kNcclDouble, op == 1 ? kNcclMax : kNcclSum, rccl_comm, stream) != 0) return VC_ERR_NO_DEVICE;
    } else if (sharded() && allreduce) { if (allreduce(allreduce_ctx, p, n, op) != 0) return VC_ERR_NO_DEVICE; }
    return VC_OK;
  }
  // first_pass: the pass right after init_ctrl (the only one that needs k_reproj_jac when k_trial carries the sweep)
  // events_only: a stand-alone pass outside a solve (parity hooks, timing): nobody would resume it after a flag time-out
  int enqueue_pass(bool first_pass = true, bool events_only = false);
  // merged mode: judge the last enqueued pass; afterwards `ctrl_result()` is the record to read back
  void finish_batch() {
    if (!dv.merged) return;
    dv.par = kpass & 1; dv.ctrl = d_ctrl.p + (kpass & 1); dv.ctrl_prev = d_ctrl.p + ((kpass + 1) & 1);
    launch_final_merged(dv, stream);
  }
  const Ctrl* ctrl_result() const { return dv.merged ? d_ctrl.p + (kpass & 1) : d_ctrl.p; }
  bool merged_enabled = true;
  // a fresh control record goes to buffer 0; buffer 1 (the "previous pass" of the first pass in merged mode) is blanked
  int upload_ctrl(const Ctrl* c) {
    launch_set_ctrl(d_ctrl.p, *c, stream);
    kpass = 0;
    return VC_OK;
  }
  void init_ctrl(Ctrl* c) {
    std::memset(c, 0, sizeof(Ctrl));
    c->radius = 1e4; c->decrease_factor = 2.0;
    c->ftol = function_tolerance; c->gtol = gradient_tolerance; c->ptol = parameter_tolerance; c->mult = (double)vis_mult;
    c->imu_mult = (double)imu_mult;
    c->cur = cur; c->reuse_diag = 0; c->need_lin = 1; c->init_scale = 1; c->max_iters = max_iters;
    c->first = 1; c->trace_cap = trace_cap; c->stage = stage;
  }

  // A device-flag hand-over ran into its bound (vc_kutil.hpp: spin_until_flag): the device has withheld the decision of pass
  // `abort_seq` and of everything queued behind it, the accepted state and the control record are those of the last valid decision.
  // Say so, switch this calibrator to event hand-overs for good, and put the solve back on its feet: both streams idle, flags
  // cleared, the weight buffer that pass was reading current again, `done` cleared and a fresh linearisation at the accepted state
  // requested (what the trial sweeps of the last judged pass left behind may be incomplete).  The passes that follow repeat the
  // withheld ones with events: the same iterates as a run that never used the flags.
  int resume_after_sync_timeout(const Ctrl& c);
  // The trust-region loop (ceres::Solve :956 with LEVENBERG_MARQUARDT).  The loop itself runs
  // on the device (lm_decide in vc_kernels.hip); the host enqueues passes in batches and polls Ctrl::done.
  int solve_once(Termination* term, double* final_cost, long* nres);
  int last_iters = 0;
  // one pass with the decision logic on hold (parity hooks): linearise at the accepted state
  int linearize_hold(double radius, double* cost);

  // per-camera RMSE, vicalibrator.h:958-971 (unrobustified, latest copy)
  int compute_rmse();
  // RemoveOutliers, vicalibrator.h:859-916
  int remove_outliers_pass();

  // Gravity initialisation, vicalibrator.h:927-949: accel at the middle frame's time (offset 0), rotated into the world
  int init_gravity();

  // SolveThread, vicalibrator.h:919-1040
  static constexpr int kMaxRepeats = 64;
  int stage_limit = -1;
  int solve();
};
