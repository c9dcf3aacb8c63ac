// vc_solve.cpp -- what replaces ceres::Solve and ViCalibrator::SolveThread (vicalibrator.h:919-1040): feeding passes against the device's
// progress word, the resume after a flag time-out, the stage machine, per-camera RMSE (:958-971), RemoveOutliers (:859-916), gravity (:927-949).
#include "vc_calibrator.hpp"

int vc_calibrator::reset_state() {
  cur = 0;
  launch_reset_state(dv, d_pose_init.p, d_cam_init.p, d_vel_init.p, d_imus_init.p, stream);
  return VC_OK;
}

int vc_calibrator::download_state() {
  const int N = (int)frames.size(), C = (int)cams.size();
  std::vector<double> poses((size_t)N * kPoseStride), camrec((size_t)C * kCamStride);
  if (N) HIP_OK(hipMemcpyAsync(poses.data(), dv.poses[cur], poses.size() * 8, hipMemcpyDeviceToHost, stream));
  if (C) HIP_OK(hipMemcpyAsync(camrec.data(), dv.cams[cur], camrec.size() * 8, hipMemcpyDeviceToHost, stream));
  std::vector<double> vels((size_t)std::max(N, 1) * 4, 0.0), imus(16, 0.0);
  if (N) HIP_OK(hipMemcpyAsync(vels.data(), dv.vel[cur], (size_t)N * 4 * 8, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(imus.data(), dv.imus[cur], 16 * 8, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  std::lock_guard<std::mutex> lk(result_mutex);
  for (int f = 0; f < N; ++f) { std::memcpy(frames[f].T, &poses[(size_t)f * kPoseStride], 56); std::memcpy(frames[f].v, &vels[(size_t)f * 4], 24); }
  g_dir[0] = imus[0]; g_dir[1] = imus[1];
  for (int i = 0; i < 6; ++i) { biases[i] = imus[2 + i]; scale[i] = imus[8 + i]; }
  time_offset = imus[14];
  for (int c = 0; c < C; ++c) {
    std::memcpy(cams[c].T_ck, &camrec[(size_t)c * kCamStride], 56);
    std::memcpy(cams[c].K, &camrec[(size_t)c * kCamStride + kCamK], cams[c].nk * 8);
  }
  return VC_OK;
}

int vc_calibrator::resume_after_sync_timeout(const Ctrl& c) {
  HIP_OK(hipStreamSynchronize(stream));
  if (stream2) HIP_OK(hipStreamSynchronize(stream2));
  ++sync_timeouts;
  flag_sync = false;
  std::fprintf(stderr, "vicalib_amd: a device-flag hand-over between the two streams ran into its bound in LM pass %d (the streams share a hardware "
                       "queue, a tool serialises the queues, or several processes share the device); no step was taken on its data -- resuming the "
                       "solve with event hand-overs, which this calibrator keeps from now on (VICALIB_AMD_FLAG_SYNC=0 selects them from the start)\n",
               c.passes + 1);
  wcur = wr_ring[c.abort_seq & 15];
  if (d_sync.p) HIP_OK(hipMemsetAsync(d_sync.p, 0, kSyncWords * sizeof(long long), stream));
  HIP_OK(hipMemsetAsync(dv.flags + 4, 0, 4 * sizeof(int), stream));      // numeric-failure marks of the void passes
  Ctrl r = c;
  // need_lin stays as the last decision left it: after a rejected step the linearisation in place is the one made when the state was
  // accepted, with the IMU weights of THAT pass -- the weights have moved on since (they are updated every pass), so linearising
  // again here would not reproduce it (costs of the following rejected steps off by 1e-7 relative: the time-out test in the middle of
  // a rejected streak); the void passes only wrote the trial-side buffers, the accepted state's records are intact
  r.done = 0; r.abort_seq = 0;
  HIP_OK(hipMemcpyAsync(d_ctrl.p, &r, sizeof(Ctrl), hipMemcpyHostToDevice, stream));
  HIP_OK(hipStreamSynchronize(stream));
  pin->down.done = 0;
  prev_pass_signals = false;
  return VC_OK;
}

int vc_calibrator::solve_once(Termination* term, double* final_cost, long* nres) {
  RoctxRange rr("vicalib_amd: solve (ceres::Solve of one stage)");
  if (device_dirty) { int rc = upload(); if (rc) return rc; }
  *nres = 2L * ((long)dv.n_obs * vis_mult - n_one_less) + (dv.imu_on ? 9L * imu_mult * std::max(0, dv.n_frames - 1) : 0L);
  if (sharded()) {
    // The global residual count changes with the observation set and with the multiplicities.  The test for a fresh collective
    // must not depend on anything rank-local (a rank that skipped it while another entered would hang the job) -- not on
    // device_dirty, which a mutator called on one rank only would set there alone: the key is the public solve call (every rank
    // enters Solve() / vc_run_iterations together, they contain collectives anyway) and the multiplicities, which all ranks
    // bump together, outlier removal included.  One collective per stage of a solve.
    if (nres_epoch_cached != solve_epoch || nres_mult_cached[0] != vis_mult || nres_mult_cached[1] != imu_mult) {
      std::vector<double> v = {(double)*nres};
      int rc = host_allreduce_sum(v); if (rc) return rc;
      nres_global_cached = (long)v[0]; nres_mult_cached[0] = vis_mult; nres_mult_cached[1] = imu_mult; nres_epoch_cached = solve_epoch;
    }
    *nres = nres_global_cached;
  }
  if (trace_cap < max_iters + 8) { trace_cap = max_iters + 8; HIP_OK(d_trace.alloc((size_t)trace_cap * kTraceCols)); dv.trace = d_trace.p; }
  // the progress word is written by the device and polled by the host: coherent (fine-grained), mapped memory whatever
  // HIP_HOST_COHERENT says -- with a non-coherent allocation the host would never see the device's stores
  if (!pin) HIP_OK(hipHostMalloc((void**)&pin, sizeof(Pinned), hipHostMallocCoherent | hipHostMallocMapped));
  init_ctrl(&pin->up);
  { int rcu = upload_ctrl(&pin->up); if (rcu) return rcu; }
  // a wait that ran into its bound in a pass queued past the end of the previous solve (nobody judged it, nobody reported it) must
  // not void this solve's first pass: the sticky word starts every solve clear
  if (d_sync.p) HIP_OK(hipMemsetAsync(d_sync.p + 6, 0, sizeof(long long), stream));
  pre_weights_fresh = false; pre_weights_pending = false;
  if (dv.imu_on && dv.weights_on) {     // UpdateImuWeights() before ceres::Solve (vicalibrator.h:955)
    if (!serial_weights && stream2) { HIP_OK(hipEventRecord(ev_pre, stream)); pre_weights_pending = true; }
    dv.sync_seq = 0;      // (not a pass: a sticky time-out mark left by the previous solve's last pass must not make this update skip itself)
    launch_imu_weights(dv, wcur, stream); wcur = 1 - wcur;
    pre_weights_fresh = true;
  }
  const size_t trace_bytes = (size_t)std::min(trace_cap, 64) * kTraceCols * 8;
  int guard = 0, n_enq = 0;
  bool first_enq = true;              // the next pass enqueued linearises at the accepted state (start of the solve, resume after a flag time-out)
  double enqueue_ms = 0.0, wait_ms = 0.0;
  const auto tso0 = std::chrono::steady_clock::now();
  for (;;) {                          // (one round, unless a device-flag hand-over runs into its bound: then a second one, with events)
  const bool feed = !sharded() && !use_graphs && feed_passes && dv.imu_on;
  if (feed) {
    // Single process, visual-inertial passes (18 launches at cfg3, ~270 us): the deciding thread publishes (decisions << 32 | done) to a page-locked word after every decision and
    // the host keeps kAhead passes queued beyond the last decision it has seen -- no stream synchronisation inside the
    // solve (each one drains the queue: ~40 us of idle device), at most kAhead passes enqueued past the end (they return at
    // their first instruction).  Enqueueing a pass takes the host a fraction of the pass's run time.
    // kAhead starts at 1 and grows (up to 4, kept for the calibrator's lifetime) whenever the host finds every enqueued pass
    // already decided -- it came back late (a busy host: one box of the pool ran cfg3 at 0.40 instead of 0.30 ms per pass, with
    // the launch-ahead schedules unaffected) and the device has been idle; a longer queue rides such gaps out.
    int& kAhead = feed_ahead;
    volatile unsigned long long* prog = &pin->progress;
    *prog = (unsigned long long)(unsigned)n_enq << 32;      // (0 at the start of a solve; the decisions taken so far when a solve is resumed)
    dv.host_progress = &pin->progress; dv.host_ctrl = &pin->dev; dv.host_trace = pin->trace;
    bool done_seen = false;
    auto t_seen = std::chrono::steady_clock::now();
    unsigned long long last = 0ull;
    while (should_run) {
      const unsigned long long f = *prog;
      if ((unsigned)(f & 0xffffffffull & ~(unsigned long long)kProgressLikelyLast) != 0u) { done_seen = true; break; }     // Ctrl::done
      if (f != last) { last = f; t_seen = std::chrono::steady_clock::now(); }
      const int decided = (int)(f >> 32);
      if (n_enq >= max_iters + 8) break;
      if (!first_enq && decided >= n_enq && kAhead < 4 && !(f & kProgressLikelyLast)) ++kAhead;
      // the device expects the pass after the last decision to end the solve (lm_decide_local: likely_last): nothing is queued
      // past it -- a pass queued past the end costs ~90 us of empty launches at cfg3 before the stream is free again, a wrong guess
      // one host round trip
      const int ahead = (f & kProgressLikelyLast) ? 0 : kAhead;
      if (n_enq - decided <= ahead) {
        int rc = enqueue_pass(first_enq); if (rc) { dv.host_progress = nullptr; return rc; }
        first_enq = false;
        ++n_enq; t_seen = std::chrono::steady_clock::now();
      } else {
        __builtin_ia32_pause();
        // the queue is full: nothing to do until the device decides a pass (~0.3 ms); past ~50 us without news, yield the core
        const auto idle = std::chrono::steady_clock::now() - t_seen;
        if (idle > std::chrono::microseconds(50)) std::this_thread::yield();
        if (idle > std::chrono::seconds(5)) {          // a stuck device (or a progress word the host cannot see): say so, then
          std::fprintf(stderr, "vicalib_amd: no progress from the device for 5 s (%d passes queued, %d decided) -- falling back to a "
                               "synchronising read\n", n_enq, decided);                     // fall through to the synchronising read
          break;
        }
      }
    }
    dv.host_progress = nullptr; dv.host_ctrl = nullptr; dv.host_trace = nullptr;
    finish_batch();
    std::atomic_thread_fence(std::memory_order_acquire);
    if (done_seen && !ktime_on && pin->dev.trace_len <= 64 && pin->dev.done != kDoneSyncTimeout) {
      // the deciding thread has left the record and the trace rows in page-locked memory before it said `done`: nothing to
      // copy and nothing to wait for -- the passes queued past the end (they return at their first instruction) drain while the
      // host goes on; whatever the caller enqueues next is ordered behind them by the streams
      pin->down = pin->dev;
    } else {
      HIP_OK(hipMemcpyAsync(&pin->down, ctrl_result(), sizeof(Ctrl), hipMemcpyDeviceToHost, stream));
      HIP_OK(hipMemcpyAsync(pin->trace, d_trace.p, trace_bytes, hipMemcpyDeviceToHost, stream));
      HIP_OK(hipStreamSynchronize(stream));
      if (ktime_on) kt_collect();
    }
  }
  // Sharded (every rank must run the same schedule: the passes contain collectives), graph replay and the vision-only path
  // (four launches of ~13 us per pass, a handful of passes per solve: measured 60 vs 63 us per iteration at cfg2 -- the
  // passes fed past the end cost more there than one synchronisation): batches.  First batch =
  // what the previous solve needed (repeated solves of similar problems: no wasted launches, one host sync per solve); then
  // small top-up batches until the device reports `done`.
  // (visual-inertial: a first batch of at most 16 passes: a long previous solve -- stage C's 38 iterations ahead of stage D's 12 -- must not
  //  queue dozens of passes past the end; then top-ups of 8: one synchronisation per ~2 ms of device work)
  int batch = std::max(1, std::min(dv.imu_on ? std::min(expected_passes, 16) : expected_passes, max_iters + 1));
  while (!feed || (!pin->down.done && should_run && n_enq < max_iters + 8)) {
    const auto tq0 = std::chrono::steady_clock::now();
    for (int b = 0; b < batch; ++b) {
      const bool first = first_enq; first_enq = false; ++n_enq;
      int rc = (first || sharded() || !use_graphs) ? enqueue_pass(first) : launch_pass_graph();
      if (rc) return rc;
    }
    finish_batch();
    enqueue_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count();
    HIP_OK(hipMemcpyAsync(&pin->down, ctrl_result(), sizeof(Ctrl), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(pin->trace, d_trace.p, trace_bytes, hipMemcpyDeviceToHost, stream));
    const auto tw0 = std::chrono::steady_clock::now();
    HIP_OK(hipStreamSynchronize(stream));
    wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
    if (ktime_on) kt_collect();
    // Stop() is a collective decision when the frames are sharded: a rank that left its enqueue loop alone would leave
    // its peers waiting in the next all-reduce (every rank runs the same batch schedule, so the counts line up)
    // (every rank holds the same control record: a finished solve ends here on all of them without another collective)
    if (pin->down.done) break;
    if (sharded()) {
      std::vector<double> v = {should_run ? 0.0 : 1.0};
      int rc = host_allreduce_sum(v); if (rc) return rc;
      if (v[0] > 0.0) should_run = false;
    }
    if (!should_run || ++guard > max_iters + 8) break;
    batch = dv.imu_on ? 8 : 2;        // (vision-only passes are 50 us: a synchronisation every two of them was the better trade there)
  }
  if (pin->down.done != kDoneSyncTimeout) break;
  { int rc = resume_after_sync_timeout(pin->down); if (rc) return rc; }
  n_enq = pin->down.passes; first_enq = true; guard = 0;
  }
  const Ctrl c = pin->down;
  pre_weights_fresh = false; pre_weights_pending = false;      // (a solve that queued no pass must not leave them to a later stand-alone pass)
  if (std::getenv("VICALIB_AMD_TIMING") && enqueue_ms > 0.0)
    std::fprintf(stderr, "[vicalib_amd]   solve: %d passes enqueued in %.3f ms of host time (batched schedule), %d decided; waited %.3f ms for the device, %.3f ms in all\n",
                 n_enq, enqueue_ms, c.passes, wait_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tso0).count());
  expected_passes = std::max(1, c.passes);
  const int n = std::min(c.trace_len, trace_cap);
  std::vector<double> rows((size_t)std::max(n, 1) * kTraceCols);
  if (n <= 64) std::memcpy(rows.data(), pin->trace, (size_t)n * kTraceCols * 8);
  else HIP_OK(hipMemcpy(rows.data(), d_trace.p, (size_t)n * kTraceCols * 8, hipMemcpyDeviceToHost));
  {
    std::lock_guard<std::mutex> lk(result_mutex);
    for (int i = 0; i < n; ++i) {
      const double* r = &rows[(size_t)i * kTraceCols];
      IterRecord rec = {(int)r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], (int)r[8], (int)r[9]};
      trace.push_back(rec);
    }
  }
  cur = c.cur;
  num_iterations += (unsigned)c.num_callbacks;
  jac_sweeps += c.jac_sweeps; res_sweeps += c.res_sweeps;
  last_iters = c.iter;
  *final_cost = c.cost;
  switch (c.done) {
    case kDoneConvergence: *term = kConvergence; break;
    case kDoneUserSuccess: *term = kUserSuccess; break;
    case kDoneFailure: *term = kFailure; break;
    default: *term = kNoConvergence; break;
  }
  return VC_OK;
}

int vc_calibrator::linearize_hold(double radius, double* cost) {
  Ctrl c;
  init_ctrl(&c);
  c.hold = 1; c.radius = radius; c.first = 0;
  { int rcu = upload_ctrl(&c); if (rcu) return rcu; }
  int rc = enqueue_pass(true, true); if (rc) return rc;
  finish_batch();
  HIP_OK(hipMemcpyAsync(&c, ctrl_result(), sizeof(Ctrl), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  if (cost) *cost = c.cost;
  return VC_OK;
}

int vc_calibrator::compute_rmse() {
  const int C = (int)cams.size();
  launch_reproj_res(dv, cur, 1.0, stream);
  launch_cam_sq(dv, d_tmp.p, stream);
  int rc = do_allreduce(d_tmp.p, 2 * C, 0); if (rc) return rc;
  double h[2 * kMaxCams];
  HIP_OK(hipMemcpyAsync(h, d_tmp.p, sizeof(double) * 2 * C, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  std::lock_guard<std::mutex> lk(result_mutex);
  cam_rmse.assign(C, 0.0);
  for (int c = 0; c < C; ++c) cam_rmse[c] = std::sqrt(0.5 * h[2 * c] / h[2 * c + 1]);
  return VC_OK;
}

int vc_calibrator::remove_outliers_pass() {
  const int C = (int)cams.size();
  std::vector<double> th(C);
  for (int c = 0; c < C; ++c) th[c] = outlier_threshold * cam_rmse[c];
  HIP_OK(hipMemcpyAsync(d_tmp.p + 32, th.data(), C * 8, hipMemcpyHostToDevice, stream));
  launch_outlier_mask(dv, cur, d_tmp.p + 32, d_mask.p, stream);
  std::vector<unsigned char> mask((size_t)dv.n_obs);
  if (!mask.empty()) HIP_OK(hipMemcpyAsync(mask.data(), d_mask.p, mask.size(), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  int rc = download_state(); if (rc) return rc;
  // The reference removes the blocks of the LATEST copy only (:911-914).  Vision-only: there is one copy, the
  // corner is gone.  With the IMU the stage loop re-adds every block right after (SetupProblem :641-649), so
  // an outlier ends up with one copy fewer than the inliers.
  const signed char mark = calibrate_imu ? 2 : 1;
  for (size_t k = 0; k < mask.size(); ++k) if (mask[k] && o_removed[h_obs_index[k]] == 0) o_removed[h_obs_index[k]] = mark;
  device_dirty = true; obs_dirty = true;
  return VC_OK;
}

int vc_calibrator::init_gravity() {
  const int N = (int)frames.size(), n = (int)imu_t.size();
  gravity_initialized = true;
  // the middle frame of the WHOLE problem; when the frames are sharded its owner computes, everybody receives
  long mid = N / 2; bool mine = true;
  if (world > 1) {
    std::vector<double> table;
    int rc = gather_shard_info(&table); if (rc) return rc;
    mid = global_total / 2 - global_first;
    mine = mid >= 0 && mid < N;
  }
  double g[2] = {0.0, 0.0};
  if (mine && N > 0 && n > 0) {
    const HostFrame& fr = frames[mid];
    double a[3];
    const double time = fr.time;
    if (imu_t[0] > time) { for (int k = 0; k < 3; ++k) a[k] = imu_a[k]; }
    else if (imu_t[n - 1] <= time || n < 2) { for (int k = 0; k < 3; ++k) a[k] = imu_a[3 * (size_t)(n - 1) + k]; }
    else {
      int lo = 0, hi = n - 1;
      while (hi - lo > 1) { const int m2 = (lo + hi) >> 1; if (imu_t[m2] <= time) lo = m2; else hi = m2; }
      const double f = (time - imu_t[lo]) / (imu_t[lo + 1] - imu_t[lo]);
      for (int k = 0; k < 3; ++k) a[k] = imu_a[3 * (size_t)lo + k] * (1.0 - f) + imu_a[3 * (size_t)(lo + 1) + k] * f;
    }
    const double nrm = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const double gb[3] = {a[0] / nrm, a[1] / nrm, a[2] / nrm};
    double gw[3];
    quat_rotate(fr.T, gb, gw);
    g[0] = std::asin(gw[1]);
    g[1] = std::asin(-gw[0] / std::cos(g[0]));
  }
  if (world > 1) {
    std::vector<double> v = {g[0], g[1]};
    int rc = host_allreduce_sum(v); if (rc) return rc;
    g[0] = v[0]; g[1] = v[1];
  } else if (!(N > 0 && n > 0)) return VC_OK;
  { std::lock_guard<std::mutex> lk(result_mutex); g_dir[0] = g[0]; g_dir[1] = g[1]; }
  return VC_OK;
}

int vc_calibrator::solve() {
  // the worker thread of Start() (and any caller's thread) starts on device 0: bind this calibrator's device first
  HIP_OK(hipSetDevice(device));
  ++solve_epoch;
  // is_finished_ is sticky until Clear() (vicalibrator.h:246, :922): a finished calibrator's Start()/Solve() returns at once
  int status = VC_OK;
  int stages_done = 0;
  while (should_run && !is_finished) {
    if (is_visual_active) vis_mult += 1;                      // SetupProblem re-adds every block (:641-649)
    if (calibrate_imu && is_inertial_active) imu_mult += 1;   // :651-655
    if (is_inertial_active && !rotation_only && !gravity_initialized) { status = init_gravity(); if (status) break; }   // :927-949
    device_dirty = true;                                      // constancy flags may have changed
    if (stage_limit >= 0 && stages_done++ >= stage_limit) break;   // bench / test hook: the next stage is set up, not run
    bool stage_done = false;
    int inner = 0;
    // "Crank optimization" (:952): the same problem is solved again while Ceres reports NO_CONVERGENCE.  The reference
    // loops without bound; here a run of kMaxRepeats unconverged solves returns VC_ERR_NO_CONVERGENCE with the
    // multiplicities untouched (the outer loop is NOT re-entered, which would re-add every block).
    while (!stage_done && should_run && !is_finished) {
      if (inner++ >= kMaxRepeats) { status = VC_ERR_NO_CONVERGENCE; break; }
      if (o_frame.empty()) { is_finished = true; break; }
      Termination t; double fc = 0; long nr = 1;
      const bool timing = std::getenv("VICALIB_AMD_TIMING") != nullptr;
      auto now = []() { return std::chrono::steady_clock::now(); };
      auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      const auto t0 = now();
      if (device_dirty) { status = upload(); if (status) break; }
      const auto t1 = now();
      status = solve_once(&t, &fc, &nr); if (status) break;
      const auto t2 = now();
      status = compute_rmse(); if (status) break;
      status = download_state(); if (status) break;
      if (timing) std::fprintf(stderr, "[vicalib_amd] stage %d: upload %.3f ms, solve %.3f ms (%d iterations), rmse + download %.3f ms\n", stage, ms(t0, t1), ms(t1, t2), last_iters, ms(t2, now()));
      { std::lock_guard<std::mutex> lk(result_mutex); mse = fc / (double)std::max(1L, nr); }
      ++stage;
      if (t != kNoConvergence && calibrate_imu) {
        if (!is_inertial_active) is_inertial_active = true;                          // :978-981
        else if (rotation_only) { rotation_only = false; is_bias_active = true; }   // :982-990
        else if (!is_scale_active) is_scale_active = true;                           // :991-994
        else if (remove_outliers && !outliers_removed) { status = remove_outliers_pass(); outliers_removed = true; }
        else is_finished = true;
        stage_done = true;
      } else if (t != kNoConvergence) {
        if (remove_outliers && !outliers_removed) { status = remove_outliers_pass(); outliers_removed = true; }
        else is_finished = true;
      }
      if (status) break;
    }
    if (status) break;
  }
  if (!device_dirty) { int rc = download_state(); if (!status) status = rc; }
  return status;
}
