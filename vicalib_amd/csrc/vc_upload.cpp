// vc_upload.cpp -- SetupProblem on the device: the reduced-system layout (constancy rules, vicalibrator.h:548-679) and the per-stage upload
// (tile-sorted observations, layout tables, state buffers, chain / IMU buffers; one packed staging image per stage).
#include "vc_calibrator.hpp"

int vc_calibrator::build_layout(std::vector<int>& col_cam, std::vector<int>& col_local) {
  const int C = (int)cams.size();
  cam_flags.assign(C, 0); cam_col0.assign(C, 0);
  int o = 0;
  col_cam.clear(); col_local.clear();
  for (int c = 0; c < C; ++c) {
    bool rf = true, tf = true;
    if (c == 0) {                               // vicalibrator.h:572-587
      if (!is_inertial_active) { rf = false; tf = false; } else { rf = true; tf = !rotation_only; }
    }
    int fl = 0;
    if (rf) fl |= kCamRotFree;
    if (tf) fl |= kCamTransFree;
    if (!fix_intrinsics) fl |= kCamKFree;       // :591-593
    cam_flags[c] = fl; cam_col0[c] = o;
    const int nc = cam_ncols(fl, cams[c].nk);
    for (int i = 0; i < nc; ++i) { col_cam.push_back(c); col_local.push_back(i); }
    o += nc;
  }
  for (int a = 0; a < 15; ++a) imu_param_col[a] = -1;
  if (imu_on()) {
    auto add = [&](int first, int n) { for (int i = 0; i < n; ++i) { imu_param_col[first + i] = o++; col_cam.push_back(-1); col_local.push_back(0); } };
    if (!rotation_only) add(0, 2);            // gravity: constant while rotation-only (vicalibrator.h:657-660, :986)
    if (is_bias_active) add(2, 6);            // :663-666, :990
    if (is_scale_active) add(8, 6);           // :668-671, :994
    if (optimize_time_offset) add(14, 1);     // :673-676
  }
  return o;
}

int vc_calibrator::upload() {
  RoctxRange rr("vicalib_amd: upload (SetupProblem of a stage)");
  HIP_OK(hipSetDevice(device));
  drop_graphs();
  pre_weights_fresh = false; pre_weights_pending = false;      // (the state is about to change under them)
  const bool up_timing = std::getenv("VICALIB_AMD_TIMING") != nullptr;
  const auto up_t0 = std::chrono::steady_clock::now();
  auto up_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - up_t0).count(); };
  double up_a = 0, up_b = 0, up_c = 0;
  const int Nown = (int)frames.size(), C = (int)cams.size();
  if (C > kMaxCams) return VC_ERR_UNSUPPORTED;
  // ---- frame-sharded IMU chain: this rank's first frame is a separator of the reduced system (rank > 0) and the
  // next rank's separator is kept here as a ghost frame (the IMU block that ends in it is ours)
  const bool shard_imu = world > 1 && imu_on();
  bool ghost = false;
  HostFrame ghost_frame{};
  if (shard_imu) {
    if (Nown < 2) return VC_ERR_BAD_ARG;
    std::vector<double> table;
    int rc = gather_shard_info(&table); if (rc) return rc;
    if (rank + 1 < world) {
      const double* o = table.data() + (size_t)(rank + 1) * 12;
      std::memcpy(ghost_frame.T, o, 56); std::memcpy(ghost_frame.v, o + 7, 24); ghost_frame.time = o[10];
      ghost = true;
    }
  }
  const int N = Nown + (ghost ? 1 : 0);
  auto frame_at = [&](int f) -> const HostFrame& { return f < Nown ? frames[f] : ghost_frame; };
  // ---- tiles: sort the active observations by (frame, camera) -- only when the observation set changed (the stage
  // machine re-uploads state and layout four times per calibration, the 10 ms sort / de-dup / 7 MB copy happen once)
  if (obs_dirty) {
    const size_t n_all = o_frame.size();
    std::vector<int> idx; idx.reserve(n_all);
    for (size_t i = 0; i < n_all; ++i) if (o_removed[i] != 1) idx.push_back((int)i);
    bool in_order = true;             // the usual caller adds frame by frame, camera by camera: nothing to sort then
    for (size_t k = 1; k < idx.size() && in_order; ++k) {
      const int a = idx[k - 1], b = idx[k];
      in_order = o_frame[a] != o_frame[b] ? o_frame[a] < o_frame[b] : o_cam[a] <= o_cam[b];
    }
    if (!in_order) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
      return o_frame[a] != o_frame[b] ? o_frame[a] < o_frame[b] : o_cam[a] < o_cam[b]; });
    h_obs_index = idx;
    h_tile_frame.clear(); h_tile_cam.clear(); h_tile_off.clear();
    if (pts.size() > kObsPointMask + 1) return VC_ERR_TOO_MANY_POINTS;
    const size_t n_o = idx.size();
    if (n_o * sizeof(double2) > ((size_t)64 << 20)) {
      // (very large observation sets keep the streaming pageable copies: a page-locked image of their size costs more to allocate
      //  than it saves)
      std::vector<double2> uv(n_o);
      std::vector<unsigned short> pt(n_o);
      n_one_less = 0;
      for (size_t k = 0; k < n_o; ++k) {
        const int i = idx[k];
        if (k == 0 || o_frame[i] != o_frame[idx[k - 1]] || o_cam[i] != o_cam[idx[k - 1]]) {
          h_tile_frame.push_back(o_frame[i]); h_tile_cam.push_back(o_cam[i]); h_tile_off.push_back((int)k);
        }
        pt[k] = (unsigned short)(o_pid[i] | (o_removed[i] == 2 ? kObsOneLess : 0));
        if (o_removed[i] == 2) ++n_one_less;
        uv[k] = make_double2(o_pc[2 * (size_t)i], o_pc[2 * (size_t)i + 1]);
      }
      h_tile_off.push_back((int)n_o);
      n_points_dev = pts.size();
      HIP_OK(d_uv.upload(uv, stream)); HIP_OK(d_pt.upload(pt, stream)); HIP_OK(d_points.upload(pts.xyz, stream));
      HIP_OK(d_tile_frame.upload(h_tile_frame, stream)); HIP_OK(d_tile_cam.upload(h_tile_cam, stream));
      HIP_OK(d_tile_off.upload(h_tile_off, stream));
      HIP_OK(d_mask.alloc(std::max<size_t>(n_o, 1)));
      HIP_OK(hipStreamSynchronize(stream));        // the staging vectors go out of scope
    } else {
    // detections and point indices are written straight into the page-locked staging image (no pageable vector in between: the 6 MB
    // copy of cfg3's detections was 1.5 ms of the first stage's upload)
    pack.begin();
    const size_t pt_bytes = ((n_o * sizeof(unsigned short) + 3) / 4) * 4;
    if (pack.reserve(n_o * sizeof(double2) + pt_bytes + pts.xyz.size() * 8 + (h_tile_frame.capacity() + n_o / 8 + 64) * 16 + 4096) != hipSuccess) return VC_ERR_NO_DEVICE;
    size_t uv_off = 0, pt_off = 0;
    double2* uv = (double2*)pack.slot(n_o * sizeof(double2), &uv_off);
    unsigned short* pt = (unsigned short*)pack.slot(pt_bytes, &pt_off);
    if (!uv || !pt) return VC_ERR_NO_DEVICE;
    if (pt_bytes > n_o * sizeof(unsigned short)) pt[n_o] = 0;      // (padding word)
    n_one_less = 0;
    for (size_t k = 0; k < idx.size(); ++k) {
      const int i = idx[k];
      if (k == 0 || o_frame[i] != o_frame[idx[k - 1]] || o_cam[i] != o_cam[idx[k - 1]]) {
        h_tile_frame.push_back(o_frame[i]); h_tile_cam.push_back(o_cam[i]); h_tile_off.push_back((int)k);
      }
      pt[k] = (unsigned short)(o_pid[i] | (o_removed[i] == 2 ? kObsOneLess : 0));
      if (o_removed[i] == 2) ++n_one_less;
      uv[k] = make_double2(o_pc[2 * (size_t)i], o_pc[2 * (size_t)i + 1]);
    }
    h_tile_off.push_back((int)idx.size());
    n_points_dev = pts.size();
    HIP_OK(d_uv.alloc(std::max<size_t>(n_o, 1))); HIP_OK(d_pt.alloc(pt_bytes / sizeof(unsigned short) + 2));
    pack.seg(d_uv.p, uv_off, n_o * sizeof(double2)); pack.seg(d_pt.p, pt_off, pt_bytes);
    HIP_OK(pack.add(d_points, pts.xyz));
    HIP_OK(pack.add(d_tile_frame, h_tile_frame)); HIP_OK(pack.add(d_tile_cam, h_tile_cam));
    HIP_OK(pack.add(d_tile_off, h_tile_off));
    HIP_OK(d_mask.alloc(std::max<size_t>(idx.size(), 1)));
    HIP_OK(pack.flush(stream));
    HIP_OK(hipStreamSynchronize(stream));        // the staging image is re-used below
    }
    obs_dirty = false;
  }
  up_a = up_ms();
  const size_t n_active = h_obs_index.size();
  const int T = (int)h_tile_frame.size();
  std::vector<int> frame_tile_off(N + 1, T), frame_cam_tile((size_t)N * std::max(C, 1), -1);
  {
    int t = 0;
    for (int f = 0; f <= N; ++f) { while (t < T && h_tile_frame[t] < f) ++t; frame_tile_off[f] = t; }
    for (int t2 = 0; t2 < T; ++t2) frame_cam_tile[(size_t)h_tile_frame[t2] * C + h_tile_cam[t2]] = t2;
  }
  std::vector<int> col_cam, col_local, cam_model(C);
  const int D0 = build_layout(col_cam, col_local);
  const int D = D0 + (shard_imu ? 9 * (world - 1) : 0);          // + one 9-column separator per shard boundary
  col_cam.resize(D, -1); col_local.resize(D, 0);
  // the widest column layout this problem can reach (every camera and IMU parameter free): the stage machine only widens the
  // layout, and growing a multi-megabyte buffer is a free + malloc of a few hundred microseconds -- the big ones get their
  // final capacity at the first upload
  int Dmax = (imu_on() ? 15 : 0) + (shard_imu ? 9 * (world - 1) : 0);
  for (int c = 0; c < C; ++c) Dmax += 6 + cams[c].nk;
  Dmax = std::max(Dmax, D);
  // the reduced solve lives in LDS (packed lower triangle + 12 KB of staging), the chain Gram handles 12 column tiles
  if (((size_t)(D + 1) * (D + 2) / 2 + 3 * (D + 1) + 528) * sizeof(double) + 13 * 1024 > 160 * 1024 || D + 1 > 12 * 16) return VC_ERR_UNSUPPORTED;
  // vision-only passes: k_frame_schur keeps at most 36 column-tile pairs (8 column tiles: D <= 127; eight 16-column cameras
  // with the first one's extrinsics fixed are 122)
  if (!imu_on() && D + 1 > 8 * 16) return VC_ERR_UNSUPPORTED;
  for (int c = 0; c < C; ++c) cam_model[c] = cams[c].model;
  up_b = up_ms();
  // ---- upload ---------------------------------------------------------------------------------
  // Everything small goes through ONE page-locked staging image, one copy and one scatter kernel (Packer above; round 3: ~30
  // pageable copies and ~8 fills per stage, 4.1 of the 19 ms of a complete cfg3 calibration).  Arrays above 4 MB keep their own copy.
  pack.begin();
  auto up = [&](auto& d, const auto& h) -> hipError_t {
    if (h.size() * sizeof(h[0]) > (size_t)4 << 20) return d.upload(h, stream);
    return pack.add(d, h);
  };
  // the same array into a second / third buffer (state double buffer, initial-state copy)
  auto up_also = [&](auto& d, const auto& h) -> hipError_t {
    if (h.size() * sizeof(h[0]) > (size_t)4 << 20) return d.upload(h, stream);
    return pack.also(d, h.size());
  };
  {   // tile headers: depend on the tile layout and on this stage's column layout
    std::vector<TileHdr> hdr((size_t)T);
    for (int t = 0; t < T; ++t) {
      TileHdr& h = hdr[(size_t)t];
      const int f = h_tile_frame[t];
      h.frame = f; h.cam = h_tile_cam[t]; h.t0 = frame_tile_off[f]; h.nt = frame_tile_off[f + 1] - frame_tile_off[f];
      h.off = h_tile_off[t]; h.cnt = h_tile_off[t + 1] - h_tile_off[t]; h.model = cams[h.cam].model; h.pad = 0;
      h.col0 = 0; h.ncols = 0;
      for (int k = 0; k < h.nt && k < kMaxCams; ++k) {
        const int c = h_tile_cam[h.t0 + k];
        h.col0 |= (unsigned long long)(cam_col0[c] & 0xff) << (8 * k);
        h.ncols |= (unsigned long long)(cam_ncols(cam_flags[c], cams[c].nk) & 0xff) << (8 * k);
      }
    }
    HIP_OK(up(d_tile_hdr, hdr));
  }
  HIP_OK(up(d_frame_tile_off, frame_tile_off));
  HIP_OK(up(d_frame_cam_tile, frame_cam_tile)); HIP_OK(up(d_cam_model, cam_model));
  HIP_OK(up(d_cam_flags, cam_flags)); HIP_OK(up(d_cam_col0, cam_col0));
  HIP_OK(up(d_col_cam, col_cam)); HIP_OK(up(d_col_local, col_local));
  std::vector<double> poses((size_t)N * kPoseStride, 0.0), camrec((size_t)C * kCamStride, 0.0);
  for (int f = 0; f < N; ++f) std::memcpy(&poses[(size_t)f * kPoseStride], frame_at(f).T, 56);
  for (int c = 0; c < C; ++c) {
    std::memcpy(&camrec[(size_t)c * kCamStride], cams[c].T_ck, 56);
    std::memcpy(&camrec[(size_t)c * kCamStride + kCamK], cams[c].K, cams[c].nk * 8);
  }
  cur = 0;
  HIP_OK(up(d_pose[0], poses)); HIP_OK(up_also(d_pose[1], poses)); HIP_OK(up_also(d_pose_init, poses));
  HIP_OK(up(d_cam[0], camrec)); HIP_OK(up_also(d_cam[1], camrec)); HIP_OK(up_also(d_cam_init, camrec));
  // one wavefront per frame, 4 frames per group: up to 2048 chunks (= partial sums) before chunks grow
  int chunk_frames = std::max(4, (((N + 2047) / 2048) + 3) / 4 * 4);
  // wide borders: a chunk's partial record is D^2 doubles -- written once and read once per pass; keep all of them under ~64 MB
  // (8 cameras, 6250 frames per rank: 197 MB at 4 frames per chunk, k_part_sum 47 -> 19 us at 16)
  while (chunk_frames < 16 && (double)((N + chunk_frames - 1) / chunk_frames) * ((double)D * D + D + (C + 1) * kGStride) * 8.0 > 64e6) chunk_frames *= 2;
  { const char* e = std::getenv("VICALIB_AMD_CHUNK_FRAMES"); if (e && std::atoi(e) >= 4) chunk_frames = std::atoi(e) / 4 * 4; }      // (A/B hook)
  // the chain assembly folded into the bottom level of the elimination (k_chain_l0): a function of the problem only, never of the
  // hand-over mode; its chunk of the partial sums is the group of 8 frames.  VICALIB_AMD_FOLD_L0=0: the two kernels apart (A/B)
  static const bool fold_env = [] { const char* e = std::getenv("VICALIB_AMD_FOLD_L0"); return !(e && e[0] == '0'); }();
  const bool fold = fold_env && imu_on() && !sharded() && !shard_imu && chain_fold_supported(N, D, C);
  if (fold) chunk_frames = 8;
  const int n_chunks = std::max(1, (N + chunk_frames - 1) / chunk_frames);
  const int part_stride = D * D + D + C * kGStride + (imu_on() ? kGStride : 0) + 2;     // ... + [x2 of observation-less frames, chunk cost] (vision path)
  for (int b = 0; b < 2; ++b) {
    HIP_OK(d_G[b].alloc((size_t)std::max(T, 1) * kGPack)); HIP_OK(d_tile_cost[b].alloc(std::max(T, 1)));
    HIP_OK(hipMemsetAsync(d_G[b].p, 0, (size_t)std::max(T, 1) * kGPack * sizeof(double), stream));   // sub-blocks a model never writes stay 0
  } HIP_OK(d_tile_trial.alloc((size_t)std::max(T, 1) * 2));
  HIP_OK(d_Y.alloc((size_t)std::max(T, 1) * kYStride)); HIP_OK(d_fr.alloc((size_t)std::max(N, 1) * kFrStride));
  HIP_OK(d_fdiag.alloc((size_t)std::max(N, 1) * 6)); HIP_OK(d_fscale2.alloc((size_t)std::max(N, 1) * 6));
  HIP_OK(d_part.alloc((size_t)(n_chunks + 1) * ((size_t)Dmax * Dmax + Dmax + C * kGStride + kGStride + 2)));      // (+ 1: the chain's top level, early Gram)
  HIP_OK(d_part.alloc((size_t)(n_chunks + 1) * part_stride)); HIP_OK(d_part_total.alloc((size_t)part_stride)); HIP_OK(d_Sbuf.alloc((size_t)D * D + 3 * D + 2)); HIP_OK(d_hadd.alloc((size_t)D * D + 3 * D + 2)); HIP_OK(d_part_ready.alloc((size_t)n_chunks + 1)); HIP_OK(hipMemsetAsync(d_part_ready.p, 0, ((size_t)n_chunks + 1) * sizeof(long long), stream));
  HIP_OK(d_sdiag.alloc(D)); HIP_OK(d_sscale2.alloc(D)); HIP_OK(d_slam.alloc(D)); HIP_OK(d_delta_s.alloc(D));
  HIP_OK(d_fpart.alloc((size_t)std::max(N, 1) * kNumScal));
  trace_cap = max_iters + 8; HIP_OK(d_trace.alloc((size_t)trace_cap * kTraceCols)); HIP_OK(d_ctrl.alloc(2));
  HIP_OK(hipMemsetAsync(d_part.p, 0, (size_t)(n_chunks + 1) * part_stride * sizeof(double), stream));
  pack.zero(d_fpart.p, (size_t)std::max(N, 1) * kNumScal * sizeof(double));
  HIP_OK(d_scal.alloc(2 * kNumScal)); HIP_OK(d_flags.alloc(8));
  HIP_OK(d_wgpart.alloc((size_t)std::max(1, (T + 3) / 4) * kNumScal));
  pack.zero(d_scal.p, 2 * kNumScal * sizeof(double));
  HIP_OK(d_tmp.alloc(128));       // [0,16) per-camera sums, [32,40) outlier thresholds, [64,96) profiling stamps
  pack.zero(d_tmp.p, 128 * sizeof(double));
  pack.zero(d_flags.p, 8 * sizeof(int));
  pack.zero(d_ctrl.p, 2 * sizeof(Ctrl));
  pack.zero(d_delta_s.p, std::max(D, 1) * sizeof(double));
  for (int c = 0; c < kMaxCams; ++c) { dv.cd[c].model = 0; dv.cd[c].flags = 0; dv.cd[c].col0 = 0; dv.cd[c].ncols = 0; }
  for (int c = 0; c < C; ++c) { dv.cd[c].model = cams[c].model; dv.cd[c].flags = cam_flags[c]; dv.cd[c].col0 = cam_col0[c]; dv.cd[c].ncols = cam_ncols(cam_flags[c], cams[c].nk); }
  dv.n_frames = N; dv.n_cams = C; dv.n_tiles = T; dv.n_points = n_points_dev; dv.D = D;
  if (!reduced_fits(dv)) {
    std::fprintf(stderr, "vicalib_amd: a reduced system of %d shared parameters (cameras, IMU, %d separator frames) does not fit k_reduced's LDS image (limit: 179)\n", D, shard_imu ? world - 1 : 0);
    return VC_ERR_UNSUPPORTED;
  }
  dv.n_chunks = n_chunks; dv.n_part = n_chunks; dv.chunk_frames = chunk_frames; dv.n_obs = (long long)n_active;
  dv.obs_uv = d_uv.p; dv.obs_pt = d_pt.p; dv.points = d_points.p;
  dv.tile_hdr = d_tile_hdr.p;
  dv.tile_frame = d_tile_frame.p; dv.tile_cam = d_tile_cam.p; dv.tile_off = d_tile_off.p;
  dv.frame_tile_off = d_frame_tile_off.p; dv.frame_cam_tile = d_frame_cam_tile.p;
  dv.cam_model = d_cam_model.p; dv.cam_flags = d_cam_flags.p; dv.cam_col0 = d_cam_col0.p;
  dv.col_cam = d_col_cam.p; dv.col_local = d_col_local.p;
  dv.poses[0] = d_pose[0].p; dv.poses[1] = d_pose[1].p; dv.cams[0] = d_cam[0].p; dv.cams[1] = d_cam[1].p;
  for (int b = 0; b < 2; ++b) { dv.Gb[b] = d_G[b].p; dv.tile_costb[b] = d_tile_cost[b].p; }
  dv.fused = 1;
 dv.tile_trial = d_tile_trial.p; dv.Y = d_Y.p; dv.fr = d_fr.p;
  dv.fdiag = d_fdiag.p; dv.fscale2 = d_fscale2.p; dv.part = d_part.p; dv.part_total = d_part_total.p; dv.Sbuf = d_Sbuf.p; dv.hadd = d_hadd.p; dv.part_ready = d_part_ready.p;
  dv.sdiag = d_sdiag.p; dv.sscale2 = d_sscale2.p; dv.slam = d_slam.p; dv.delta_s = d_delta_s.p;
  dv.fpart = d_fpart.p; dv.scal = d_scal.p; dv.flags = d_flags.p;
  dv.pre_backsub = (T > 2048) ? 1 : 0;
  { const char* e = std::getenv("VICALIB_AMD_PRE_BACKSUB"); if (e && (e[0] == '0' || e[0] == '1')) dv.pre_backsub = e[0] - '0'; }   // test hook     // 1024 SIMDs x 2 resident waves: beyond that the per-tile repeat of the back-substitution is pure cost
  {
    // bottom-level groups of the chain elimination (launch_chain_solve_*: groups of 8 while more than 7 frames are active)
    const int cm = chain_group_size();
    const int groups = (N > cm - 1) ? (N - 1) / cm + 1 : 1;
    HIP_OK(d_grp_part.alloc((size_t)groups * kNumScal)); HIP_OK(d_wg_trial.alloc((size_t)std::max(1, (T + 3) / 4)));
    HIP_OK(d_wg_imu_trial.alloc((size_t)std::max(1, (N + 6) / 8)));
    dv.grp_part = d_grp_part.p; dv.wg_trial = d_wg_trial.p; dv.wg_imu_trial = d_wg_imu_trial.p; dv.n_chain_groups = groups;
  }
  dv.wgpart = d_wgpart.p; dv.merged = 0; dv.par = 0; dv.ctrl_prev = d_ctrl.p + 1;
  dv.part_stride = part_stride; dv.ctrl = d_ctrl.p; dv.trace = d_trace.p; dv.dbg = (long long*)(d_tmp.p + 64);
  // ---- inertial terms ------------------------------------------------------------------------------
  std::vector<double> vels((size_t)std::max(N, 1) * 4, 0.0), imus(16, 0.0), ftime(std::max(N, 1), 0.0);
  for (int f = 0; f < N; ++f) { std::memcpy(&vels[(size_t)f * 4], frame_at(f).v, 24); ftime[f] = frame_at(f).time; }
  imus[0] = g_dir[0]; imus[1] = g_dir[1];
  for (int i = 0; i < 6; ++i) { imus[2 + i] = biases[i]; imus[8 + i] = scale[i]; }
  imus[14] = time_offset;
  HIP_OK(up(d_vel[0], vels)); HIP_OK(up_also(d_vel[1], vels)); HIP_OK(up_also(d_vel_init, vels));
  HIP_OK(up(d_imus[0], imus)); HIP_OK(up_also(d_imus[1], imus)); HIP_OK(up_also(d_imus_init, imus));
  HIP_OK(up(d_frame_time, ftime));
  dv.imu_on = imu_on() ? 1 : 0; dv.rotation_only = rotation_only ? 1 : 0;
  dv.weights_on = (is_inertial_active && !rotation_only) ? 1 : 0;
  dv.n_imu = (int)imu_t.size();
  dv.imu_avg_dt = imu_average_dt(imu_t.data(), (int)imu_t.size());
  dv.gyro_sigma = gyro_sigma; dv.accel_sigma = accel_sigma;
  for (int a = 0; a < 15; ++a) dv.imu_param_col[a] = imu_param_col[a];
  dv.ldw = (((D + 1 + 15) / 16) * 16 % 32 == 0) ? ((D + 1 + 15) / 16) * 16 + 16 : ((D + 1 + 15) / 16) * 16;
  dv.ldx = dv.ldw + 32;
  {
    // early Gram (vc_device.h): narrow reduced systems of a single process; a function of the problem only, never of the hand-over
    // mode -- a solve resumed with events after a flag time-out must repeat the withheld passes with the same arithmetic
    static const bool early_env = [] { const char* e = std::getenv("VICALIB_AMD_EARLY_GRAM"); return !(e && e[0] == '0'); }();
    // (where it pays: the top level's one group must outlast the Gram sums beside it -- at 6250 frames x 8 cameras, D = 115, the top
    //  level is two frames and the sums take 50 us: 0.906 -> 0.938 ms per pass with them in its launch; at 2500 frames, D = 67: -4.5 us)
    const bool narrow = D + 1 + 27 <= 128;      // at most two image columns per lane
    dv.gram_top_stride = (early_env && dv.imu_on && N >= 1 && N <= 4096 && narrow) ? chain_top_stride(N) : 0;
    dv.fold_l0 = fold ? 1 : 0;
    // the back-substitution as one launch (k_chain_back_path): round 6 -- any border width, sharded passes (pinned frames) included
    static const bool path_env = [] { const char* e = std::getenv("VICALIB_AMD_BACK_PATH"); return !(e && e[0] == '0'); }();
    // (every bottom group recomputes the levels above it: (levels + 1) x the level-by-level form's work -- free while the bottom groups
    //  fit the chip in one round, 120 us against 71 at 6250 frames x D = 115 (profiles/r06_per_rank_passes.txt): up to 4096 frames)
    dv.back_path = (path_env && dv.imu_on && N <= 4096) ? 1 : 0;
    // the top level's own frames: added by k_reduced (single process, narrow system) or a partial record of their own
    top_gram_launch = dv.gram_top_stride > 0 && !(D <= kEarlyTopD && !sharded());
    dv.n_part = dv.n_chunks + (top_gram_launch ? 1 : 0);
  }
  dv.pin_first = (shard_imu && rank > 0) ? 1 : 0; dv.pin_last = ghost ? 1 : 0;
  dv.sep_col0 = D0 + 9 * (rank - 1); dv.sep_col1 = D0 + 9 * rank;
  HIP_OK(d_sep_strip.alloc((size_t)2 * 9 * dv.ldw)); dv.sep_strip = d_sep_strip.p;
  HIP_OK(d_gath.alloc((size_t)world * kNumScal)); dv.gath = d_gath.p; dv.rank = rank; dv.world = world;
  if (dv.imu_on) {
    if (imu_uploaded != imu_t.size() || imu_uploaded_last != (imu_t.empty() ? 0.0 : imu_t.back())) {      // the samples do not change from stage to stage
      HIP_OK(d_imu_t.upload(imu_t, stream)); HIP_OK(d_imu_w.upload(imu_w, stream)); HIP_OK(d_imu_a.upload(imu_a, stream));
      imu_uploaded = imu_t.size(); imu_uploaded_last = imu_t.empty() ? 0.0 : imu_t.back();
    }
    const size_t ns = (size_t)std::max(N - 1, 1);
    if (wsqrt_frames != (size_t)N) {          // initial weight 500 * I (vicalibrator.h:616); later stages keep the current weights
      std::vector<double> w(ns * 81, 0.0);
      for (size_t k = 0; k < ns; ++k) for (int i = 0; i < 9; ++i) w[k * 81 + i * 10] = 500.0;
      HIP_OK(d_wsqrt[0].upload(w, stream)); HIP_OK(d_wsqrt[1].upload(w, stream)); wsqrt_frames = (size_t)N; wcur = 0;
      HIP_OK(hipStreamSynchronize(stream));
    }
    for (int b = 0; b < 2; ++b) { HIP_OK(d_seg[b].alloc(ns * kSegStride)); HIP_OK(d_seg_cost[b].alloc(ns)); }
    HIP_OK(d_imu_delta_blk.alloc(ns * kBlockDeltaStride)); HIP_OK(d_imu_grav.alloc(32));
    const size_t nf = (size_t)std::max(N, 1);
    {
      const int ldw_max = (((Dmax + 1 + 15) / 16) * 16 % 32 == 0) ? ((Dmax + 1 + 15) / 16) * 16 + 16 : ((Dmax + 1 + 15) / 16) * 16;
      HIP_OK(d_cW.alloc(nf * 9 * (ldw_max + 32)));
      for (int b = 0; b < 2; ++b) HIP_OK(d_rX[b].alloc((nf / std::min(chain_group_size(), chain_group_size_upper()) + 2) * 9 * (ldw_max + 32)));
    }
    HIP_OK(d_cW.alloc(nf * 9 * dv.ldx)); HIP_OK(d_cdelta.alloc(nf * 9)); HIP_OK(d_ct0.alloc(nf * 9)); HIP_OK(d_cg.alloc(nf * 9)); HIP_OK(d_clam.alloc(nf * 9));
    HIP_OK(d_cdiag.alloc(nf * 9)); HIP_OK(d_cscale2.alloc(nf * 9));
    HIP_OK(d_cready.alloc(nf)); pack.zero(d_cready.p, nf * sizeof(long long));
    for (int b = 0; b < 2; ++b) HIP_OK(d_rX[b].alloc((nf / std::min(chain_group_size(), chain_group_size_upper()) + 2) * 9 * dv.ldx));
  }
  dv.imu_t = d_imu_t.p; dv.imu_w = d_imu_w.p; dv.imu_a = d_imu_a.p; dv.frame_time = d_frame_time.p;
  dv.vel[0] = d_vel[0].p; dv.vel[1] = d_vel[1].p; dv.imus[0] = d_imus[0].p; dv.imus[1] = d_imus[1].p;
  dv.wsqrtb[0] = d_wsqrt[0].p; dv.wsqrtb[1] = d_wsqrt[1].p;
  dv.imu_delta_blk = d_imu_delta_blk.p; dv.imu_grav = d_imu_grav.p;
  for (int b = 0; b < 2; ++b) { dv.segb[b] = d_seg[b].p; dv.seg_costb[b] = d_seg_cost[b].p; }
  dv.cW = d_cW.p; dv.cdelta = d_cdelta.p; dv.ct0 = d_ct0.p; dv.cg = d_cg.p;
  dv.clam = d_clam.p; dv.cdiag = d_cdiag.p; dv.cscale2 = d_cscale2.p; dv.cready = d_cready.p;
  for (int b = 0; b < 2; ++b) dv.rX[b] = d_rX[b].p;
  HIP_OK(pack.flush(stream));
  up_c = up_ms();
  HIP_OK(hipStreamSynchronize(stream));   // the staging vectors above go out of scope
  if (up_timing) std::fprintf(stderr, "[vicalib_amd]   upload: observations %.3f, layout %.3f, copies + allocations %.3f, drain %.3f ms\n", up_a, up_b - up_a, up_c - up_b, up_ms() - up_c);
  device_dirty = false;
  return VC_OK;
}
