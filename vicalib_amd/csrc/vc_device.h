// vc_device.h -- device-side views shared by the host driver (vc_calibrator.hpp and its translation units)
// and the HIP kernels (vc_kernels.hip).  HBM layout (see DESIGN.md "Data layout"):
//   observations  tile-sorted SoA: obs_uv[n] (16 B) + obs_pt[n] (u16 index into points[]) = 18 B / corner
//   tiles         (frame, camera) groups: tile_frame, tile_cam, tile_off[n_tiles+1]
//   frame poses   n_frames x 8 doubles [q(4) t(3) pad], double-buffered (accepted / trial)
//   cameras       n_cams x 24 doubles [T_ck(7) pad K(<=10) ...], double-buffered
// The Levenberg-Marquardt control state (radius, accepted buffer, flags, iteration trace) lives on the
// device (struct Ctrl): the host enqueues passes ahead of time and only polls `done`.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace vc {

constexpr int kMaxCams = 8;
// obs_pt[i]: bits 0..14 = index into points[], bit 15 = this corner has one residual-block copy fewer than Ctrl::mult
// (an outlier removed from the latest copy before SetupProblem re-added every block, vicalibrator.h:641-649, :911-914)
constexpr int kObsPointMask = 0x7fff, kObsOneLess = 0x8000;
constexpr int kGStride = 272;      // Gram record per tile: the 16 x 16 block + the 16-entry side vector at kGGrad (vc_math.hpp)
constexpr int kYStride = 96;       // 6 x 16 per tile
constexpr int kFrStride = 48;      // per frame: L(21) z(6) g(6) lam(6) 1/diag(L)(6) pad
constexpr int kFrL = 0, kFrZ = 21, kFrG = 27, kFrLam = 33, kFrDinv = 39;
constexpr int kNumScal = 8;
enum { kScGd = 0, kScDld = 1, kScStep2 = 2, kScX2 = 3, kScG2 = 4, kScCost = 5, kScGmax = 6, kScSq = 7 };
constexpr int kTraceCols = 10;     // iteration cost cost_change gmax gnorm step_norm rho radius accepted stage
constexpr int kSmallD = 32;        // reduced systems up to this width are solved by one wavefront; above it the workgroup-wide LDS factorisation is faster
constexpr int kEarlyTopD = kSmallD - 1;   // early Gram inside k_reduced: its three 16 x 16 column tiles cover the D + 1 <= 32 columns of [Y | z] (D = 32: the z column would fall outside)
constexpr int kSyncWords = 16;     // DevView::sync_flags

// termination codes in Ctrl::done (0 = keep running)
constexpr double kShardMark = 1048576.0;                 // sharded solves: "my pass is void" on top of a rank's failure count in the gathered step scalars
constexpr unsigned kProgressLikelyLast = 1u << 30;      // progress word, low half: Ctrl::done | this hint
enum { kRunning = 0, kDoneConvergence = 1, kDoneNoConvergence = 2, kDoneUserSuccess = 3, kDoneFailure = 4,
       kDoneSyncTimeout = 5 };   // a device-flag hand-over between the two streams ran into its bound: the pass is void, the host re-runs it with events

// Normal-equation record of one IMU block (k_imu_jac -> k_chain_init).  The block's 33 local columns are [frame j ("cur": pose 6,
// velocity 3) | frame j - 1 ("prev": 9) | IMU parameters 15].  Of the symmetric 33 x 33 matrix and the gradient only what the chain
// assembly reads is stored, once, grouped by reader -- frame j takes [kSegAcc, kSegApp), frame j - 1 takes [kSegApp, kSegHii), the
// shared IMU block is [kSegHii, kSegLen): 771 doubles instead of 33 x 33 + 33 = 1122, and each reader touches a third of them
// (the full blocks cost 35 of k_chain_init's 45 MB per pass at BASELINE cfg3).
//   kSegAcc cur x cur 9 x 9 | kSegWc cur x imu 9 x 15 | kSegGc gradient cur 9 | kSegApp prev x prev 9 x 9 | kSegBpc prev (rows) x cur
//   (cols) 9 x 9 | kSegWp prev x imu 9 x 15 | kSegGp gradient prev 9 | kSegHii imu x imu 15 x 15 | kSegGi gradient imu 15
constexpr int kSegAcc = 0, kSegWc = 81, kSegGc = 216, kSegApp = 225, kSegBpc = 306, kSegWp = 387, kSegGp = 522, kSegHii = 531,
              kSegGi = 756, kSegLen = 771, kSegStride = 776;
// entry e of the record = (row a, column b) of the 33 x 33 block, b = 33: the gradient's entry a
constexpr void seg_entry(int e, int* a, int* b) {
  if (e < kSegWc) { *a = e / 9; *b = e % 9; }
  else if (e < kSegGc) { const int k = e - kSegWc; *a = k / 15; *b = 18 + k % 15; }
  else if (e < kSegApp) { *a = e - kSegGc; *b = 33; }
  else if (e < kSegBpc) { const int k = e - kSegApp; *a = 9 + k / 9; *b = 9 + k % 9; }
  else if (e < kSegWp) { const int k = e - kSegBpc; *a = 9 + k / 9; *b = k % 9; }
  else if (e < kSegGp) { const int k = e - kSegWp; *a = 9 + k / 15; *b = 18 + k % 15; }
  else if (e < kSegHii) { *a = 9 + (e - kSegGp); *b = 33; }
  else if (e < kSegGi) { const int k = e - kSegHii; *a = 18 + k / 15; *b = 18 + k % 15; }
  else { *a = 18 + (e - kSegGi); *b = 33; }
}

struct Ctrl {
  double radius, decrease_factor, cost, gmax, gnorm, last_gnorm;
  double ftol, gtol, ptol, mult, imu_mult;
  double pend[kTraceCols];   // record of an accepted iteration waiting for the linearisation at the new point
  int cur;            // index of the accepted state buffer
  int reuse_diag;     // LevenbergMarquardtStrategy::reuse_diagonal_
  int need_lin;       // the next pass re-linearises (Jacobian sweep)
  int init_scale;     // first linearisation of a Solve: estimate the Jacobi scaling
  int iter, invalid, done, max_iters;
  int first, pending, trace_len, trace_cap;
  int stage, hold, num_callbacks, jac_sweeps;
  int res_sweeps, passes;
  int needs_decision;   // merged mode: the pass that used this record has produced a trial point nobody has judged yet
  int abort_seq;        // kDoneSyncTimeout: low 31 bits of the pass number (DevView::sync_seq) whose decision was withheld
  int likely_last;      // the step just accepted cut the cost by less than 1e3 x the function tolerance and by a tenth of the step before: the
  int pad1;             // next decision will very probably end the solve (published to the feeding host, which then queues nothing past it)
  double last_rel;      // relative cost change of the last accepted step
};

// per-camera descriptor, carried in the kernel arguments (scalar loads, no dependent global look-ups)
struct CamDesc { int model, flags, col0, ncols; };
// Everything a wavefront of k_trial needs to start on its tile, in one 48-byte record: one load instead of a chain of
// dependent ones (tile -> frame -> the frame's tiles -> their cameras -> their reduced-system columns).  col0 / ncols: one
// byte per tile of the frame (at most kMaxCams tiles per frame, at most 191 reduced columns).
struct TileHdr { int frame, cam, t0, nt, off, cnt, model, pad; unsigned long long col0, ncols; };

struct DevView {
  CamDesc cd[kMaxCams];
  int n_frames, n_cams, n_tiles, n_points, D, n_chunks, chunk_frames;
  int n_part;                      // partial records k_part_sum adds: n_chunks, + 1 where the chain's top level has a record of its own (launch_chain_gram_top)
  long long n_obs;
  const double2* obs_uv;
  const unsigned short* obs_pt;
  const double* points;            // n_points x 3
  const TileHdr* tile_hdr;         // n_tiles
  const int* tile_frame;
  const int* tile_cam;
  const int* tile_off;             // n_tiles + 1
  const int* frame_tile_off;       // n_frames + 1
  const int* frame_cam_tile;       // n_frames x n_cams -> tile or -1
  const int* cam_model;            // per camera
  const int* cam_flags;            // kCamRotFree | kCamTransFree | kCamKFree
  const int* cam_col0;             // first shared column of the camera
  const int* col_cam;              // D: owning camera of a shared column (-1: IMU block)
  const int* col_local;            // D: column inside the camera's tile block
  double* poses[2];                // state double buffer
  double* cams[2];
  // Linearisation of the vision terms, double-buffered like the state: Gb[b] / tile_costb[b] belong to state buffer b.
  // Vision-only passes evaluate the Jacobian sweep AT THE TRIAL POINT inside k_trial (one projection sweep per LM
  // iteration instead of two); accepting the step flips `cur` and the linearisation is already there.
  double* Gb[2];                   // n_tiles x kGPack (packed upper triangle + side vector, vc_math.hpp)
  double* tile_costb[2];           // n_tiles   (Jacobian sweep: cost at the linearisation point)
  int fused;                       // 1: the trial point is evaluated by the Jacobian sweeps themselves (k_trial on vision-only passes,
                                   // k_reproj_jac / k_imu_jac in trial mode with the IMU), which leave the next linearisation in buffer 1-cur
  double* tile_trial;              // n_tiles x 2: trial cost, sum of squared residuals
  double* Y;                       // n_tiles x 96
  double* fr;                      // n_frames x 40
  double* fdiag;                   // n_frames x 6   clamped scaled diagonal (kept while reuse_diagonal)
  double* fscale2;                 // n_frames x 6   Jacobi scale^2 (fixed per solve)
  double* part;                    // n_chunks x part_stride
  double* part_total;              // part_stride: k_part_sum's fixed-order sum; only the entries behind S and g_red are used (those go straight to Sbuf)
  double* Sbuf;                    // D*D (S, no damping) + D (g_red) + D (H_ss diag) + D (g_s) + 2 (cost, spare)
  double* sdiag;                   // D
  double* sscale2;                 // D
  double* slam;                    // D
  double* delta_s;                 // D
  double* fpart;                   // n_frames x kNumScal: per-frame step terms (vision-only passes)
  // visual-inertial passes pre-reduce what the decision needs where it is produced, so that the single workgroup of k_final
  // adds a few hundred records instead of one per tile / block / frame:
  double* grp_part;                // n_chain_groups x kNumScal: step terms of the 8 frames of a bottom-level chain group (k_chain_back)
  double* wg_trial;                // (n_tiles + 3) / 4: trial cost of the 4 tiles of a k_reproj_jac workgroup
  double* wg_imu_trial;            // (n_frames - 1 + 7) / 8: trial cost of the 8 blocks of a k_imu_jac workgroup
  int n_chain_groups;
  double* scal;                    // kNumScal (frame sums) + kNumScal (shared-parameter terms)
  struct Ctrl* host_ctrl;          // page-locked host copies the deciding thread fills when host_progress is set: the control record once the solve is
  double* host_trace;              // over, every trace row (the first 64) as it is pushed -- the host needs neither a copy nor a synchronisation to read a solve's result
  unsigned long long* host_progress;   // page-locked host word, (decisions taken << 32) | Ctrl::done, stored by the deciding thread after
                                   // every decision: the host feeds passes against it without synchronising the stream (null: off)
  int* flags;                      // [0]: frame Cholesky failures, [1]: reduced Cholesky failure (per pass)
  // cross-stream hand-overs without event records on the main stream (visual-inertial pass, single process): single-workgroup
  // kernels publish the pass number when they are done, a one-wavefront kernel on the second stream waits for it
  long long* sync_flags;           // [0]: k_final, [1]: k_reduced, [2]: back-substitution done (k_reproj_jac(trial) has started), [3]: second stream's trial-point kernels done, [4]: workgroups of k_imu_jac(trial) that have delivered their cost share (a running count, never reset inside a solve), [5]: that count at the end of the last judged pass (k_final's own book-keeping), [6]: STICKY -- number of the first pass in which a wait ran into its bound (0: none), [7]: bottom level of the chain elimination complete, [11]: workgroups of k_imu_jac(trial) whose records have been performed (running count), [12]: that count at the end of the last judged pass
  long long sync_seq;              // this pass's number (0: no signalling)
  long long block_wait;            // k_imu_block(trial): one thread waits for sync_flags[2] >= block_wait before the kernel ends (0: no)
  long long final_wait;            // k_final waits for sync_flags[3] >= final_wait before it reads the second stream's sums (0: ordered by an event)
  long long sync_bound;            // polls of a flag wait before it gives up and marks its pass (spin_until_flag; ~0.5 us each)
  Ctrl* ctrl;
  long long* dbg;                  // 32 cycle-counter stamps (profiling aid)
  double* trace;                   // trace_cap x kTraceCols
  int part_stride;
  // ---- inertial terms (SwitchedFullImuCostFunction, one block per consecutive frame pair) ----------------
  int imu_on;                      // FLAGS_calibrate_imu && is_inertial_active_: IMU blocks are in the problem
  int rotation_only;               // optimize_rotation_only_ (residual switch, ceres-cost-functions.h:479-482)
  int weights_on;                  // UpdateImuWeights acts (vicalibrator.h:725)
  int n_imu;                       // IMU samples
  const double* imu_t;             // n_imu
  const double* imu_w;             // n_imu x 3 gyro
  const double* imu_a;             // n_imu x 3 accel
  const double* frame_time;        // n_frames
  double imu_avg_dt;               // InterpolationBufferT::average_dt_ of the sample stream (vc_imu.hpp: imu_average_dt)
  double* vel[2];                  // n_frames x 4, double-buffered like poses
  double* imus[2];                 // 16: g(2) b(6) sf(6) toff(1) pad
  int imu_param_col[15];           // shared column of g0 g1 b0..5 sf0..5 toff, -1 = constant
  double gyro_sigma, accel_sigma;
  double* wsqrtb[2];               // (n_frames-1) x 81  weight_sqrt_ of every IMU cost, double-buffered: the update of pass p
                                   // (k_imu_weights on a second stream) writes the buffer the Jacobian sweep of pass p is not reading
  // linearisation of the IMU blocks, double-buffered like the state (buffer b belongs to state buffer b)
  double* segb[2];                 // (n_frames-1) x kSegStride: weighted J^T J and J^T r of the block, compact (below)
  double* seg_costb[2];            // (n_frames-1)  imu_mult * rho at the linearisation point
  // delta form of the IMU blocks (vc_imu.hpp): the block's sample intervals, each an RK4 step from the identity state, appended in
  // order -- values + 13 partials (biases, scale factors, time offset).  The interval deltas themselves never reach memory
  long long* cready;               // n_frames: pass number in which the frame's step of the back-substitution was published (k_chain_back_levels)
  long long pass_id;               // this pass's number (monotonic over the calibrator's life)
  double* imu_grav;                // 2 x 16: gravity vector and its partials (vc_imu.hpp: imu_gravity_record) of state buffer b, written by k_imu_block, read by k_imu_jac
  double* imu_delta_blk;           // (n_frames - 1) x kBlockDeltaStride, written by k_imu_block, read by k_imu_jac; T = -1: empty range
  // ---- block-tridiagonal frame chain (9 x 9 blocks: pose 6 + velocity 3), cyclic reduction ----------------
  double* cW;                      // n_frames x 9 x ldx, one image per frame: columns 0..D-1 W -> Y = L^-1 W, column D: g -> z, then
                                   // from column ldw three 9 x 9 blocks: C (coupling to the group's left separator) -> X_s = L^-1 C,
                                   // A (diagonal block) -> L, B (coupling to the next active frame, rows = this frame) -> X_n = L^-1 B
  double* cdelta;                  // n_frames x 9
  double* ct0;                     // n_frames x 9: z + Y delta_s of every frame (k_chain_back's top-level launch)
  double* cg;                      // n_frames x 9  gradient
  double* clam;                    // n_frames x 9
  double* cdiag;                   // n_frames x 9
  double* cscale2;                 // n_frames x 9
  int ldw, ldx;                    // ldw: padded width of the border (D + 1 columns); ldx = ldw + 32: row stride of a frame's image
  // side images of the partitioned chain elimination: the update a group sends to its RIGHT separator (a frame of the
  // neighbouring group), by level parity; entry = frame / stride of the level that reads it
  double* rX[2];                   // (n_frames / 8 + 2) x 9 x ldx
  // frame sharding of the IMU chain: the first frame of every rank but rank 0 is a *separator* -- its 9 unknowns live in
  // the reduced system (columns sep_col0..+8) instead of the chain, so the interior chains of the ranks are independent.
  // pin_first: local frame 0 is this rank's separator; pin_last: local frame n_frames-1 is a copy ("ghost") of the next
  // rank's separator (columns sep_col1..+8), kept here because the IMU block that ends in it belongs to this rank.
  int pin_first, pin_last, sep_col0, sep_col1;
  // Early Gram (visual-inertial passes): sum [Y | z]^T [Y | z] of every frame below the chain's top level is formed by extra workgroups of
  // the top level's launch (beside its one group); the top level's own frames (index = 0 mod gram_top_stride, at most 7) are added by
  // k_reduced itself from their images (single process, D <= kEarlyTopD) or summed by a one-workgroup launch into partial record n_chunks
  // (n_part = n_chunks + 1).  0: k_chain_gram is a launch of its own and covers every frame.
  // (Tried: the partial sums as extra workgroups of k_reduced's launch, delivered with device-coherent stores and a count the first
  //  workgroup waits for -- 36 us against 7.4 + 23 for the two launches: coherent stores, the count and the coherent loads behind it cost
  //  more than a kernel boundary)
  int gram_top_stride;
  int fold_l0;                     // 1: k_chain_init's work rides in the bottom level's launch (k_chain_l0; chunk = group of 8 frames)
  int back_path;                   // 1: the whole back-substitution in one launch, upper levels recomputed per bottom group (k_chain_back_path)
  int rank, world;                 // frame sharding: this process's rank, number of ranks
  // Merged decision (vision-only, single process): the accept/reject decision on pass k's trial point is taken at the head
  // of pass k+1's k_frame_schur -- by every workgroup, redundantly and identically -- instead of a k_final launch per pass.
  // Control records alternate between two buffers: `ctrl` is the record of the current pass, `ctrl_prev` the previous one.
  int merged, par;                 // par = pass parity (selects the numeric-failure flag pair)
  int pre_backsub;                 // 1: k_backsub computes the frames' trial poses before k_trial (more tiles than resident waves)
  int shard_src;                   // merged decision reads the gathered per-rank scalars (`gath`) instead of k_trial's workgroup sums
  const Ctrl* ctrl_prev;
  double* wgpart;                  // k_trial: per-workgroup sums of the step scalars [n_workgroups][kNumScal]
  double* gath;                    // world x kNumScal: every rank's step scalars (one all-reduce(SUM) of disjoint slots = all-gather)
  double* sep_strip;               // 2 x 9 x ldw: rows of the reduced system contributed directly by the pinned frames
  // (round 6) 1: k_reduced ends with the step and the trial IMU parameters stored; the trial cameras and the shared parameters' terms of the
  // step scalars (vc_reduced_tail.hpp) are formed by one extra workgroup of the back-substitution's launch (k_chain_back_path).  Set per pass
  // by enqueue_pass (chain_back_is_path)
  int tail_deferred;
  // (round 6) the camera blocks, the IMU-parameter block and the chunk costs of the reduced system, formed ahead of k_reduced by side jobs of
  // the chain's upper-level launches (vc_shared_blocks.hpp): hadd has Sbuf's layout; hadd_early = 1: this pass's k_reduced starts from Sbuf + hadd
  double* hadd;
  int hadd_early;
  // (round 6) flag hand-overs, narrow single-process systems: the fixed-order sum of the chunk records (k_part_sum's work) rides in the top level's
  // launch -- extra workgroups behind the Gram chunks that wait for the chunks' ready words (part_ready[chunk] = pass number, published behind
  // device-coherent stores of the record) instead of for a kernel boundary.  Same slices, same order: Sbuf is identical to k_part_sum's to the bit.
  long long* part_ready;           // n_chunks + 1
  int part_ride;
};

// launchers (vc_kernels.hip); all asynchronous on `s`
void launch_reproj_jac(const DevView& v, hipStream_t s, int trial = 0);   // trial: sweep the trial state into buffer 1-cur (its cost = the trial cost)
void launch_part_sum(const DevView& v, hipStream_t s);         // fixed-order sum of the chunk partials
void launch_frame_schur(const DevView& v, hipStream_t s);      // frame elimination + per-chunk partial Schur sums
// mode 0: packed reduced system (Sbuf) + damped solve + trial shared parameters; 1: Sbuf only; 2: solve only
void launch_reduced(const DevView& v, int mode, hipStream_t s);
bool reduced_fits(const DevView& v);              // the reduced system fits k_reduced's LDS (D <= 179 on gfx950)
void launch_trial(const DevView& v, hipStream_t s);            // back-substitution + manifold update + trial residual sweep
void launch_final(const DevView& v, int mode, hipStream_t s);
int chain_forward_launches(const DevView& v);      // launches of the chain's forward elimination (levels + top)
int chain_top_stride(int n_frames);                // stride of the frames the chain's top level eliminates
bool chain_fold_supported(int n_frames, int D, int n_cams);      // k_chain_l0 can serve this problem (vc_imu_kernels.hip)
bool chain_hadd_early(const DevView& v);           // the forward elimination of this problem has the two launches above the bottom level that carry the side jobs of DevView::hadd
bool chain_back_is_path(const DevView& v);         // the back-substitution of this problem is one launch of k_chain_back_path with >= 256 threads per workgroup (it can carry the reduced solve's tail)
// a segment of a packed upload: `bytes` (a multiple of 4) from offset src_off of the staging image to dst; src_off = ~0: zero-fill
struct UnpackSeg { unsigned long long dst, src_off, bytes; };
void launch_unpack(const UnpackSeg* segs, int n, const void* image, size_t total_bytes, hipStream_t s);
void launch_set_ctrl(Ctrl* d, const Ctrl& c, hipStream_t s);      // d[0] <- c, d[1] <- 0
void launch_wait_flag(const DevView& v, int idx, long long seq, hipStream_t s);      // returns when sync_flags[idx] >= seq
void launch_signal_flag(const DevView& v, int idx, hipStream_t s);                    // sync_flags[idx] <- sync_seq once everything before it in the stream is done
void launch_final_merged(const DevView& v, hipStream_t s);   // merged mode, batch end: judges the last pass (ctrl = next record, ctrl_prev = last pass's)  // mode 0: reduce + decide, 1: reduce only, 2: decide only
void launch_reproj_res(const DevView& v, int state, double mult, hipStream_t s);   // residual sweep; state 0/1 buffer, 2 accepted, 3 trial (mult from Ctrl)
void launch_reset_state(const DevView& v, const double* pose0, const double* cam0, const double* vel0, const double* imu0, hipStream_t s);
void launch_sum_tile_cost(const DevView& v, double* out_cost_sq /*2*/, hipStream_t s);
void launch_cam_sq(const DevView& v, double* out /*n_cams x 2: sum sq, count*/, hipStream_t s);
void launch_outlier_mask(const DevView& v, int state, const double* thresh /*device, n_cams*/, unsigned char* mask, hipStream_t s);

// inertial path (vc_imu_kernels.hip)
int chain_group_size_upper();    // ... above the bottom level (VICALIB_AMD_CHAIN_M_UPPER)
int chain_group_size();          // frames per group of the partitioned chain elimination (test hook: VICALIB_AMD_CHAIN_M)
void launch_imu_delta(const DevView& v, hipStream_t s, int trial = 0);         // block deltas under the IMU parameters of the accepted (0) / trial (1) state (k_imu_block)
void launch_imu_jac(const DevView& v, int wr, hipStream_t s, int trial = 0);   // wr: weight buffer to read; trial as for launch_reproj_jac; needs launch_imu_delta
void launch_imu_weights(const DevView& v, int wr, hipStream_t s);          // reads wsqrtb[wr], writes wsqrtb[1 - wr];                   // weight_sqrt_ from the accepted state
void launch_chain_init(const DevView& v, hipStream_t s);                    // frame images from the tile Gram records and the IMU blocks
void launch_chain_fwd(const DevView& v, hipStream_t s);                     // forward elimination, one launch per level
void launch_chain_gram(const DevView& v, hipStream_t s);                    // sum of [Y | z]^T [Y | z] per chunk
void launch_chain_gram_top(const DevView& v, hipStream_t s);                // early Gram: the top level's frames as one more partial record
void launch_chain_solve_b(const DevView& v, hipStream_t s);                 // back-substitution + trial frame state

}  // namespace vc
