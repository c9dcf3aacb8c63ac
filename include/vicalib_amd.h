/* vicalib_amd.h -- C ABI of the MI355X-native calibration solver.
 *
 * Drop-in boundary for the optimisation core of arpg/vicalib: every entry point replaces one
 * public member of visual_inertial_calibration::ViCalibrator (include/vicalib/vicalibrator.h:119-544),
 * the header-only class that VicalibTask owns by value (vicalib-task.h:120) and drives from
 * vicalib-task.cc / vicalib-engine.cc.  Plain pointers and sizes only; the library copies in and
 * copies out, the handle is opaque, and nothing aborts: every call returns a status
 * (the reference CHECK()s / LOG(FATAL)s instead, vicalibrator.h:254, :377, :396, :456).
 *
 * Conventions (same as the reference's parameter blocks):
 *   SE3  = 7 doubles [qx qy qz qw tx ty tz]   (Sophus::SE3d::data(), vicalibrator.h:460, :604)
 *   T_wk = pose of the rig ("k") in the world; T_ck maps rig coordinates into camera c
 *   intrinsics = [fu fv u0 v0 distortion...]  (calibu parameter order, vicalib-engine.cc:207-257)
 *
 * The solver needs a HIP device (gfx950).  vc_create() fails with VC_ERR_NO_DEVICE on a machine
 * without one; there is no CPU fallback.
 */
#ifndef VICALIB_AMD_H_
#define VICALIB_AMD_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vc_calibrator vc_calibrator;

enum {
  VC_OK = 0,
  VC_ERR_NO_DEVICE = -1,      /* no HIP device / HIP runtime error */
  VC_ERR_BAD_ARG = -2,        /* index out of range, null pointer, unsupported model */
  VC_ERR_RUNNING = -3,        /* setter called while the solver runs (reference: CHECK(!is_running_)) */
  VC_ERR_TIME_ORDER = -4,     /* IMU timestamps not strictly increasing (vicalibrator.h:373-378) */
  VC_ERR_TOO_MANY_POINTS = -5,/* more than 32768 distinct target points */
  VC_ERR_NUMERIC = -6,        /* factorisation failed repeatedly */
  VC_ERR_UNSUPPORTED = -7,    /* feature of the reference not available in this build */
  VC_ERR_NO_CONVERGENCE = -8  /* one stage's problem ended NO_CONVERGENCE 64 times in a row (the reference would keep
                                 cranking, vicalibrator.h:952); state and multiplicities are left as they were */
};

/* -models strings of vicalib-engine.cc:203-253, in this order */
enum { VC_MODEL_FOV = 0, VC_MODEL_POLY2 = 1, VC_MODEL_POLY3 = 2, VC_MODEL_KB4 = 3, VC_MODEL_LINEAR = 4, VC_MODEL_RATIONAL6 = 5 };

/* ViCalibrator() vicalibrator.h:124-155 + Clear() :232-249.  device = HIP ordinal. */
int vc_create(vc_calibrator** out, int device);
void vc_destroy(vc_calibrator* h);
int vc_clear(vc_calibrator* h);                                   /* Clear() :232 */

/* AddCamera(cam, T_ck) :332-342 -> camera id (>= 0) or error */
int vc_add_camera(vc_calibrator* h, int model, const double* params, int nparams, int width, int height,
                  const double T_ck[7]);
int vc_fix_camera_intrinsics(vc_calibrator* h, int should_fix);   /* FixCameraIntrinsics :346 */
/* AddFrame(T_wk, time) :355-367 -> frame id */
int vc_add_frame(vc_calibrator* h, const double T_wk[7], double time);
/* GetFrame(id)->t_wp_ = ... (vicalib-task.cc:347-348) */
int vc_set_frame_pose(vc_calibrator* h, int frame, const double T_wk[7]);
/* Pose initialisation of the frames from their detections: replaces calibu::PosePnPRansac + the pose write at
   vicalib-task.cc:335-348 (T_wk = T_cw^-1 * T_ck, camera 0 if it tracked the grid, otherwise the last camera that
   did).  Deterministic planar homography + LM refinement on the current intrinsics; host code, runs once. */
int vc_init_frame_poses_pnp(vc_calibrator* h, int* n_initialised);
/* The same for one view, no handle: T_cw of a camera of `model`/`params` seeing n >= 4 corners of the planar grid. */
int vc_pnp_planar(int model, const double* params, int nparams, int n, const double* p_w /* n x 3 */,
                  const double* p_c /* n x 2 */, double T_cw[7], double* rms_px);
/* The robust branch of PosePnPRansac (robust_3pt_its > 0; the reference passes 0, 0 at vicalib-task.cc:323-325, which is
   vc_pnp_planar): `iterations` minimal 4-corner samples, consensus at tol_px pixels of reprojection error on the full camera
   model, refit on the consensus set.  inlier (n flags) and n_inliers are optional.  vc_set_pnp_ransac makes
   vc_init_frame_poses_pnp use it (iterations = 0 restores the reference's call). */
int vc_pnp_planar_ransac(int model, const double* params, int nparams, int n, const double* p_w, const double* p_c,
                         int iterations, double tol_px, double T_cw[7], double* rms_px, int* n_inliers, char* inlier);
int vc_set_pnp_ransac(vc_calibrator* h, int iterations, double tol_px);
/* AddObservation(frame, cam, p_w, p_c, time) :385-468, in bulk: n corners of one (frame, camera) */
int vc_add_observations(vc_calibrator* h, int frame, int camera, int n, const double* p_w /* n x 3 */,
                        const double* p_c /* n x 2 */);
/* The same over many (frame, camera) groups in one call, target points by index: group t holds corners
 * [tile_off[t], tile_off[t+1]) of point_id / p_c; points is the caller's table of target points (n_points x 3). */
int vc_add_observation_tiles(vc_calibrator* h, int n_tiles, const int* tile_frame, const int* tile_cam,
                             const long long* tile_off /* n_tiles + 1 */, const double* points, int n_points,
                             const int* point_id, const double* p_c /* x 2 */);
/* AddImuMeasurements(gyro, accel, time) :370-380, in bulk */
int vc_add_imu(vc_calibrator* h, int n, const double* gyro /* n x 3 */, const double* accel /* n x 3 */,
               const double* time /* n */);

int vc_set_sigmas(vc_calibrator* h, double gyro_sigma, double accel_sigma);       /* SetSigmas :290 */
int vc_set_biases(vc_calibrator* h, const double biases[6]);                       /* SetBiases :301 */
int vc_set_scale_factor(vc_calibrator* h, const double scale[6]);                  /* SetScaleFactor :308 */
int vc_set_time_offset(vc_calibrator* h, double offset);                           /* SetTimeOffset :296 */
int vc_set_function_tolerance(vc_calibrator* h, double tol);                       /* SetFunctionTolerance :277 */
/* SetOptimizationFlags(bias_active, inertial_active, rotation_only, optimize_imu_time_offset) :252-260 */
int vc_set_optimization_flags(vc_calibrator* h, int bias_active, int inertial_active, int rotation_only,
                              int optimize_time_offset);
/* gflags read inside the calibrator: FLAGS_max_iters (:142), FLAGS_calibrate_imu (:214, :651, :977),
 * FLAGS_remove_outliers / FLAGS_outlier_threshold (:870, :995, :1024) */
int vc_set_max_iters(vc_calibrator* h, int max_iters);
/* ceres::Solver::Options::gradient_tolerance / parameter_tolerance, which the reference leaves at the Ceres defaults
 * (1e-10 / 1e-8, vicalibrator.h:141-151); settable here so that tests can converge a solve to rounding level */
int vc_set_tolerances(vc_calibrator* h, double gradient_tolerance, double parameter_tolerance);
int vc_set_calibrate_imu(vc_calibrator* h, int calibrate_imu);
int vc_set_remove_outliers(vc_calibrator* h, int remove_outliers, double outlier_threshold);

/* Start() :263 / IsRunning() :314 / Stop() :317; vc_solve = Start() + join (blocking SolveThread :919) */
int vc_solve(vc_calibrator* h);
int vc_start(vc_calibrator* h);
/* is_finished_ is sticky until Clear() as in the reference (:246, :922): Start()/Solve() on a finished calibrator return at
 * once.  vc_resume clears the flag (engine-level, no reference counterpart) so that the next Solve() runs SetupProblem +
 * the solve loop again from the current state (re-adding every block once more, as any pass of the outer loop does). */
int vc_resume(vc_calibrator* h);
/* Bench / test hook: Solve() runs the first n stages of the schedule (:977-1000), sets up stage n + 1 (constancy flags,
 * block multiplicities, gravity) and returns without running it; n < 0 (default) = the whole schedule.  With the state
 * left there, vc_prepare + vc_run_iterations time LM iterations of exactly that stage. */
int vc_set_stage_limit(vc_calibrator* h, int n);
int vc_is_running(vc_calibrator* h);
int vc_stop(vc_calibrator* h);

/* readers (fields of CalibrationStats, calibration-stats.h:34-42) */
int vc_num_frames(vc_calibrator* h);                                               /* NumFrames :471 */
int vc_num_cameras(vc_calibrator* h);                                              /* NumCameras :484 */
int vc_get_camera(vc_calibrator* h, int camera, double* params, int* nparams, double T_ck[7]);   /* GetCamera :492 */
int vc_get_frame(vc_calibrator* h, int frame, double T_wk[7], double v_w[3], double* time);      /* GetFrame :477 */
int vc_get_biases(vc_calibrator* h, double biases[6]);                             /* GetBiases :286 */
int vc_get_scale_factor(vc_calibrator* h, double scale[6]);                        /* GetScaleFactor :306 */
int vc_get_gravity(vc_calibrator* h, double g_dir[2]);                             /* imu_.g_ :1020 */
double vc_time_offset(vc_calibrator* h);                                           /* time_offset :474 */
double vc_mean_squared_error(vc_calibrator* h);                                    /* MeanSquaredError :506 */
int vc_get_camera_proj_rmse(vc_calibrator* h, double* rmse /* n_cameras */);       /* GetCameraProjRMSE :160 */
unsigned vc_get_num_iterations(vc_calibrator* h);                                  /* GetNumIterations :283 */
/* imu_buffer() :487: the stored measurements in time order (any pointer may be NULL); returns the number copied */
int vc_num_imu_measurements(vc_calibrator* h);
int vc_get_imu_measurements(vc_calibrator* h, double* gyro, double* accel, double* time, int max_n);
/* GetIntegrationPoses(id) :508-533: the poses the IMU integration passes through between frame id and id + 1 -- the start pose,
 * then one per measurement of the range; rows of 11 doubles [q(4) t(3) v_w(3) time].  Returns the count (0 unless the inertial
 * terms are fully active, :510), which may exceed max_poses. */
int vc_get_integration_poses(vc_calibrator* h, int id, double* poses, int max_poses);
/* PrintResults() :536-544 into buf: per camera its parameters and T_ck as a 4 x 4 matrix; returns the length of the text.
 * len = 0 (buf may be NULL): only the length is returned -- the text needs a buffer of that + 1 bytes; a buffer too small is VC_ERR_BAD_ARG */
int vc_print_results(vc_calibrator* h, char* buf, int len);
/* WriteCameraModels(filename) :208-229 (calibu rig XML) */
int vc_write_camera_models(vc_calibrator* h, const char* filename);

/* GetSolutionCovariance(problem) :802-857 (compiled in the reference only with COMPUTE_VICALIB_COVARIANCE, :1004-1013):
 * covariance of the blocks of covariance_params_ (:561, :567, :594) at the current state -- per camera q_ck (4, lifted
 * through the SO3 local parameterisation), p_ck (3) and, unless the intrinsics are fixed, the model parameters; frames
 * and IMU states marginalised; constant blocks are zero.  n x n row-major, n = vc_solution_covariance_dim().
 * The names string is the reference's column header ("c[0].q_ck:(4) c[0].p_ck:(3) c[0].params:(5) ..."). */
int vc_solution_covariance_dim(vc_calibrator* h);
int vc_get_solution_covariance(vc_calibrator* h, double* cov, int max_n, int* n);
int vc_get_solution_covariance_names(vc_calibrator* h, char* buf, int len);

/* ---- engine-level entry points (no counterpart in the reference: it has no GPU, no sharding) ---- */
/* State the reference keeps inside the class with no setter (imu_.g_, VicalibFrame::v_w_): start values for tests and for a
 * caller that resumes from a stored calibration.  vc_set_gravity also marks gravity as initialised (:927-949 is skipped). */
int vc_set_gravity(vc_calibrator* h, const double g_dir[2]);
int vc_set_frame_velocities(vc_calibrator* h, const double* v_w /* n x 3 */, int n);
/* Per-iteration record of the trust-region loop = the columns of the reference's log line (:698-707).
 * rows of 10 doubles: iteration, cost, cost_change, gradient_max_norm, gradient_norm, step_norm,
 * relative_decrease, trust_region_radius, accepted, stage */
int vc_trace_len(vc_calibrator* h);
int vc_get_trace(vc_calibrator* h, double* rows, int max_rows);
/* Frame sharding across processes (one process per GPU): this handle holds frames
 * [first_global_frame, first_global_frame + n_local) of a problem that world_size ranks solve together.
 * allreduce_sum / allreduce_max are called on every LM iteration with a DEVICE pointer and must return
 * only when the reduction is complete on the calibrator's stream (see vc_get_stream). */
typedef int (*vc_allreduce_fn)(void* ctx, double* device_buf, int count, int op /*0 sum, 1 max*/);
int vc_set_shard(vc_calibrator* h, int rank, int world_size, vc_allreduce_fn fn, void* ctx);
void* vc_get_stream(vc_calibrator* h);    /* hipStream_t */
/* The same sharding with the library's own RCCL communicator (librccl bound at run time): the per-iteration all-reduces
 * are enqueued directly on the calibrator's stream, no callback.  Rank 0 creates the id (ncclGetUniqueId, 128 bytes),
 * the host distributes it by whatever means it has, every rank calls vc_set_shard_rccl (collective: ncclCommInitRank). */
int vc_rccl_unique_id(void* out128);
int vc_set_shard_rccl(vc_calibrator* h, int rank, int world_size, const void* unique_id128);
/* One RCCL communicator for several calibrators of a process (a launcher that solves more than once: one ncclCommInitRank and
 * one ncclCommDestroy per process instead of one pair per calibrator).  vc_shard_comm_create is collective like
 * vc_set_shard_rccl; vc_set_shard_comm lends the communicator to a calibrator on the same device (rank / world size are the
 * communicator's); the caller destroys it after the calibrators that used it. */
typedef struct vc_shard_comm vc_shard_comm;
int vc_shard_comm_create(int device, int rank, int world_size, const void* unique_id128, vc_shard_comm** out);
int vc_set_shard_comm(vc_calibrator* h, vc_shard_comm* comm);
void vc_shard_comm_destroy(vc_shard_comm* comm);
long long vc_allreduce_calls(vc_calibrator* h);    /* all-reduces issued through the library's own communicator */
/* The sharding a calibrator runs with: rank / world_size as set by vc_set_shard*, and -- for the library's own communicator -- what
 * RCCL itself reports (ncclCommCount, ncclCommUserRank; -1: no RCCL communicator attached): a launcher can check that RCCL saw the
 * ranks it was started with.  Any of the pointers may be NULL.  (The reference is single-process: vicalibrator.h:263-274.) */
int vc_shard_info(vc_calibrator* h, int* rank, int* world_size, int* rccl_ranks, int* rccl_rank);
/* Which forms of the visual-inertial pass the uploaded problem runs (after vc_prepare / a solve; a parity hook: the tests assert that the
 * kernels they mean to check are the ones that ran): out4 = { chain assembly folded into the bottom level (k_chain_l0), back-substitution as
 * one launch (k_chain_back_path), Gram sums in the top level's launch, top-level frames as a partial record of their own, the reduced
 * solve's tail in the back-substitution's launch, the shared parameters' blocks formed ahead of the reduced solve }. */
int vc_pass_paths(vc_calibrator* h, int* out6);
/* Text behind the last failing status of vc_set_shard_rccl on this thread (which library call failed, RCCL's error string and
 * last-error text): what a launcher prints before it falls back to another transport.  Empty if nothing failed. */
const char* vc_last_error(void);
/* Upload the problem and linearise once at the current state (stage flags as set): fills the device
 * normal equations.  Used by the parity tests and the benchmark. */
int vc_prepare(vc_calibrator* h);
/* Copies of device results after vc_prepare / vc_linearize (any pointer may be NULL):
 *   cost, per-frame H_pp (n x 36), g_p (n x 6), reduced S (D x D, undamped Schur complement), g_red (D),
 *   H_ss diagonal (D), g_s (D) */
int vc_linearize(vc_calibrator* h, double* cost, double* Hpp, double* gp, double* S, double* g_red,
                 double* hss_diag, double* g_s);
int vc_shared_dim(vc_calibrator* h);
/* Runs exactly `iters` LM iterations of the real solver (complete solves back to back from the uploaded
 * initial state, the last one cut short); returns the number of iterations run (>= 0) or an error. */
int vc_run_iterations(vc_calibrator* h, int iters, int* jac_sweeps, int* res_sweeps);
/* Per-tile reprojection residual sweep on the accepted state: cost (1/2 sum rho) and sum of squares */
int vc_evaluate(vc_calibrator* h, double* cost, double* sum_sq);
/* Times the dominant kernels with HIP events on the calibrator's stream: average ms per launch over reps */
int vc_time_kernels(vc_calibrator* h, int reps, double* jac_ms, double* res_ms);
/* In-loop kernel timing: with `on`, every launch group of every LM pass of the following solves is bracketed by HIP events
 * on the calibrator's stream (the real loop, decisions live -- not held back-to-back launches); vc_get_kernel_timing
 * returns, per group that ran, its name (';'-joined into names), the summed duration and the launch count. */
int vc_set_kernel_timing(vc_calibrator* h, int on);
/* Cross-stream hand-overs of the visual-inertial pass go through device flags.  A wait that runs into its bound is
 * never a silent change of results: the device withholds that pass's decision, the library reports it on stderr, resumes the
 * solve with event hand-overs (same iterates) and keeps them for this calibrator.  Returns how often that has happened. */
int vc_sync_timeouts(const vc_calibrator* h);
int vc_get_kernel_timing(vc_calibrator* h, char* names, int names_len, double* total_ms, long long* count, int max_entries);
/* Average ms per launch of each stage of one LM pass (Jacobian sweep, frame elimination, Schur partials,
 * reduced solve, trial sweep, decision), `reps` back-to-back launches each */
int vc_time_stages(vc_calibrator* h, int reps, double out[6]);
/* After vc_linearize with inertial terms active: weighted J^T J (33 x 33), J^T r (33), cost of each IMU block,
 * columns [frame j: pose 6, vel 3 | frame j-1: pose 6, vel 3 | g 2, b 6, sf 6, time offset 1] */
int vc_get_imu_blocks(vc_calibrator* h, double* H, double* g, double* cost);
/* Current weight_sqrt_ factors W (9 x 9 per IMU block, row-major) with W W^T = (J Sigma J^T)^-1, after vc_linearize
 * with the weight update active (UpdateImuWeights, vicalibrator.h:723-799). */
int vc_get_imu_weights(vc_calibrator* h, double* W);
int vc_get_debug_stamps(vc_calibrator* h, long long out[32]);   /* shader-clock stamps of the last k_reduced (profiling aid) */
long long vc_num_observations(vc_calibrator* h);
int vc_num_tiles(vc_calibrator* h);

/* ---- image front-end, first slice -----------------------------------------------------------------------
 * Replaces, per image stream, the pair  calibu::ImageProcessing image_processing_[i](width, height)  +  calibu::ConicFinder
 * conic_finder_[i]  that VicalibTask owns (vicalib-task.h, constructed at vicalib-task.cc:115) and the two calls
 *     image_processing_[ii].Process(img->data(), img->Width(), img->Height(), img->Width());     vicalib-task.cc:264-267
 *     conic_finder_[ii].Find(image_processing_[ii]);                                             vicalib-task.cc:268
 * of AddImageMeasurements; the result is what the reference reads as conics[i].center (:296).  Parameters and defaults are the
 * ones VicalibTask sets (:116-122).  The image is an 8-bit greyscale HOST buffer (what HAL hands over); centres come back in
 * pixel coordinates (x, y), ordered by the smallest pixel index of their dot.  Grid matching (FindTarget, :274) is not part of
 * this slice.  No CPU fallback: vc_detector_create fails with VC_ERR_NO_DEVICE without a HIP device. */
typedef struct vc_detector vc_detector;
int vc_detector_create(int device, int width, int height, vc_detector** out);
void vc_detector_destroy(vc_detector* d);
int vc_detector_set_params(vc_detector* d, int black_on_white, double at_threshold, double at_window_ratio, double conic_min_area,
                           double conic_min_density, double conic_min_aspect);
/* centres: 2 x max_conics doubles; *n_found is the number of dots found (may exceed max_conics: the first max_conics are written).
 * A detector handle is single-threaded: one image at a time (its staging buffers and stream belong to the call in progress); use one
 * handle per thread. */
int vc_detector_find(vc_detector* d, const unsigned char* image, int pitch, double* centres, int max_conics, int* n_found);
/* The same with the rest of what calibu::Conic carries (vicalib-task.cc:270-277 hands the conics to TargetGridDot::FindTarget):
 * conics (nullable): 9 doubles per dot, the ellipse as a symmetric 3 x 3 matrix C with x^T C x = 0 on its edge, image coordinates
 * (pixel centres at integers), unit Frobenius norm, C[0][0] > 0 (calibu::Conic::C; Dual is its inverse, center what `centres` holds);
 * boxes (nullable): 4 ints per dot, the dot's bounding box x0, y0, x1, y1 inclusive (calibu::Conic::bbox). */
int vc_detector_find_conics(vc_detector* d, const unsigned char* image, int pitch, double* centres, double* conics, int* boxes, int max_conics,
                            int* n_found);
/* Second half of the front-end (vicalib-task.cc:274-277, calibu::TargetGridDot::FindTarget; vicalib-engine.cc:459-461,
 * calibu::MakePattern): which dot of the target is every detected conic?  Host code, no device needed (a few hundred dots per image).
 * vc_target_make_pattern: the large (1) / small (0) pattern of a rows x cols target from a seed (own generator: Calibu's is not in the
 * reference tree -- a printed Calibu target needs its own pattern).  vc_target_find: centres (2 per dot) and image ellipses (9 per dot,
 * as vc_detector_find_conics returns them) of one image -> dot_index (row * cols + col, or -1) per conic, the reference's
 * `ellipse_target_map`; *n_matched = 0 when no unambiguous placement exists (the reference then skips the frame).  Calibu's source
 * being absent, parity with FindTarget is unpinned; tests hold it to rendered views (tests/test_grid_cpu.py). */
int vc_target_make_pattern(int rows, int cols, unsigned seed, int* pattern);
int vc_target_find(const double* centres, const double* conics, int n, const int* pattern, int rows, int cols, int* dot_index, int* n_matched);

#ifdef __cplusplus
}
#endif
#endif /* VICALIB_AMD_H_ */
