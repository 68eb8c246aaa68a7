/*
 * ba_amd.h -- C ABI of the MI355X-native sparse bundle-adjustment solver (libba_amd.so).
 *
 * This is the drop-in boundary for the hot path of AubreyC/visual-odometry's
 * src/bundle_adjustment.py.  The reference has no FFI: the path is one Python method,
 *     BundleAdjustment.optimize(image_features_list, camera_pose_initial, camera_model,
 *                               camera_poses_guess, points3d_guess)        (bundle_adjustment.py:156-163)
 * which packs a parameter vector (:178-194), builds a sparsity mask (:196 -> :90-154) and calls
 * scipy.optimize.least_squares(residual_function, ...) (:201-219).  The entry points below are what a
 * ctypes binding of that method calls instead (visual-odometry_amd/src/_native.py; INTEGRATION.md
 * shows the stub a maintainer of the reference would add).
 *
 * Conventions (the reference's): pose = [tx,ty,tz, qw,qx,qy,qz] (position of the camera centre in
 * the world frame, unit quaternion rotating camera->world, geometry.py:192-213); camera-frame point
 * Xc = R(q)^T (P - t) (camera_pose.py:85-86); pixel = (fx x/z + cx, fy y/z + cy), optionally times the
 * radial factor 1 + k1 r^2 + k2 r^4 + k3 r^6 (camera.py:134-165); residual = observed - projected
 * (bundle_adjustment.py:80-83); cost = 0.5 * sum residual^2 (scipy convention).
 *
 * All pointers are HOST pointers to C-contiguous arrays owned by the caller; the library owns every
 * device allocation behind the opaque handle.  Every call is synchronous.  A handle may be used by one
 * thread at a time; distinct handles are independent.  Every function that returns int returns a
 * ba_status (0 = ok); ba_last_error(handle) gives the text for the last non-zero status.
 * fp64 throughout; indices are int32.
 */
#ifndef BA_AMD_H
#define BA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BA_AMD_VERSION 102 /* 0.1.2: ba_problem_info writes 13 entries (BA_INFO_COUNT was 12), ba_debug_table_digests is new.
                              0.1.1: ba_report gained bad_pivots / guard_rejections / shortened_steps, BA_KERNEL_CLASSES
                              went from 6 to 8 (ba_get_kernel_times writes 8 entries), ba_comm_info is new.  A caller built
                              against an older version must be rebuilt (compare ba_version() with the header it was built with) */
#define BA_HISTORY 128     /* per-iteration history entries kept in ba_report: the first BA_HISTORY - 1 iterations and,
                              in the last entry, the most recent one (a solve may run longer; n_history stops at 128) */

typedef enum {
  BA_OK = 0,
  BA_ERR_INVALID = 1,   /* bad argument (shape, index out of range, non-finite input)  -> ValidationError */
  BA_ERR_NUMERICAL = 2, /* non-finite cost at the initial point, z == 0               -> ProcessingError */
  BA_ERR_DEVICE = 3,    /* HIP / RCCL runtime error                                   -> RuntimeError    */
  BA_ERR_NOMEM = 4,     /* host or device allocation failed                           -> RuntimeError    */
  BA_ERR_STATE = 5      /* call order (e.g. solve before set_problem)                 -> RuntimeError    */
} ba_status;

/* Termination reason, same vocabulary as scipy's OptimizeResult.status that the reference's
 * least_squares call produces (scipy/optimize/_lsq/common.py:705-717): */
typedef enum { BA_STOP_MAX_ITERS = 0, BA_STOP_GTOL = 1, BA_STOP_FTOL = 2, BA_STOP_XTOL = 3 } ba_stop;

typedef enum {
  BA_SOLVER_AUTO = 0,
  BA_SOLVER_PCG = 1,
  BA_SOLVER_DENSE_CHOLESKY = 2,
  BA_SOLVER_BANDED_CHOLESKY = 3, /* block-banded (+ arrow) Cholesky: cameras that only share landmarks with their
                                    neighbours along a trajectory, possibly closed into a loop */
  BA_SOLVER_CYCLIC_REDUCTION = 4 /* the same systems by block cyclic reduction (every other group of cameras eliminated at
                                    once, log2 depth): what AUTO takes for a long trajectory */
} ba_linear_solver;

/* Robust loss, applied to every scalar residual with the semantics of the `loss=` / `f_scale=`
 * arguments of scipy.optimize.least_squares (scipy/optimize/_lsq/least_squares.py:186-236): the hook
 * that the reference's call (bundle_adjustment.py:201-212) leaves at its default, 'linear'.
 * cost = 0.5 * f_scale^2 * sum rho((r / f_scale)^2).  The objective is scipy's; the linearisation weights
 * each residual row by sqrt(rho') (iteratively re-weighted least squares) instead of scipy's clamped
 * second-order correction, which stalls LM on outlier-dominated initial guesses. */
typedef enum {
  BA_LOSS_LINEAR = 0,  /* rho(z) = z                      (the reference's setting) */
  BA_LOSS_SOFT_L1 = 1, /* rho(z) = 2 (sqrt(1 + z) - 1)    */
  BA_LOSS_HUBER = 2,   /* rho(z) = z if z <= 1 else 2 sqrt(z) - 1 */
  BA_LOSS_CAUCHY = 3,  /* rho(z) = ln(1 + z)              */
  BA_LOSS_ARCTAN = 4   /* rho(z) = arctan(z)              */
} ba_loss;

/* Solver settings.  The reference hard-codes its settings as literals in the least_squares call
 * (bundle_adjustment.py:201-212: ftol = xtol = gtol = 1e-6, max_nfev = 2000); defaults here are
 * tighter (ftol = scipy's own default) so that the converged cost is at or below the reference's
 * (SURVEY.md section 7). */
typedef struct {
  int32_t max_iters;        /* LM iterations, accepted + rejected; default 2000 (the reference's max_nfev=2000) */
  double ftol;              /* stop when an accepted step lowers the cost by <= ftol * cost; 1e-8 */
  double xtol;              /* stop when |step| <= xtol * (xtol + |x|); 1e-12 */
  double gtol;              /* stop when |J^T r|_inf <= gtol; 1e-10 */
  double lambda0;           /* initial LM damping (Marquardt-scaled), 1e-4 */
  double lambda_min;        /* 1e-12 */
  double lambda_max;        /* 1e12 */
  double min_diag;          /* floor of the Marquardt diagonal, 1e-12 */
  double accept_ratio;      /* accept a step when actual/predicted decrease > this, 1e-4 */
  int32_t guard_cheirality; /* 1 (default): a landmark step that would carry the landmark through the principal plane of one
                               of its cameras is shortened (depths keep >= 1/10 of their value), and a whole step that still
                               moves more observations behind their camera is rejected; 0: neither (the reference has no
                               cheirality test, camera.py:101-165) */
  int32_t linear_solver;    /* ba_linear_solver, default BA_SOLVER_AUTO */
  double pcg_tol;           /* relative residual (M^-1 norm) at which block-Jacobi PCG stops, 3e-2 */
  int32_t pcg_max_iter;     /* 1000 */
  int32_t pcg_check_every;  /* PCG iterations between host polls of the device-side stop flag, 10 */
  int32_t lm_check_every;   /* LM iterations enqueued between host polls of the device-side LM state, 1 */
  int32_t device;           /* HIP device ordinal; -1 = current device */
  int32_t verbose;          /* 1 = print one line per LM iteration to stderr */
  int32_t profile_kernels;  /* 1 = bracket every kernel class with HIP events (ba_get_kernel_times) */
  int32_t loss;             /* ba_loss, default BA_LOSS_LINEAR */
  double f_scale;           /* soft margin between inlier and outlier residuals, pixels; default 1.0 */
} ba_options;

typedef struct {
  double cost0;        /* cost at the initial guess: 0.5 sum r^2, or the robust cost when opt.loss is set */
  double cost;         /* at the returned estimate */
  double grad_inf;     /* |J^T r|_inf at the last linearisation */
  double lambda;       /* final damping */
  double rms_reproj;   /* sqrt(2 cost / nobs), pixels */
  int32_t iterations;  /* linearise+solve rounds, accepted + rejected */
  int32_t accepted;
  int32_t linearizations; /* Jacobian evaluations (1 + accepted) */
  int32_t status;      /* ba_stop */
  int32_t pcg_iters;   /* total PCG iterations */
  int32_t nobs;
  int32_t bad_pivots;       /* non-positive pivots / non-SPD blocks met by the linear solves (each such step was rejected) */
  int32_t guard_rejections; /* steps refused because they put more observations behind their camera */
  int32_t shortened_steps;  /* landmark steps shortened so that the landmark stays in front of its cameras (summed over iterations) */
  int32_t reserved_;
  double solve_ms;     /* wall time inside the solve, device-synchronised */
  double setup_ms;     /* wall time of the last ba_set_problem (index build + upload) */
  int32_t n_history;
  double cost_history[BA_HISTORY];   /* cost after each iteration */
  double lambda_history[BA_HISTORY]; /* damping used by each iteration */
  int32_t pcg_history[BA_HISTORY];   /* PCG iterations of each iteration */
  int32_t accepted_history[BA_HISTORY];
} ba_report;

typedef struct ba_handle ba_handle;

/* Library / device queries (no reference counterpart; the reference is CPU-only). */
int ba_version(void);
int ba_device_count(void);

/* Fill `opt` with the defaults listed above.  Replaces the literal kwargs of
 * bundle_adjustment.py:201-212. */
void ba_default_options(ba_options* opt);

/* Create / destroy a solver bound to one GPU.  BundleAdjustment() is stateless
 * (bundle_adjustment.py:15-16); the handle holds the device buffers of one problem at a time. */
int ba_create(const ba_options* opt, ba_handle** out);
void ba_destroy(ba_handle* h);
const char* ba_last_error(const ba_handle* h);

/* Define the problem structure.  Replaces the packing of bundle_adjustment.py:178-194 and the mask
 * construction of bundle_adjustment_sparsity (:90-154): the (camera, landmark) index pairs ARE the
 * sparsity pattern.
 *   n_cams   C  images/poses including the fixed ones (reference: len(image_features_list))
 *   n_pts    L  landmarks (reference: len(points3d_guess.get_ids()))
 *   n_obs       observations, in any order; observation k is pixel (obs_uv[2k], obs_uv[2k+1]) of
 *               landmark INDEX obs_pt[k] seen in image obs_cam[k]
 *   intr[7]     fx, fy, cx, cy, k1, k2, k3   (PinHoleCamera.to_vector, camera.py:93-99)
 *   fixed[C]    non-zero = pose is held constant (reference: only image 0, :51-52,141); may be NULL
 *               for "camera 0 fixed"
 * The library sorts observations into landmark tracks and builds the block pattern of the reduced
 * camera system once here. */
int ba_set_problem(ba_handle* h, int32_t n_cams, int32_t n_pts, int32_t n_obs, const int32_t* obs_cam,
                   const int32_t* obs_pt, const double* obs_uv, const double* intr, const uint8_t* fixed);

/* A quadratic prior on some of the poses of the NEXT ba_set_problem (and of every later one, until replaced; n_prior =
 * 0 removes it): what a sliding-window driver keeps of the keyframes and landmarks it drops (SURVEY.md section 8 row
 * f2).  No counterpart in the reference: its keyframe list (visual_odometry.py:13-25,268-277) is never optimised.
 *   E_prior(x) = c0 + g^T d + 0.5 d^T H d,   d = [d_0 .. d_{n-1}],  d_a = [t_a - tref_a ; 2 vec(qref_a^-1 q_a) / w(qref_a^-1 q_a)]
 * i.e. d_a are the coordinates of pose cams[a] in the chart of the library's own retraction (t + dt, q (x) [1, dtheta/2])
 * around ref_poses[a]; H[(6n)^2] row-major, symmetric positive semi-definite; g[6n].  The energy is added to the cost
 * that ba_solve minimises and reports (ba_evaluate keeps returning the re-projection cost alone); its gradient is exact
 * and its Hessian the Gauss-Newton one (J = d d_a / d step = I + [a]x + a a^T on the rotation part, a = vec / w).  cams must be distinct, free
 * cameras of the problem (checked by ba_set_problem); n_prior <= BA_PRIOR_MAX_CAMS.  The camera pairs of the prior
 * join the block pattern of the reduced camera system.  Not available with ba_comm_init (several ranks) and not in
 * ba_solve_batched: both return BA_ERR_STATE while a prior is set. */
#define BA_PRIOR_MAX_CAMS 16
int ba_set_pose_prior(ba_handle* h, int32_t n_prior, const int32_t* cams, const double* ref_poses, const double* H,
                      const double* g, double c0);

/* Optimise.  poses[C*7] and points[L*3] are in/out: initial guess in, estimate out (quaternions come
 * back unit-length; fixed poses and unobserved landmarks come back untouched).  Replaces the
 * least_squares call of bundle_adjustment.py:201-219 plus the unpacking of :221-244.
 * Non-convergence is not an error (the reference ignores scipy's status): read report->status. */
int ba_solve(ba_handle* h, double* poses, double* points, ba_report* report);

/* The three steps of ba_solve separately, for callers that keep data resident on the GPU between
 * calls (bench.py times ba_solve_resident only): upload a guess; solve from the uploaded guess
 * (repeatable: each call restarts from it); download the estimate of the last solve. */
int ba_upload_guess(ba_handle* h, const double* poses, const double* points);
int ba_solve_resident(ba_handle* h, ba_report* report);
int ba_download_estimate(ba_handle* h, double* poses, double* points);

/* Batched local bundle adjustment (BASELINE.json configs[4]: thousands of independent windows of
 * ~10 keyframes; SURVEY.md section 8b "solve_batched(B problems, offsets)").  The reference would call
 * BundleAdjustment.optimize (bundle_adjustment.py:156-163) once per window from its keyframe list
 * (visual_odometry.py:268-277 keeps the keyframes; the call itself is missing there).  Problem b owns
 * cameras [cam_off[b], cam_off[b+1]), landmarks [lm_off[b], lm_off[b+1]) and observations
 * [obs_off[b], obs_off[b+1]); obs_cam / obs_pt are indices LOCAL to the observation's problem; the
 * first camera of every problem is held fixed; one camera model for all.  poses[cam_off[B]*7] and
 * points[lm_off[B]*3] are in/out.  All windows are solved together as one block-diagonal problem: the
 * streaming kernels run once over the whole batch, the reduced camera system is solved exactly (one dense
 * Cholesky per window), and every window (connected component of the camera graph) has its own LM damping
 * sequence, its own accept / reject decisions and its own ftol test: a window that rejects a step or has
 * converged does not hold the others back, and the batch ends when every window has (status 2) or at max_iters.
 * (Tracks longer than 32 observations fall back to one damping sequence for the batch.)  cost0 / cost (may be NULL) receive each
 * problem's 0.5 * sum r^2 before / after; report (may be NULL) describes the batch.  Leaves the
 * handle set to the concatenated problem. */
int ba_solve_batched(ba_handle* h, int32_t n_problems, const int32_t* cam_off, const int32_t* lm_off,
                     const int32_t* obs_off, const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv,
                     const double* intrinsics, double* poses, double* points, double* cost0, double* cost,
                     ba_report* report);

/* Residual vector and cost at (poses, points), residuals[2*n_obs] in the caller's observation
 * order ([u0, v0, u1, v1, ...]); residuals may be NULL.  Replaces
 * BundleAdjustment.residual_function (bundle_adjustment.py:20-87).  n_behind (may be NULL) receives
 * the number of observations with z <= 0.  Returns BA_ERR_NUMERICAL if some z == 0, as
 * PinHoleCamera.project raises ProcessingError there (camera.py:129-132).  The residuals are always the
 * plain re-projection errors; *cost is the robust cost when opt.loss is set (as scipy's result.cost).
 * Side effect: the point is staged in the handle's trial buffers, so the estimate of an earlier ba_solve_resident is
 * gone (ba_download_estimate then returns BA_ERR_STATE until the next solve); the same holds for
 * ba_jacobian_blocks, ba_debug_linearization and ba_triangulate. */
int ba_evaluate(ba_handle* h, const double* poses, const double* points, double* residuals, double* cost,
                int32_t* n_behind);

/* Residuals and cost WITHOUT any allocation.  No counterpart in the reference (which allocates freely: every call
 * of BundleAdjustment.residual_function builds new arrays, bundle_adjustment.py:41-87); this entry point is for
 * callers that cannot call hipMalloc: tight serving loops that evaluate the residual as a callback, harnesses that
 * forbid allocation.  (NOT for use under stream capture: it makes blocking copies on the default stream.)  It is ba_evaluate without a handle and without ba_set_problem:
 * the same residual arithmetic on the observations in the caller's order, camera 0 being the fixed pose
 * (its quaternion is normalised only if off by more than 1e-6, the others always, as in ba_evaluate), plain
 * squared loss.  Everything it needs on the device (staged copies of the inputs, the residuals, per-block partial
 * sums) lives in a static device array of about 6.9 MB that every GPU used by the library holds until the process
 * ends; it calls no hipMalloc / hipFree and creates no stream (default stream, blocking copies).
 * Limits: n_cams <= BA_STATIC_MAX_CAMS, n_pts <= BA_STATIC_MAX_PTS, n_obs <= BA_STATIC_MAX_OBS; beyond them it
 * returns BA_ERR_INVALID and the caller uses ba_evaluate.  device = HIP ordinal, -1 = current device.
 * The caller's current HIP device is the same after the call as before it (the call switches to `device` and back).
 * err (may be NULL) receives the message for a non-zero status (there is no handle to ask).
 * Calls on one GPU share that GPU's array: the library serialises them with a process-wide lock, so calls from
 * several threads are safe but do not overlap; calls from different PROCESSES use different copies. */
#define BA_STATIC_MAX_OBS 131072
#define BA_STATIC_MAX_CAMS 1024
#define BA_STATIC_MAX_PTS 65536
int ba_evaluate_static(int32_t device, int32_t n_cams, int32_t n_pts, int32_t n_obs, const int32_t* obs_cam,
                       const int32_t* obs_pt, const double* obs_uv, const double* intr, const double* poses,
                       const double* points, double* residuals, double* cost, int32_t* n_behind, char* err,
                       int32_t err_len);

/* Multi-view triangulation of every landmark of the current problem from KNOWN poses: the step on the
 * input side of the path (initial points for ba_solve).  For two views this is what
 * VisualOdometry.triangulate_points computes (visual_odometry.py:508-570: cv2.triangulatePoints, the linear
 * DLT estimate, then (e1 + e2) / 2 as reprojection error); here for any number of views of a landmark, and
 * with refine_iters > 0 followed by that many Gauss-Newton steps on the reprojection error at most (stops
 * at |step| < 1e-6), the refinement initialization.triangulate_landmark sketches (initialization.py:96-143).
 *   poses[C*7]     in, all held constant (quaternions are normalised)
 *   points[L*3]    out: written for landmarks flagged valid, others left untouched
 *   reproj_err[L]  out (may be NULL): mean pixel distance over the landmark's views; NaN if it has < 2 views
 *   valid[L]       out (may be NULL): 1 = at least 2 views, finite, and deeper than min_depth in every view
 *                  (initialization.py:148-155 uses 0.1; add_pts3d_to_map, visual_odometry.py:483-485, uses 0)
 * Radial distortion: pixels are undistorted for the linear estimate; the refinement and the error use the
 * full camera model.  Uses the trial buffers of the handle: a following ba_download_estimate needs a new solve. */
int ba_triangulate(ba_handle* h, const double* poses, int32_t refine_iters, double min_depth, double* points,
                   double* reproj_err, uint8_t* valid);

/* Analytic Jacobian blocks in the caller's observation order: Jc[n_obs*12] = d r / d(dt, dtheta)
 * (2x6 row-major, zero for fixed poses), Jp[n_obs*6] = d r / dP (2x3).  The reference has no analytic
 * Jacobian (scipy differentiates numerically, scipy/optimize/_numdiff.py); exported so tests can check
 * the blocks against finite differences of the reference residual (never scaled by a robust loss). */
int ba_jacobian_blocks(ba_handle* h, const double* poses, const double* points, double* Jc, double* Jp);

/* Debug export of one linearisation at (poses, points) with damping lambda, in the caller's index
 * order: U[C*36], V[L*9], W[n_obs*18] (6x3 row-major), gc[C*6], gp[L*3], S[(6C)^2] dense row-major with
 * identity rows for fixed cameras, rhs[6C] (the reduced system is S dc = -rhs), and the LM step
 * dc[C*6], dp[L*3] from the configured linear solver.  Any pointer may be NULL.  With a robust loss the
 * blocks are those of the re-weighted system (residuals and Jacobian rows times sqrt(rho')).  No reference counterpart (the reference never forms normal equations). */
int ba_debug_linearization(ba_handle* h, const double* poses, const double* points, double lambda, double* U,
                           double* V, double* W, double* gc, double* gp, double* S, double* rhs, double* dc,
                           double* dp);

/* Device time per kernel class, measured with HIP events on the solver's own stream during the
 * solves since the last ba_set_problem (needs opt.profile_kernels = 1).  Classes:
 * 0 k_linearize*, 1 k_schur*, 2 k_backsub, 3 k_cost, 4 the reduced camera solve (one-workgroup PCG, or every launch of
 * the Cholesky solvers), 5 other (k_block_precond, k_triangulate), 6 k_gather_S / k_reduce_S, 7 k_fold_stats + k_lm_decide
 * (two launches per LM iteration since version 101: the average is over both).
 * No reference counterpart (bench.py's roofline figures). */
#define BA_KERNEL_CLASSES 8
int ba_get_kernel_times(ba_handle* h, double* total_ms, int32_t* launches);
/* Switch the per-launch start/stop events (ba_options.profile_kernels) on or off for later solves.
 * No counterpart in the reference; used by bench.py, which times the solves without events and then
 * measures the kernels in a second pass. */
int ba_set_kernel_timing(ba_handle* h, int32_t on);

/* Multi-GPU (one process per GPU, landmark-sharded; no reference counterpart -- the reference is one
 * Python thread).  Rank 0 calls ba_comm_unique_id and ships the 128 bytes to the other ranks by any
 * means (bench.py: torch.distributed); every rank then calls ba_comm_init on its handle BEFORE
 * ba_set_problem.  Afterwards each rank describes ITS shard to ba_set_problem: all C cameras, its own
 * landmarks and their observations (obs_pt indexes the local landmark array).  Poses are replicated
 * (every rank passes and receives all of them); points are rank-local.  ba_solve* then runs the same
 * iteration on every rank with one RCCL all-reduce of [partial S | reduced rhs | U, gc] per linear solve and one
 * all-reduce of the scalar statistics; report fields are global.  There is no broadcast of the camera step: every rank
 * solves the same all-reduced system with a solver whose summation order is fixed, which keeps the replicated poses
 * bit-identical PROVIDED the all-reduce hands every rank the same bits (true of RCCL's ring and tree algorithms, where
 * every rank receives the one reduced buffer; checked with the test stand-in for librccl, not yet on a multi-GPU node).
 * Only the multi-kernel PCG, which reduces with atomics, ships rank 0's step.
 * BA_SHARDED_SOLVE=1 in the environment of EVERY rank (opt-in) shards the reduced solve of a trajectory by camera segment
 * instead: partial S per rank, one ncclReduce per rank of its segments of the two-level banded solver, local elimination
 * of the segment interiors, an all-reduce of the separator system, a replicated separator solve, local back-substitution,
 * an all-reduce of the camera step (one writer per entry: bit-identical poses); ba_problem_info[10] & 32 reports it.
 * Every rank needs at least one landmark and one observation.  ba_set_problem, ba_upload_guess and ba_solve* agree on
 * a status before their first data collective: if any rank's arguments are bad (an empty shard, an index out of range,
 * a non-finite guess) EVERY rank returns that status and none is left waiting inside RCCL.  ba_evaluate /
 * ba_jacobian_blocks / ba_triangulate stay rank-local; ba_solve_batched refuses a handle with a communicator (windows
 * are independent: give each rank its own plain handle).  With world == 1 the same code path runs with single-rank
 * collectives. */
int ba_comm_unique_id(char* id128);
int ba_comm_init(ba_handle* h, int32_t rank, int32_t world, const char* id128);
/* Rank and size of the handle's communicator as RCCL itself reports them (ncclCommUserRank / ncclCommCount); 0 and 1
 * for a handle without one.  No reference counterpart; bench.py's `n_gpus` is this figure. */
int ba_comm_info(const ba_handle* h, int32_t* rank, int32_t* world);

/* Digests of the index tables that the last ba_set_problem built and uploaded (the internal observation order, the block
 * pattern of S, the chunk / group / write-out / gather tables of the kernels), one 64-bit value per table in upload order;
 * recorded only when the environment holds BA_TABLE_DIGEST=1 (else *count = 0).  `out` receives min(cap, *count) values.
 * No reference counterpart (the reference has no tables: its pattern is the jac_sparsity mask of
 * bundle_adjustment.py:90-154); tests use it to assert that a faster set-up builds the same tables bit for bit. */
int ba_debug_table_digests(const ba_handle* h, uint64_t* out, int32_t cap, int32_t* count);

/* Which solver ba_set_problem chose for the reduced camera system: 0 multi-kernel PCG, 1 one-workgroup PCG,
 * 2 whole-GPU dense Cholesky, 3 banded + arrow Cholesky, 4 dense Cholesky per connected component (batched
 * windows), 5 block cyclic reduction.  No reference counterpart. */
int ba_reduced_solver(const ba_handle* h, int32_t* kind);

/* Sizes behind the per-kernel byte counts of bench.py (no reference counterpart): info[BA_INFO_COUNT] =
 * 0 stored upper 6x6 blocks of S, 1 contribution rows (42 doubles each) that the staged Schur kernel writes and the
 * gather reads per launch (0 on the atomics path), 2 Schur chunks, 3 copies of the U / gc slabs, 4 reduced solver (as
 * ba_reduced_solver), 5 / 6 half-bandwidth and arrow size of the banded solver (level 2 when segmented; cyclic reduction:
 * cameras per super-block and levels), 7 segments of the two-level banded solver (0: one level; cyclic reduction:
 * super-blocks), 8 free cameras, 9 observed landmarks, 10 kernel choices as bits (1: k_linearize_flat, 2: k_backsub_flat,
 * 4: matrix-free path: no stored W, the Schur and back-substitution kernels rebuild it; 8: k_linearize_win;
 * 16: k_linearize_rows, the fixed-order linearisation of round 3; 32: reduced solve sharded by camera segment, see the
 * multi-GPU notes), 11 connected components of the camera graph,
 * 12 rows of 27 doubles (U | gc of one camera within one chunk) that k_linearize_rows writes and the fold blocks of the
 * Schur launch read per linearisation (0 on the other paths). */
#define BA_INFO_COUNT 13
int ba_problem_info(const ba_handle* h, int64_t* info);

/* Number of stored 6x6 blocks of the reduced camera system (upper triangle incl. diagonal). */
int ba_schur_block_count(const ba_handle* h, int64_t* n_upper_blocks);

#ifdef __cplusplus
}
#endif
#endif /* BA_AMD_H */
