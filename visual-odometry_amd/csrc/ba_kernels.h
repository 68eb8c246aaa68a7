// HIP kernels of the sparse bundle-adjustment solver (gfx950 / MI355X, fp64).
//
// Data layout in HBM (all device pointers live in DeviceProblem, ba_solver.hip fills it):
//   observations, sorted by (internal landmark order, camera): SoA  o_cam[n] o_u[n] o_v[n]  (+ o_src[n]
//       = caller's index, used only by the export kernels); tracks are CSR: lm_ptr[L_obs+1]
//   poses  AoS [C][7], points AoS [L][3] (internal landmark order)
//   W      SoA [18][n]   (6x3 coupling block of observation k = W[(3 i + j) * n + k])
//   V      SoA [6][L]    (xx xy xz yy yz zz), gp SoA [3][L]
//   Ugc    AoS [C][27]   (21 upper-triangle entries of U, then 6 of gc), accumulated by atomics
//   S      6x6 blocks, row-major 36 doubles: upper blocks (incl. diagonal) in row-major (i, j>=i)
//          order = slots [0, nu); mirrored strictly-lower blocks = slots [nu, nu + nl)
//   vectors over free cameras: [Cf][6]
//
// Every kernel states the bytes it must move per unit (its HBM roofline numerator).
#pragma once
#include "ba_device.h"

enum ScalarSlot {
  SC_COST = 0, SC_COST_T, SC_GDOT, SC_DHD, SC_DNORM2, SC_XNORM2, SC_GINF, SC_NBEHIND, SC_NBEHIND_T, SC_BADV,
  SC_ZEROZ, SC_NONFINITE,
  // camera part of the step statistics: computed from replicated data, identical on every rank
  SC_GDOT_C, SC_DHD_C, SC_DNORM2_C, SC_XNORM2_C, SC_GINF_C,
  SC_NSHORT,  // landmark steps shortened by the per-landmark cheirality guard of k_backsub
  SC_NSUM = 20,  // slots [0, SC_NSUM) are folded over the privatised copies (and over ranks)
  SC_RZ = 24, SC_RZ0, SC_PQ, SC_RZNEW, SC_DONE, SC_PCGIT,
  SC_COUNT = 32
};
// Scalar accumulators are privatised too: block b adds into copy (b mod SCAL_SLABS); the host sums
// (max for SC_GINF) the copies after the read-back.  Same-address fp64 atomics from hundreds of
// blocks otherwise serialise at the memory side and dominate these short kernels.
constexpr int SCAL_SLABS = 64;
#define SCAL_MINE(d) ((d).scal + (size_t)(blockIdx.x & (SCAL_SLABS - 1)) * SC_COUNT)

constexpr int LM_HISTORY = 128;  // == BA_HISTORY (include/ba_amd.h)
constexpr int LM_ACC_BITS = 16384;  // accept / reject of every iteration up to this many (1 bit each), for the kernel timers

// Levenberg-Marquardt state that lives on the device: k_lm_decide takes the accept / reject decision
// and updates the damping, so the host enqueues iterations without waiting for each one.
struct LmHead {
  // options
  double ftol, xtol, gtol, lambda_min, lambda_max, accept_ratio;
  int max_iters, guard_cheirality;
  // state
  double lam, nu, cost, cost0, nbehind, ginf_pts, ginf;
  int sel;       // poses_buf[sel] / pts_buf[sel] hold the current estimate, [sel ^ 1] the trial point
  int need_lin;  // the next iteration starts with a linearisation
  int done, status, error;  // error: 1 = z == 0 at the initial guess, 2 = non-finite initial cost
  int it, accepted, linearizations, pcg_total, n_hist;
  int bad_pivots, guard_rejections;  // non-positive pivots / non-SPD blocks met by the solves; steps the cheirality guard refused
  int shortened_steps, pad_;         // landmark steps that k_backsub shortened to keep the landmark in front of its cameras
};
struct LmCtrl : LmHead {
  double cost_hist[LM_HISTORY], lam_hist[LM_HISTORY], trial_hist[LM_HISTORY], rho_hist[LM_HISTORY],
      ginf_hist[LM_HISTORY], step_hist[LM_HISTORY];
  int pcg_hist[LM_HISTORY], acc_hist[LM_HISTORY];
  unsigned acc_bits[LM_ACC_BITS / 32];  // bit i: iteration i + 1 accepted its step (the histories stop at LM_HISTORY)
};
static_assert(sizeof(LmHead) % 8 == 0, "LmHead must not have tail padding");
// Copy of the head in pinned host memory: k_lm_decide stores it there and then raises `seq`, the host
// spins on `seq` instead of paying a copy + stream synchronisation per LM iteration.
struct LmMirror {
  LmHead head;
  int seq, pad;
};

// Quadratic pose prior (ba_set_pose_prior; kernels in ba_prior.h)
constexpr int PRIOR_MAX_CAMS = 16;  // == BA_PRIOR_MAX_CAMS (include/ba_amd.h)
struct PriorDev {
  int n;               // cameras under the prior (0: none)
  double c0;
  const int* cam;      // [n] camera index
  const int* fr;       // [n] free-camera index
  const int* slot;     // [n * n] upper S slot of pair (a, b) when free(a) <= free(b), else -1
  const double* ref;   // [n][7] chart origins
  const double* H;     // [6n][6n]
  const double* g;     // [6n]
};

// Levenberg-Marquardt per connected component of the camera graph (batched windows: every window has its own
// damping sequence and its own accept / reject decisions; k_lm_decide_comp, k_commit_comp).  All null: one damping
// for the whole problem.
constexpr int CL_STRIDE = 6;  // per-component statistics: 0 g.d + d^T H d / 2 | 2 trial cost | 3 trial behind | 4 bad pivot
struct CompLm {
  double* lam;          // [n_comp]
  double* nu;           // [n_comp]
  double* cost;         // [n_comp] cost of the component at the current estimate
  double* nbehind;      // [n_comp]
  double* stat;         // [n_comp][CL_STRIDE]
  int* accept;          // [n_comp] decision of the last iteration
  int* done;            // [n_comp] converged (or given up): the component is frozen
  double* lam_lm;       // [L_obs] damping per landmark (= lam[lm_comp]): what the streaming kernels load
  const int* lm_comp;   // [L_obs]
  const int* free_comp; // [Cf]
  // Round 3: the per-component statistics without atomics.  Every chunk of the flat kernels (a wave) stores one entry
  // per distinct component among its landmarks (usually one): cp_ptr [chunks + 1] / cp_comp [entries] list them,
  // cp_val [entries][3] = model decrease of the landmark steps | trial cost | trial observations behind a camera;
  // cc_ptr [n_comp + 1] / cc_ent list a component's entries in ascending order, pose_val [Cf] is the model decrease of
  // every free camera's step.  k_lm_decide_comp (k_comp_fold_cost for the initial cost) adds them up in that order.
  const int* cp_ptr; const int* cp_comp; double* cp_val;
  const int* cc_ptr; const int* cc_ent; double* pose_val;
};

struct DeviceProblem {
  int C, Cf, L, L_obs, n;  // cameras, free cameras, landmarks, landmarks with >= 1 observation, observations
  Intrinsics K;
  RobustLoss loss;
  // structure (constant per problem)
  const int* o_cam; const double* o_u; const double* o_v; const int* o_src; const int* o_lm;
  const int* o_free;     // [n]  free-camera index of the observation's image, -1 = fixed pose
  const int* lm_ptr;     // [L_obs + 1]
  const int* cam2free;   // [C]   -1 = fixed
  const int* free2cam;   // [Cf]
  const int* fixed_cams; // [C - Cf]
  const int* urow_ptr;   // [Cf + 1] upper-triangle block rows
  const int* ucol;       // [nu]
  const int* urow;       // [nu]   row of each upper slot
  const int* diag_slot;  // [Cf]
  const int* low_of;     // [nu]   lower-mirror slot of upper slot s, or -1 on the diagonal
  const int* brow_ptr;   // [Cf + 1] full (both triangles) block rows for SpMV
  const int* bcol;       // [nb]
  const int* bslot;      // [nb]
  const int* sm_slot;    // [nu]  upper slots in diagonal-major order (k_solve_small item order)
  const int* sm_desc;    // [nu][4] per item: slot, row camera i, column camera j, 27 * camera of i (one load instead of a chain)
  const int* sm_gidx;    // [nb]  k_solve_small gather list, in brow_ptr order: LDS slot offset of each block of a row
  int c_ag, c_nc;        // coarse space of the one-workgroup PCG (ba_coarse.h): cameras per aggregate, coarse unknowns
  // connected components of the camera graph (k_solve_components)
  const int* comp_ptr;       // [n_comp + 1] into comp_cams
  const int* comp_cams;      // [Cf] free cameras grouped by component
  const int* comp_local;     // [Cf] index of a free camera inside its component
  const int* comp_slot_ptr;  // [n_comp + 1] into comp_slots
  const int* comp_slots;     // [nu] upper S slots grouped by component
  int n_comp;
  const int* lin_lm;     // [n_lin_chunks + 1] first landmark of each k_linearize_flat chunk
  const int* lin_k;      // [n_lin_chunks + 1] first observation of each chunk
  const unsigned char* lin_perm;  // [n] k_linearize_win: row of the observation when its chunk's free observations are put in camera order
  const int* lin_gptr;   // [n_lin_chunks + 1] first entry of each chunk in the group tables (groups + 1 sentinel per chunk)
  const int* lin_gbeg;   // [...] first position of each camera group of a chunk (sentinel: the chunk's free observations)
  const int* lin_gcam;   // [...] the group's camera
  const int* lin_gslot;  // [...] its slot in the workgroup's LDS copy of U | gc, or -1 (straight to the global slab)
  const int* lin_wincam; // [workgroups][LINW_WIN] camera of each slot, -1 = unused
  int n_lin_chunks;      // 0: no chunk tables (some track is longer than LINF_MAXLM)
  // fixed-order accumulation of U | gc and of the scalar statistics (k_linearize_rows, fold_u_rows, k_fold_stats): no
  // atomics, so that two runs of a solve take the same LM path bit for bit
  double* Urows;         // [3][n_urows][9] U | gc summed per (chunk, camera group of the chunk): row = index into lin_gcam
  long n_urows;          // entries of the group tables (one sentinel per chunk included); 0: the atomics kernels linearise
  const int* ucr_ptr;    // [C * nslab_u + 1] rows of camera c, cut into nslab_u segments: segment s is ucr_row[ucr_ptr[c * nslab_u + s] .. [+ 1])
  const int* ucr_row;    // [...] ascending
  double* lin_part;      // [n_lin_chunks][4] per chunk: cost, z <= 0 count, z == 0 count, max |gp|
  double* bs_part;       // [n_bs_blocks][8] per block of k_backsub_flat: its eight statistics
  double* pose_part;     // [n_pose_blocks][5] per pose block: the camera part of the step statistics
  int n_bs_blocks, n_pose_blocks;
  int lin_flat;          // 1: k_linearize_flat takes the chunks (few cameras); k_backsub_flat takes them whenever they exist
  const int* sch_lm;     // [n_chunks + 1] first landmark of each k_schur_staged chunk
  const int* sch_k;      // [n_chunks + 1] first observation of each chunk (= lm_ptr[sch_lm[.]])
  const unsigned* sch_flush;  // [n_chunks] bit s: the wave writes its accumulators out before step s of the chunk
  const int* sch_blk_row;     // [blocks]   first contribution row of each block of k_schur_staged
  const int* ctr_ptr;         // [nu + 1]   inverted index: contribution rows of each upper S block ...
  const int* ctr_row;         // [...]      ... in ascending row order (fixed summation order)
  double* contrib;            // [rows][CTR_ROW] contribution rows (6x6 block | 6 rhs entries)
  int n_chunks;          // 0: some track is too long for the staged kernel, k_schur + k_reduce_S are used
  int nu, nl, nb;
  // state
  double* W; double* V; double* gp; double* Ugc;
  double* zmin;          // [L] smallest depth of the landmark over its cameras at the last linearisation (k_backsub's guard)
  // S (upper blocks), rhs and Ugc are consecutive in ONE allocation [nu*36 | 6*Cf | 27*C]: the buffer
  // that the multi-GPU all-reduce sums (ba_solver.hip).  Ugc_part is this GPU's partial U / gc.
  double* S; double* rhs; double* Ugc_part; double* Minv;
  // privatised accumulation slabs (atomics from different groups land in different copies; a reduce
  // kernel sums the copies): Ugc_slab [nslab_u][C][27]; S_slab [nslab_s][nu*36 + 6*Cf] (S upper | rhs)
  double* Ugc_slab; double* S_slab;
  int nslab_u, nslab_s;
  long s_slab_stride;
  double* dc; double* r; double* z; double* p; double* q;   // PCG vectors (dc is the solution x)
  double* scal;
  LmCtrl* ctrl;
  double* poses_buf[2]; double* pts_buf[2];
  PriorDev prior;
  CompLm clm;
};

// ------------------------------------------------------------------------------------------------
// k_linearize: residual + analytic 2x6 / 2x3 Jacobian per observation, block assembly.
// One group of G lanes walks a contiguous range of landmark tracks; lane s of the group takes
// observation slot s (+G, +2G, ...) of each track.  V and gp are reduced across the group with
// shuffles and stored (tracks are exclusively owned: no atomics).  U/gc are accumulated in registers
// for as long as the lane keeps seeing the same camera (tracks are camera-sorted and landmarks are
// ordered by camera signature, so runs are long) and flushed with one fp64 atomic per entry per run.
// Algorithmic bytes: read 20 B/obs (cam idx + u + v) + 24 B/landmark (P), write W 144 B/obs,
// V 48 + gp 24 B/landmark, U/gc 216 B per camera run.
// W = Jc^T Jp (6x3 per observation) is stored FACTORED: the translation half of Jc is -Jp (ba_device.h:
// linearize_obs), so the 12 numbers [Jp (2x3) | rotation half of Jc (2x3)] are the whole of it: 96 B per observation
// instead of 144 B, written once by the linearisation and read by the Schur and back-substitution kernels, which
// rebuild W with 36 multiply-adds.  An observation of a fixed camera stores zeros (its W is zero).  Rows: structure
// of arrays, d.W[r * n + k], r < W_ROWS.
constexpr int W_ROWS = 12;
__device__ __forceinline__ void w_pack(const double Jc[12], const double Jp[6], bool is_free, double j[W_ROWS]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    j[c] = is_free ? Jp[c] : 0.0; j[3 + c] = is_free ? Jp[3 + c] : 0.0;
    j[6 + c] = is_free ? Jc[3 + c] : 0.0; j[9 + c] = is_free ? Jc[9 + c] : 0.0;
  }
}
__device__ __forceinline__ void w_from_j(const double j[W_ROWS], double w[18]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      w[3 * i + c] = -(j[i] * j[c] + j[3 + i] * j[3 + c]);
      w[9 + 3 * i + c] = j[6 + i] * j[c] + j[9 + i] * j[3 + c];
    }
}

constexpr int LIN_STRIDE = 29;    // 27 values + pad (odd number of doubles: 2-way LDS write conflicts at most)
constexpr int SCHUR_STRIDE = 37;  // 36 values + pad

#ifndef LIN_WAVES
#define LIN_WAVES 2
#endif
// ROBUST: compile the re-weighting of a robust loss in (a run-time branch costs the plain kernel registers
// it does not have: 78 spilled VGPRs, 24.5 -> 29.4 us at BASELINE configs[1]).
template <int G, bool ROBUST>
__global__ __launch_bounds__(64, ROBUST ? 1 : LIN_WAVES) void k_linearize(DeviceProblem d, const double* __restrict__ poses,
                                                  const double* __restrict__ pts, int lm_per_group) {
  __shared__ double stage[64 * LIN_STRIDE];
  if (poses == nullptr) {  // solver mode: state and the run / skip decision come from the device-side LM control
    const LmCtrl* c = d.ctrl;
    if (c->done || !c->need_lin) return;
    poses = d.poses_buf[c->sel];
    pts = d.pts_buf[c->sel];
  }
  const int lane = threadIdx.x;  // one wave per block
  const int gl = lane & (G - 1);
  // the wave owns GPW * lm_per_group consecutive tracks; in iteration jj its GPW groups take GPW
  // CONSECUTIVE tracks (coalesced index / pixel loads and W stores across the groups)
  constexpr int GPW = 64 / G;
  const long j0 = min((long)d.L_obs, (long)blockIdx.x * GPW * lm_per_group) + lane / G;
  const long j1 = min((long)d.L_obs, ((long)blockIdx.x + 1) * GPW * lm_per_group);
  const long n = d.n;
  double* __restrict__ Uslab = d.Ugc_slab + (long)(blockIdx.x % d.nslab_u) * d.C * 27;
  double acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = 0.0;
  int cur_cam = -1;
  double cost = 0.0, gmax = 0.0;
  int nbehind = 0, nzero = 0;

  auto flush = [&]() {  // wave-uniform: every lane calls it
    if (cur_cam >= 0) {
#pragma unroll
      for (int i = 0; i < 27; ++i) stage[lane * LIN_STRIDE + i] = acc[i];
    }
    __syncthreads();
    wave_scatter_add<27, LIN_STRIDE>(stage, cur_cam >= 0, (long)cur_cam * 27, Uslab);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
    cur_cam = -1;
  };

  // software prefetch of the next track's bounds, first observation and point (hides two dependent
  // round trips per track)
  int kb_n = 0, ke_n = 0, cam_n = -1, fr_n = -1;
  double u_n = 0.0, v_n = 0.0, P_n[3] = {0, 0, 0};
  auto prefetch = [&](long j) {
    kb_n = 0; ke_n = 0; cam_n = -1; fr_n = -1;
    if (j < j1) {
      kb_n = d.lm_ptr[j]; ke_n = d.lm_ptr[j + 1];
      P_n[0] = pts[3 * j]; P_n[1] = pts[3 * j + 1]; P_n[2] = pts[3 * j + 2];
      if (kb_n + gl < ke_n) {
        cam_n = d.o_cam[kb_n + gl]; fr_n = d.o_free[kb_n + gl];
        u_n = d.o_u[kb_n + gl]; v_n = d.o_v[kb_n + gl];
      }
    }
  };
  prefetch(j0);
  for (int jj = 0; jj < lm_per_group; ++jj) {
    const long j = j0 + (long)jj * GPW;
    const bool lm_ok = j < j1;
    const int kb = kb_n, ke = ke_n, cam_first = cam_n, fr_first = fr_n;
    const double u_first = u_n, v_first = v_n;
    const double P[3] = {P_n[0], P_n[1], P_n[2]};
    prefetch(j + GPW);
    double v[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    double zmn = 1e300;
    for (int k0 = kb; __any(k0 < ke); k0 += G) {
      const int k = k0 + gl;
      const bool act = k < ke;
      const bool first = (k0 == kb);
      const int cam = !act ? -1 : first ? cam_first : d.o_cam[k];
      const bool is_free = act && (first ? fr_first : d.o_free[k]) >= 0;
      // flush BEFORE the linearisation so that only the accumulators are live across it
      if (__any(is_free && cur_cam >= 0 && cam != cur_cam)) flush();
      if (act) {
        double r[2], Jc[12], Jp[6], zc;
        linearize_obs(d.K, poses + 7 * cam, P, first ? u_first : d.o_u[k], first ? v_first : d.o_v[k], r, Jc, Jp, zc);
        nbehind += (zc <= 0.0);
        nzero += (zc == 0.0);
        zmn = fmin(zmn, zc);
        if (!ROBUST) {
          cost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
        } else {  // robust loss: scale each residual and its Jacobian row
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            double s;
            cost += robust_scale(d.loss, r[i], s);
#pragma unroll
            for (int c = 0; c < 6; ++c) Jc[6 * i + c] *= s;
#pragma unroll
            for (int c = 0; c < 3; ++c) Jp[3 * i + c] *= s;
          }
        }
        v[0] += Jp[0] * Jp[0] + Jp[3] * Jp[3];
        v[1] += Jp[0] * Jp[1] + Jp[3] * Jp[4];
        v[2] += Jp[0] * Jp[2] + Jp[3] * Jp[5];
        v[3] += Jp[1] * Jp[1] + Jp[4] * Jp[4];
        v[4] += Jp[1] * Jp[2] + Jp[4] * Jp[5];
        v[5] += Jp[2] * Jp[2] + Jp[5] * Jp[5];
        g[0] += Jp[0] * r[0] + Jp[3] * r[1];
        g[1] += Jp[1] * r[0] + Jp[4] * r[1];
        g[2] += Jp[2] * r[0] + Jp[5] * r[1];
        if (d.W) {  // (no buffer: the consumers rebuild W themselves)
          double jw[W_ROWS];
          w_pack(Jc, Jp, is_free, jw);
#pragma unroll
          for (int i = 0; i < W_ROWS; ++i) d.W[(long)i * n + k] = jw[i];
        }
        if (is_free) {
          cur_cam = cam;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int c = i; c < 6; ++c) acc[sym6_index(i, c)] += Jc[i] * Jc[c] + Jc[6 + i] * Jc[6 + c];
            acc[21 + i] += Jc[i] * r[0] + Jc[6 + i] * r[1];
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = group_sum<G>(v[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) g[i] = group_sum<G>(g[i]);
    zmn = -group_max<G>(-zmn);
    if (gl == 0 && lm_ok) {
      d.zmin[j] = zmn;
#pragma unroll
      for (int i = 0; i < 6; ++i) d.V[(long)i * d.L + j] = v[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) d.gp[(long)i * d.L + j] = g[i];
      gmax = fmax(gmax, fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2]))));
    }
  }
  flush();
  cost = group_sum<64>(cost);
  gmax = group_max<64>(gmax);
  const double bb = group_sum<64>((double)nbehind), bz = group_sum<64>((double)nzero);
  if (lane == 0) {
    double* sc = SCAL_MINE(d);
    atomic_add_f64(&sc[SC_COST], cost);
    if (bb != 0.0) atomic_add_f64(&sc[SC_NBEHIND], bb);
    if (bz != 0.0) atomic_add_f64(&sc[SC_ZEROZ], bz);
    atomic_max_nonneg(&sc[SC_GINF], gmax);
  }
}

// ------------------------------------------------------------------------------------------------
// k_linearize_flat: the same linearisation with one LANE PER OBSERVATION, for problems with few cameras
// (<= LINF_MAXC, so U / gc of all of them fit in LDS) and tracks of at most LINF_MAXLM observations.
// k_linearize<G> keeps a group of G lanes on one track: with 10 observations per track 40 of 64 lanes work,
// and its instruction stream per track is dominated by bookkeeping (measured: 826 VALU instructions per
// wave-iteration of 40 observations, two waves per SIMD more than half VALU-busy).  Here the host cuts the
// landmark list into chunks of whole tracks with <= 64 observations; a wave takes a chunk per iteration:
//   lane t loads observation t (coalesced), its point and pose (gathers from cache), linearises, stores its
//   W block (18 coalesced stores), parks its 2x3-block products in LDS; lane l < landmarks-in-chunk then sums
//   the rows of track l in track order (deterministic) and stores V, gp.  U / gc go into a per-workgroup LDS
//   copy by LDS atomics, and the workgroup adds that copy to its global slab once.
constexpr int LINF_MAXC = 64;     // cameras: the per-workgroup copy of U | gc costs C * 27 LDS words to clear and scan (at 150 cameras
                                  // the kernel is slower than k_linearize<G>: 45 vs 32 us on 480 k observations)
constexpr int LINF_MAXLM = 32;    // landmarks per chunk == longest track
constexpr int LINF_VSTRIDE = 11;  // 9 values per observation + pad (odd: conflict-free column access)
constexpr int LINF_WAVES = 4;     // waves per workgroup
constexpr int LINR_VSTRIDE = 9;   // k_linearize_rows: nine values per row, no pad.  ds_read_b64 banks by (address / 4) mod 64 in two
                                  // halves of 32 lanes: the lanes (camera, entry) of the camera sums read nine consecutive doubles per
                                  // camera, windows that collide least when rows are dense (simulated for typical chunks and measured:
                                  // stride 11 cost 60-72 LDS cycles per phase where 9 costs 30-42 and the ideal is 18-27)

template <bool ROBUST>
__global__ __launch_bounds__(64 * LINF_WAVES, ROBUST ? 2 : 4) void k_linearize_flat(DeviceProblem d, const double* __restrict__ poses,
                                                                       const double* __restrict__ pts) {
  // One chunk per wave and no loop: every load of the wave is issued in two dependent rounds (chunk bounds
  // by scalar loads; then observation records, track bounds and the chunk's points, all independent), poses
  // come from an LDS copy.  (A software-pipelined loop over several chunks was slower: loads and stores
  // retire in order through one counter, and the compiler waits for everything outstanding, the previous
  // chunk's stores included, before each dependent round.)
  extern __shared__ double lds[];
  double* sU = lds;                                                   // [C][27]
  double* sPose = sU + d.C * 27;                                      // [C][7]
  double* sVg = sPose + d.C * 7;                                      // [LINF_WAVES][64][LINF_VSTRIDE]
  double* sP = sVg + LINF_WAVES * 64 * LINF_VSTRIDE;                  // [LINF_WAVES][LINF_MAXLM][3]
  if (poses == nullptr) {  // solver mode
    const LmCtrl* c = d.ctrl;
    if (c->done || !c->need_lin) return;
    poses = d.poses_buf[c->sel];
    pts = d.pts_buf[c->sel];
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * LINF_WAVES + wave);
  const bool have = c < d.n_lin_chunks;
  // round 1: chunk bounds (scalar), poses -> LDS
  int kb = 0, ke = 0, jb = 0, je = 0;
  if (have) { jb = d.lin_lm[c]; je = d.lin_lm[c + 1]; kb = d.lin_k[c]; ke = d.lin_k[c + 1]; }
  for (int i = tid; i < d.C * 7; i += 64 * LINF_WAVES) sPose[i] = poses[i];
  for (int i = tid; i < d.C * 27; i += 64 * LINF_WAVES) sU[i] = 0.0;
  // round 2: observation record, track bounds and points of the chunk
  const int k = kb + lane;
  const bool act = k < ke;
  int cam = -1, fr = -1, lm = jb, tb = 0, te = 0;
  double u = 0.0, v = 0.0, Pl[3] = {0.0, 0.0, 1.0};
  if (act) { cam = d.o_cam[k]; fr = d.o_free[k]; lm = d.o_lm[k]; u = d.o_u[k]; v = d.o_v[k]; }
  const bool lm_lane = lane < je - jb;
  if (lm_lane) {
    tb = d.lm_ptr[jb + lane] - kb; te = d.lm_ptr[jb + lane + 1] - kb;
#pragma unroll
    for (int i = 0; i < 3; ++i) Pl[i] = pts[3 * (long)(jb + lane) + i];
  }
  double* myP = sP + wave * LINF_MAXLM * 3;
  double* myV = sVg + (wave * 64) * LINF_VSTRIDE;
  if (lm_lane) {
#pragma unroll
    for (int i = 0; i < 3; ++i) myP[3 * lane + i] = Pl[i];
  }
  __syncthreads();  // sPose, sU, myP
  const bool is_free = act && fr >= 0;
  double cost = 0.0, gmax = 0.0;
  int nbehind = 0, nzero = 0;
  double vg[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) vg[i] = 0.0;
  double zc = 1e300;
  if (act) {
    const double P[3] = {myP[3 * (lm - jb)], myP[3 * (lm - jb) + 1], myP[3 * (lm - jb) + 2]};
    double pose[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) pose[i] = sPose[7 * cam + i];
    double r[2], Jc[12], Jp[6];
    linearize_obs(d.K, pose, P, u, v, r, Jc, Jp, zc);
    nbehind += (zc <= 0.0);
    nzero += (zc == 0.0);
    if (!ROBUST) {
      cost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        double sc;
        cost += robust_scale(d.loss, r[i], sc);
#pragma unroll
        for (int q = 0; q < 6; ++q) Jc[6 * i + q] *= sc;
#pragma unroll
        for (int q = 0; q < 3; ++q) Jp[3 * i + q] *= sc;
      }
    }
    vg[0] = Jp[0] * Jp[0] + Jp[3] * Jp[3];
    vg[1] = Jp[0] * Jp[1] + Jp[3] * Jp[4];
    vg[2] = Jp[0] * Jp[2] + Jp[3] * Jp[5];
    vg[3] = Jp[1] * Jp[1] + Jp[4] * Jp[4];
    vg[4] = Jp[1] * Jp[2] + Jp[4] * Jp[5];
    vg[5] = Jp[2] * Jp[2] + Jp[5] * Jp[5];
    vg[6] = Jp[0] * r[0] + Jp[3] * r[1];
    vg[7] = Jp[1] * r[0] + Jp[4] * r[1];
    vg[8] = Jp[2] * r[0] + Jp[5] * r[1];
    if (!is_free) {
#pragma unroll
      for (int i = 0; i < 12; ++i) Jc[i] = 0.0;
    }
    if (d.W) {  // (no buffer: the consumers rebuild W themselves)
      double* __restrict__ Wk = d.W + k;
      const long n = d.n;
      double jw[W_ROWS];
      w_pack(Jc, Jp, is_free, jw);
#pragma unroll
      for (int i = 0; i < W_ROWS; ++i) Wk[(long)i * n] = jw[i];
    }
    if (is_free) {  // U / gc of this observation into the workgroup's copy
      double* Uc = sU + cam * 27;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int q = i; q < 6; ++q) atomic_add_f64(&Uc[sym6_index(i, q)], Jc[i] * Jc[q] + Jc[6 + i] * Jc[6 + q]);
        atomic_add_f64(&Uc[21 + i], Jc[i] * r[0] + Jc[6 + i] * r[1]);
      }
    }
  }
  // V, gp of the chunk's landmarks: rows parked per observation, summed per track in track order
#pragma unroll
  for (int i = 0; i < 9; ++i) myV[lane * LINF_VSTRIDE + i] = vg[i];
  myV[lane * LINF_VSTRIDE + 9] = zc;
  __syncthreads();  // myV rows; all U / gc contributions of the workgroup
  if (lm_lane) {
    const int jl = jb + lane;
    double s9[9], zmn = 1e300;
#pragma unroll
    for (int i = 0; i < 9; ++i) s9[i] = 0.0;
    for (int t = tb; t < te; ++t) {
#pragma unroll
      for (int i = 0; i < 9; ++i) s9[i] += myV[t * LINF_VSTRIDE + i];
      zmn = fmin(zmn, myV[t * LINF_VSTRIDE + 9]);
    }
    d.zmin[jl] = zmn;
#pragma unroll
    for (int i = 0; i < 6; ++i) d.V[(long)i * d.L + jl] = s9[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) d.gp[(long)i * d.L + jl] = s9[6 + i];
    gmax = fmax(fabs(s9[6]), fmax(fabs(s9[7]), fabs(s9[8])));
  }
  double* __restrict__ Uslab = d.Ugc_slab + (long)(blockIdx.x % d.nslab_u) * d.C * 27;
  for (int i = tid; i < d.C * 27; i += 64 * LINF_WAVES) {
    const double x = sU[i];
    if (x != 0.0) atomic_add_f64(Uslab + i, x);
  }
  cost = wave_sum_dpp(cost);
  gmax = group_max<64>(gmax);
  const double bb = (double)__popcll(__ballot(nbehind != 0)), bz = (double)__popcll(__ballot(nzero != 0));  // (0 or 1 per lane)
  if (lane == 0 && have) {
    double* sc = SCAL_MINE(d);
    atomic_add_f64(&sc[SC_COST], cost);
    if (bb != 0.0) atomic_add_f64(&sc[SC_NBEHIND], bb);
    if (bz != 0.0) atomic_add_f64(&sc[SC_ZEROZ], bz);
    atomic_max_nonneg(&sc[SC_GINF], gmax);
  }
}

// ------------------------------------------------------------------------------------------------
// k_linearize_win: the lane-per-observation linearisation for ANY number of cameras.  k_linearize_flat keeps a copy of
// U | gc of all cameras per workgroup (LDS atomics, <= 64 cameras); here the workgroup's copy holds the <= LINW_WIN
// cameras that its four chunks touch (the host assigns the slots: the structure is static; a camera beyond the window
// goes straight to the global slab), and the 27 values of an observation do not go through LDS atomics, where the
// lanes of one camera collide (13 of 64 lanes on tracks of five): every lane parks its products in LDS, nine entries
// at a time, in the row that a host-made permutation gives it (camera order: the rows of a camera are consecutive), and
// one lane per (camera of the chunk, entry) sums that camera's rows, then adds the sum to the window.  Poses are gathered from global memory (L2 hits).
#ifndef LINW_WIN_SLOTS
#define LINW_WIN_SLOTS 16  // 32 KB of LDS per workgroup: five workgroups per CU (32 slots: four; 49 vs 47.5 us at config 3, 391 vs 367 at config 4)
#endif
constexpr int LINW_WIN = LINW_WIN_SLOTS;
template <bool ROBUST>
__global__ __launch_bounds__(64 * LINF_WAVES, ROBUST ? 2 : 4) void k_linearize_win(DeviceProblem d, const double* __restrict__ poses,
                                                                      const double* __restrict__ pts) {
  __shared__ double sWin[LINW_WIN * 27];
  __shared__ double sVg[LINF_WAVES][64 * LINF_VSTRIDE];
  __shared__ double sP[LINF_WAVES][LINF_MAXLM * 3];
  __shared__ int sGbeg[LINF_WAVES][65], sGslot[LINF_WAVES][64], sGcam[LINF_WAVES][64];
  if (poses == nullptr) {  // solver mode
    const LmCtrl* c = d.ctrl;
    if (c->done || !c->need_lin) return;
    poses = d.poses_buf[c->sel];
    pts = d.pts_buf[c->sel];
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * LINF_WAVES + wave);
  const bool have = c < d.n_lin_chunks;
  int kb = 0, ke = 0, jb = 0, je = 0, gb = 0, ng = 0;
  if (have) {
    jb = d.lin_lm[c]; je = d.lin_lm[c + 1]; kb = d.lin_k[c]; ke = d.lin_k[c + 1];
    gb = d.lin_gptr[c]; ng = d.lin_gptr[c + 1] - gb - 1;
  }
  for (int i = tid; i < LINW_WIN * 27; i += 64 * LINF_WAVES) sWin[i] = 0.0;
  const int k = kb + lane;
  const bool act = k < ke;
  int cam = 0, fr = -1, lm = jb, tb = 0, te = 0, pos = lane;  // pos: this observation's row in camera order
  double u = 0.0, v = 0.0, Pl[3] = {0.0, 0.0, 1.0};
  if (act) {
    cam = d.o_cam[k]; fr = d.o_free[k]; lm = d.o_lm[k]; u = d.o_u[k]; v = d.o_v[k];
    pos = d.lin_perm[k];
  }
  if (lane < ng) {
    sGbeg[wave][lane] = d.lin_gbeg[gb + lane]; sGslot[wave][lane] = d.lin_gslot[gb + lane]; sGcam[wave][lane] = d.lin_gcam[gb + lane];
  }
  if (lane == 0 && have) sGbeg[wave][ng] = d.lin_gbeg[gb + ng];  // sentinel: the number of free observations
  const bool lm_lane = lane < je - jb;
  if (lm_lane) {
    tb = d.lm_ptr[jb + lane] - kb; te = d.lm_ptr[jb + lane + 1] - kb;
#pragma unroll
    for (int i = 0; i < 3; ++i) Pl[i] = pts[3 * (long)(jb + lane) + i];
  }
  double pose[7] = {0, 0, 0, 1, 0, 0, 0};
  if (act) {
#pragma unroll
    for (int i = 0; i < 7; ++i) pose[i] = poses[7 * cam + i];
  }
  double* myP = sP[wave];
  double* myV = sVg[wave];
  if (lm_lane) {
#pragma unroll
    for (int i = 0; i < 3; ++i) myP[3 * lane + i] = Pl[i];
  }
  __syncthreads();  // sWin clear, myP, tables
  const bool is_free = act && fr >= 0;
  double cost = 0.0, gmax = 0.0;
  int nbehind = 0, nzero = 0;
  double vg[9], Jc[12], r[2] = {0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 9; ++i) vg[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 12; ++i) Jc[i] = 0.0;
  double zc = 1e300;
  if (act) {
    const double P[3] = {myP[3 * (lm - jb)], myP[3 * (lm - jb) + 1], myP[3 * (lm - jb) + 2]};
    double Jp[6];
    linearize_obs(d.K, pose, P, u, v, r, Jc, Jp, zc);
    nbehind += (zc <= 0.0);
    nzero += (zc == 0.0);
    if (!ROBUST) {
      cost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        double sc;
        cost += robust_scale(d.loss, r[i], sc);
#pragma unroll
        for (int q = 0; q < 6; ++q) Jc[6 * i + q] *= sc;
#pragma unroll
        for (int q = 0; q < 3; ++q) Jp[3 * i + q] *= sc;
      }
    }
    vg[0] = Jp[0] * Jp[0] + Jp[3] * Jp[3];
    vg[1] = Jp[0] * Jp[1] + Jp[3] * Jp[4];
    vg[2] = Jp[0] * Jp[2] + Jp[3] * Jp[5];
    vg[3] = Jp[1] * Jp[1] + Jp[4] * Jp[4];
    vg[4] = Jp[1] * Jp[2] + Jp[4] * Jp[5];
    vg[5] = Jp[2] * Jp[2] + Jp[5] * Jp[5];
    vg[6] = Jp[0] * r[0] + Jp[3] * r[1];
    vg[7] = Jp[1] * r[0] + Jp[4] * r[1];
    vg[8] = Jp[2] * r[0] + Jp[5] * r[1];
    if (!is_free) {
#pragma unroll
      for (int i = 0; i < 12; ++i) Jc[i] = 0.0;
    }
    if (d.W) {  // (no buffer: the consumers rebuild W themselves)
      double* __restrict__ Wk = d.W + k;
      const long n = d.n;
      double jw[W_ROWS];
      w_pack(Jc, Jp, is_free, jw);
#pragma unroll
      for (int i = 0; i < W_ROWS; ++i) Wk[(long)i * n] = jw[i];
    }
  }
  // V, gp of the chunk's landmarks: rows parked per observation, summed per track in track order
#pragma unroll
  for (int i = 0; i < 9; ++i) myV[lane * LINF_VSTRIDE + i] = vg[i];
  myV[lane * LINF_VSTRIDE + 9] = zc;
  // From here to the final flush a wave touches only its own rows (and the window through atomics): the LDS serves one
  // wave's accesses in order, so a wave-level barrier (no reordering by the compiler) is all that the row exchange needs.
  __builtin_amdgcn_wave_barrier();
  if (lm_lane) {
    const int jl = jb + lane;
    double s9[9], zmn = 1e300;
#pragma unroll
    for (int i = 0; i < 9; ++i) s9[i] = 0.0;
    for (int t = tb; t < te; ++t) {
#pragma unroll
      for (int i = 0; i < 9; ++i) s9[i] += myV[t * LINF_VSTRIDE + i];
      zmn = fmin(zmn, myV[t * LINF_VSTRIDE + 9]);
    }
    d.zmin[jl] = zmn;
#pragma unroll
    for (int i = 0; i < 6; ++i) d.V[(long)i * d.L + jl] = s9[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) d.gp[(long)i * d.L + jl] = s9[6 + i];
    gmax = fmax(fabs(s9[6]), fmax(fabs(s9[7]), fabs(s9[8])));
  }
  // U | gc: 27 values per observation, nine at a time through the same rows
  double* __restrict__ Uslab = d.Ugc_slab + (long)(blockIdx.x % d.nslab_u) * d.C * 27;
#pragma unroll
  for (int ph = 0; ph < 3; ++ph) {
    __builtin_amdgcn_wave_barrier();  // the rows' previous readers are done
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int q = i; q < 6; ++q) {
        const int e = sym6_index(i, q);
        if (e >= 9 * ph && e < 9 * ph + 9) myV[pos * LINF_VSTRIDE + e - 9 * ph] = Jc[i] * Jc[q] + Jc[6 + i] * Jc[6 + q];
      }
      const int e = 21 + i;
      if (e >= 9 * ph && e < 9 * ph + 9) myV[pos * LINF_VSTRIDE + e - 9 * ph] = Jc[i] * r[0] + Jc[6 + i] * r[1];
    }
    __builtin_amdgcn_wave_barrier();
    for (int item = lane; item < ng * 9; item += 64) {
      const int g = item / 9, q = item - 9 * g;
      // the rows of a camera are consecutive: four independent LDS reads in flight
      const int p1 = sGbeg[wave][g + 1];
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (int p = sGbeg[wave][g]; p < p1; p += 4) {
        const double* row = myV + p * LINF_VSTRIDE + q;
        const double a0 = row[0], a1 = p + 1 < p1 ? row[LINF_VSTRIDE] : 0.0, a2 = p + 2 < p1 ? row[2 * LINF_VSTRIDE] : 0.0,
                     a3 = p + 3 < p1 ? row[3 * LINF_VSTRIDE] : 0.0;
        s0 += a0; s1 += a1; s2 += a2; s3 += a3;
      }
      const double sum = (s0 + s1) + (s2 + s3);
      const int slot = sGslot[wave][g];
      if (slot >= 0) atomic_add_f64(&sWin[slot * 27 + 9 * ph + q], sum);
      else atomic_add_f64(&Uslab[(long)sGcam[wave][g] * 27 + 9 * ph + q], sum);
    }
  }
  __syncthreads();
  for (int i = tid; i < LINW_WIN * 27; i += 64 * LINF_WAVES) {
    const int s27 = i / 27, camw = d.lin_wincam[(long)blockIdx.x * LINW_WIN + s27];
    const double x = sWin[i];
    if (camw >= 0 && x != 0.0) atomic_add_f64(Uslab + (long)camw * 27 + (i - 27 * s27), x);
  }
  cost = wave_sum_dpp(cost);
  gmax = group_max<64>(gmax);
  const double bb = (double)__popcll(__ballot(nbehind != 0)), bz = (double)__popcll(__ballot(nzero != 0));  // (0 or 1 per lane)
  if (lane == 0 && have) {
    double* sc = SCAL_MINE(d);
    atomic_add_f64(&sc[SC_COST], cost);
    if (bb != 0.0) atomic_add_f64(&sc[SC_NBEHIND], bb);
    if (bz != 0.0) atomic_add_f64(&sc[SC_ZEROZ], bz);
    atomic_max_nonneg(&sc[SC_GINF], gmax);
  }
}

// ------------------------------------------------------------------------------------------------
// k_linearize_rows: the lane-per-observation linearisation WITHOUT atomics, for any number of cameras (round 3; it
// replaces k_linearize_flat and k_linearize_win wherever the chunk tables exist).  Those two sum U | gc through LDS
// atomics and then through fp64 atomics into privatised global slabs: the order in which the waves and workgroups
// arrive changes the last bits of U and gc from run to run, and Levenberg-Marquardt amplifies that on a flat valley
// (BASELINE configs[3]: 680 to 1900 iterations and 2.4e-5 of the cost between two runs of the same solve).  Here every
// sum has ONE owner and a fixed order:
//   * a wave takes one chunk of whole tracks (<= 64 observations), a lane per observation, exactly as before;
//   * V, gp: rows parked in LDS, summed per track in track order (unchanged);
//   * U | gc: every lane parks its 27 products, nine at a time, in the row that the host's permutation gives it (camera
//     order inside the chunk); one lane per (camera group of the chunk, entry) sums that camera's consecutive rows and
//     stores the sum, with a plain store, into row `first group of the chunk + g` of d.Urows: a row per (chunk, camera).
//     fold_u_rows (the leading blocks of the Schur launch) then sums, for every camera, the rows that the host's
//     inverted index lists for it, segment by segment, in ascending row order, into what used to be the slabs;
//   * cost, the two cheirality counters and max |gp|: one plain store per chunk into d.lin_part; k_fold_stats sums them.
// A wave touches nobody else's data: no workgroup barrier at all (the LDS serves one wave's accesses in order).
// Algorithmic bytes: those of k_linearize_win + 216 B per (chunk, camera) row written.
template <bool ROBUST>
__global__ __launch_bounds__(64 * LINF_WAVES, ROBUST ? 2 : 7) void k_linearize_rows(DeviceProblem d, const double* __restrict__ poses,
                                                                       const double* __restrict__ pts) {
  // 20.8 KB of LDS and at most 72 VGPRs: seven workgroups per CU (the landmark positions travel by shuffles, not through
  // a table in LDS, and the group table holds bytes: with both as before a CU took six)
  __shared__ double sVg[LINF_WAVES][64 * LINR_VSTRIDE];
  __shared__ double sZc[LINF_WAVES][64];  // the depth of every observation (the rows hold nine values: V | gp, then U | gc nine at a time)
  __shared__ unsigned char sGbeg[LINF_WAVES][68];
  if (poses == nullptr) {  // solver mode
    const LmCtrl* c = d.ctrl;
    if (c->done || !c->need_lin) return;
    poses = d.poses_buf[c->sel];
    pts = d.pts_buf[c->sel];
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * LINF_WAVES + wave);
  if (c >= d.n_lin_chunks) return;  // (no workgroup barrier below)
  const int jb = d.lin_lm[c], je = d.lin_lm[c + 1], kb = d.lin_k[c], ke = d.lin_k[c + 1];
  const int gb = d.lin_gptr[c], ng = d.lin_gptr[c + 1] - gb - 1;
  // the lanes beyond the chunk's observations repeat its last one (no branch around the arithmetic, no defaults to
  // set up for it): they park their products in rows of their own that nobody sums, and store nothing
  const bool act = kb + lane < ke;
  const int k = act ? kb + lane : ke - 1;
  int tb = 0, te = 0;
  double Pl[3] = {0.0, 0.0, 1.0};
  const int cam = d.o_cam[k], fr = d.o_free[k], lm = d.o_lm[k];
  const double u = d.o_u[k], v = d.o_v[k];
  const int pos = act ? (int)d.lin_perm[k] : lane;  // this observation's row in camera order
  if (lane <= ng) sGbeg[wave][lane] = (unsigned char)d.lin_gbeg[gb + lane];  // (entry ng: the sentinel = the chunk's free observations, <= 64)
  const bool lm_lane = lane < je - jb;
  if (lm_lane) {
    tb = d.lm_ptr[jb + lane] - kb; te = d.lm_ptr[jb + lane + 1] - kb;
#pragma unroll
    for (int i = 0; i < 3; ++i) Pl[i] = pts[3 * (long)(jb + lane) + i];
  }
  double pose[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) pose[i] = poses[7 * cam + i];
  double* myV = sVg[wave];
  __builtin_amdgcn_wave_barrier();  // the group table
  const bool is_free = act && fr >= 0;
  double cost = 0.0, gmax = 0.0;
  int nbehind = 0, nzero = 0;
  double vg[9], Jc[12], r[2];
  double zc;
  {
    const double P[3] = {__shfl(Pl[0], lm - jb), __shfl(Pl[1], lm - jb), __shfl(Pl[2], lm - jb)};  // from the landmark's lane
    double Jp[6];
    linearize_obs(d.K, pose, P, u, v, r, Jc, Jp, zc);
    nbehind += (act && zc <= 0.0);
    nzero += (act && zc == 0.0);
    if (!ROBUST) {
      cost = act ? 0.5 * (r[0] * r[0] + r[1] * r[1]) : 0.0;
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        double sc;
        const double rho_i = robust_scale(d.loss, r[i], sc);
        cost += act ? rho_i : 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) Jc[6 * i + q] *= sc;
#pragma unroll
        for (int q = 0; q < 3; ++q) Jp[3 * i + q] *= sc;
      }
    }
    vg[0] = Jp[0] * Jp[0] + Jp[3] * Jp[3];
    vg[1] = Jp[0] * Jp[1] + Jp[3] * Jp[4];
    vg[2] = Jp[0] * Jp[2] + Jp[3] * Jp[5];
    vg[3] = Jp[1] * Jp[1] + Jp[4] * Jp[4];
    vg[4] = Jp[1] * Jp[2] + Jp[4] * Jp[5];
    vg[5] = Jp[2] * Jp[2] + Jp[5] * Jp[5];
    vg[6] = Jp[0] * r[0] + Jp[3] * r[1];
    vg[7] = Jp[1] * r[0] + Jp[4] * r[1];
    vg[8] = Jp[2] * r[0] + Jp[5] * r[1];
    if (!is_free) {
#pragma unroll
      for (int i = 0; i < 12; ++i) Jc[i] = 0.0;
    }
    if (d.W && act) {  // (no buffer: the consumers rebuild W themselves)
      double* __restrict__ Wk = d.W + k;
      const long n = d.n;
      double jw[W_ROWS];
      w_pack(Jc, Jp, is_free, jw);
#pragma unroll
      for (int i = 0; i < W_ROWS; ++i) Wk[(long)i * n] = jw[i];
    }
  }
  // V, gp of the chunk's landmarks: rows parked per observation, summed per track in track order
#pragma unroll
  for (int i = 0; i < 9; ++i) myV[lane * LINR_VSTRIDE + i] = vg[i];
  sZc[wave][lane] = zc;
  __builtin_amdgcn_wave_barrier();
  // a lane per (track, entry) for the nine sums of a track (V 6, gp 3), the track's rows added in their order (the same
  // sums, bit for bit, as one lane per track made; with a dozen tracks per chunk that left four lanes in five idle and
  // cost half of the kernel's LDS instructions); the smallest depth of a track stays with the track's lane
  {
    const int n_items = (je - jb) * 9;
    for (int base = 0; base < n_items; base += 64) {  // (uniform: every lane takes part in the shuffles)
      const int item = base + lane;
      const bool on = item < n_items;
      const int t = on ? item / 9 : 0, i = item - 9 * t;
      const int p0 = __shfl(tb, t), p1 = __shfl(te, t);
      if (on) {
        double a = 0.0;
        int p = p0;
        const double* row = myV + p0 * LINR_VSTRIDE + i;
        for (; p + 4 <= p1; p += 4, row += 4 * LINR_VSTRIDE) {  // four LDS reads in flight, added one after the other
          const double x0 = row[0], x1 = row[LINR_VSTRIDE], x2 = row[2 * LINR_VSTRIDE], x3 = row[3 * LINR_VSTRIDE];
          a = (((a + x0) + x1) + x2) + x3;
        }
        for (; p < p1; ++p, row += LINR_VSTRIDE) a += row[0];
        const int jl = jb + t;
        if (i < 6) d.V[(long)i * d.L + jl] = a;
        else { d.gp[(long)(i - 6) * d.L + jl] = a; gmax = fmax(gmax, fabs(a)); }
      }
    }
    if (lm_lane) {
      double zmn = 1e300;
      for (int t = tb; t < te; ++t) zmn = fmin(zmn, sZc[wave][t]);
      d.zmin[jb + lane] = zmn;
    }
  }
  // U | gc: 27 values per observation, nine at a time through the same rows; the sum of a camera group goes out as
  // nine consecutive doubles of plane ph of d.Urows (a wave's stores of a phase are contiguous: ng * 9 doubles)
#pragma unroll
  for (int ph = 0; ph < 3; ++ph) {
    __builtin_amdgcn_wave_barrier();  // the rows' previous readers are done
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int q = i; q < 6; ++q) {
        const int e = sym6_index(i, q);
        if (e >= 9 * ph && e < 9 * ph + 9) myV[pos * LINR_VSTRIDE + e - 9 * ph] = Jc[i] * Jc[q] + Jc[6 + i] * Jc[6 + q];
      }
      const int e = 21 + i;
      if (e >= 9 * ph && e < 9 * ph + 9) myV[pos * LINR_VSTRIDE + e - 9 * ph] = Jc[i] * r[0] + Jc[6 + i] * r[1];
    }
    __builtin_amdgcn_wave_barrier();
    double* __restrict__ plane = d.Urows + ((long)ph * d.n_urows + gb) * 9;
    for (int item = lane; item < ng * 9; item += 64) {
      const int g = item / 9, q = item - 9 * g;
      // the rows of a camera are consecutive: four independent LDS reads in flight; fixed order
      const int p1 = sGbeg[wave][g + 1];
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int p = sGbeg[wave][g];
      const double* row = myV + p * LINR_VSTRIDE + q;
      for (; p + 4 <= p1; p += 4, row += 4 * LINR_VSTRIDE) {  // whole steps without the tail's tests
        const double a0 = row[0], a1 = row[LINR_VSTRIDE], a2 = row[2 * LINR_VSTRIDE], a3 = row[3 * LINR_VSTRIDE];
        s0 += a0; s1 += a1; s2 += a2; s3 += a3;
      }
      if (p < p1) {  // the last one to three rows, into the accumulators they would have met in a padded step
        s0 += row[0];
        if (p + 1 < p1) s1 += row[LINR_VSTRIDE];
        if (p + 2 < p1) s2 += row[2 * LINR_VSTRIDE];
      }
      plane[item] = (s0 + s1) + (s2 + s3);
    }
  }
  cost = wave_sum_dpp(cost);
  gmax = group_max<64>(gmax);
  const double bb = (double)__popcll(__ballot(nbehind != 0)), bz = (double)__popcll(__ballot(nzero != 0));  // (0 or 1 per lane)
  if (lane < 4) d.lin_part[4L * c + lane] = lane == 0 ? cost : lane == 1 ? bb : lane == 2 ? bz : gmax;
}

// fold_u_rows: segment `s` of camera `cam`: the sum of the rows of d.Urows that the inverted index lists for it, in
// ascending row order, into slab s of d.Ugc_slab (which the gather kernels sum over s, as they summed the atomics'
// slabs).  One wave: lanes 0..26 take the even rows of the segment for entry `lane`, lanes 27..53 the odd ones, eight
// loads in flight each; the two halves are added at the end.  Every (camera, segment) is written by every
// linearisation, rows or no rows: nothing needs clearing.
__device__ __forceinline__ void fold_u_rows(const DeviceProblem& d, int idx, int lane) {
  const int cam = idx / d.nslab_u, s = idx - cam * d.nslab_u;
  const int r0 = d.ucr_ptr[idx], cnt = d.ucr_ptr[idx + 1] - r0;
  const int h = lane >= 27 ? 1 : 0, e = lane - 27 * h, ph = e / 9, q = e - 9 * ph;  // (lanes >= 54: idle, e clamped below)
  const bool worker = lane < 54;
  const double* __restrict__ plane = d.Urows + (long)(worker ? ph : 0) * d.n_urows * 9 + (worker ? q : 0);
  double acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.0;
  for (int kb = 0; kb < cnt; kb += 64) {
    const int mine = kb + lane < cnt ? d.ucr_row[r0 + kb + lane] : 0;
    const int m = min(64, cnt - kb);
    for (int k0 = 0; k0 < m; k0 += 16) {  // 16 rows per round: eight per half
      double v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int kk = k0 + 2 * i + h;
        const long row = __shfl(mine, kk & 63);
        v[i] = (worker && kk < m) ? plane[row * 9] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
  }
  double sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  sum += __shfl(sum, (lane + 27) & 63);
  if (lane < 27) d.Ugc_slab[((long)s * d.C + cam) * 27 + lane] = sum;
}

// the fold as a launch of its own (the path whose Schur kernel is not k_schur_staged, and the explicit-mode exports)
__global__ __launch_bounds__(64) void k_fold_U(DeviceProblem d, int solver_mode) {
  if (solver_mode && (d.ctrl->done || !d.ctrl->need_lin)) return;
  fold_u_rows(d, blockIdx.x, threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// k_fold_stats: the per-chunk statistics of k_linearize_rows, the per-block statistics of k_backsub_flat and those of
// the pose blocks, summed in a fixed order into the SCAL_SLABS copies of the scalar block that k_lm_decide (or
// k_fold_scalars, or the host) folds as before: workgroup s sums a contiguous range of each array (a thread per
// (entry, sub-range), then a fixed tree over the sub-ranges) and adds the result to copy s with plain read-modify-
// writes (it is the only writer of these slots in its launch).  Replaces eight + four + five fp64 atomics per block.
// with_lin: the linearisation's partial sums are current (solver mode: the control block knows).
// The statistics of every component for k_lm_decide_comp (batched solves), a wave per component: lane l adds the entries
// l, l + 64, ... of the component's list and its cameras l, l + 64, ... (ascending), the lanes are added in the fixed
// order of wave_sum_dpp: stat[w][0] = model decrease, [2] = trial cost, [3] = trial observations behind a camera.  No
// atomics.  Runs as the blocks after the first SCAL_SLABS of k_fold_stats' grid (no launch of its own).
__device__ __forceinline__ void comp_fold_stats_block(const DeviceProblem& d, int block) {
  const CompLm& q = d.clm;
  const int w = block * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= d.n_comp) return;
  double st0 = 0.0, cost_t = 0.0, behind_t = 0.0;
  for (int i = q.cc_ptr[w] + lane; i < q.cc_ptr[w + 1]; i += 64) {
    const long e = q.cc_ent[i];
    st0 += q.cp_val[3 * e]; cost_t += q.cp_val[3 * e + 1]; behind_t += q.cp_val[3 * e + 2];
  }
  for (int i = d.comp_ptr[w] + lane; i < d.comp_ptr[w + 1]; i += 64) st0 += q.pose_val[d.comp_cams[i]];
  st0 = wave_sum_dpp(st0); cost_t = wave_sum_dpp(cost_t); behind_t = wave_sum_dpp(behind_t);
  if (lane == 0) {
    double* st = q.stat + (long)CL_STRIDE * w;
    st[0] = st0; st[2] = cost_t; st[3] = behind_t;
  }
}

__global__ __launch_bounds__(256) void k_fold_stats(DeviceProblem d, int solver_mode, int with_lin, int with_bs) {
  __shared__ double red_bs[4][8], red_lin[4][4];
  if (solver_mode) {
    if (d.ctrl->done) return;
    with_lin = with_lin && d.ctrl->need_lin;
  }
  if ((int)blockIdx.x >= SCAL_SLABS) {  // (batched solves only: the host adds ceil(n_comp / 4) blocks to the grid)
    comp_fold_stats_block(d, blockIdx.x - SCAL_SLABS);
    return;
  }
  with_lin = with_lin && d.n_urows > 0;
  with_bs = with_bs && d.n_bs_blocks > 0;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, s = blockIdx.x;
  double* __restrict__ sc = d.scal + (size_t)s * SC_COUNT;
  // k_backsub_flat's blocks: thread = (entry e, sub-range): sub-range q takes blocks b0 + q, b0 + q + 32, ...; a wave
  // holds 8 sub-ranges of every entry (lanes e + 8 j): three shuffle steps fold them, the four waves meet in LDS
  double v_bs = 0.0;
  if (with_bs) {
    const long b0 = (long)d.n_bs_blocks * s / SCAL_SLABS, b1 = (long)d.n_bs_blocks * (s + 1) / SCAL_SLABS;
    const int e = tid & 7, sub = tid >> 3;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (long b = b0 + sub; b < b1; b += 128) {
      const double* p = d.bs_part + 8 * b + e;
      const double x0 = p[0], x1 = b + 32 < b1 ? p[8 * 32] : 0.0, x2 = b + 64 < b1 ? p[8 * 64] : 0.0, x3 = b + 96 < b1 ? p[8 * 96] : 0.0;
      a0 += x0; a1 += x1; a2 += x2; a3 += x3;
    }
    v_bs = (a0 + a1) + (a2 + a3);
    v_bs += __shfl_xor(v_bs, 8, 64);
    v_bs += __shfl_xor(v_bs, 16, 64);
    v_bs += __shfl_xor(v_bs, 32, 64);
  }
  // k_linearize_rows' chunks: (entry e of 4, sub-range of 64); entry 3 is a maximum
  double v_lin = 0.0;
  if (with_lin) {
    const long b0 = (long)d.n_lin_chunks * s / SCAL_SLABS, b1 = (long)d.n_lin_chunks * (s + 1) / SCAL_SLABS;
    const int e = tid & 3, sub = tid >> 2;
    const bool is_max = e == 3;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (long b = b0 + sub; b < b1; b += 256) {
      const double* p = d.lin_part + 4 * b + e;
      const double x0 = p[0], x1 = b + 64 < b1 ? p[4 * 64] : 0.0, x2 = b + 128 < b1 ? p[4 * 128] : 0.0, x3 = b + 192 < b1 ? p[4 * 192] : 0.0;
      if (is_max) { a0 = fmax(a0, x0); a1 = fmax(a1, x1); a2 = fmax(a2, x2); a3 = fmax(a3, x3); }
      else { a0 += x0; a1 += x1; a2 += x2; a3 += x3; }
    }
    v_lin = is_max ? fmax(fmax(a0, a1), fmax(a2, a3)) : (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) {
      const double o = __shfl_xor(v_lin, m, 64);
      v_lin = is_max ? fmax(v_lin, o) : v_lin + o;
    }
  }
  if (lane < 8) red_bs[wv][lane] = v_bs;
  if (lane < 4) red_lin[wv][lane] = v_lin;
  __syncthreads();
  // one owner thread per slot of copy s: plain read-modify-writes (other kernels' atomics into this copy are complete)
  if (tid < 8) {
    const int slot = tid == 0 ? SC_GDOT : tid == 1 ? SC_DHD : tid == 2 ? SC_DNORM2 : tid == 3 ? SC_XNORM2 : tid == 4 ? SC_NSHORT
                     : tid == 5 ? SC_COST_T : tid == 6 ? SC_NBEHIND_T : SC_ZEROZ;
    double v = (red_bs[0][tid] + red_bs[1][tid]) + (red_bs[2][tid] + red_bs[3][tid]);
    if (tid == 7) v += (red_lin[0][2] + red_lin[1][2]) + (red_lin[2][2] + red_lin[3][2]);
    if (v != 0.0) sc[slot] += v;
  } else if (tid < 10) {
    const int e = tid - 8;
    const double v = (red_lin[0][e] + red_lin[1][e]) + (red_lin[2][e] + red_lin[3][e]);
    if (v != 0.0) sc[e == 0 ? SC_COST : SC_NBEHIND] += v;
  } else if (tid == 10) {
    const double v = fmax(fmax(red_lin[0][3], red_lin[1][3]), fmax(red_lin[2][3], red_lin[3][3]));
    if (v > 0.0) sc[SC_GINF] = fmax(sc[SC_GINF], v);
  } else if (tid >= 16 && tid < 21 && d.n_pose_blocks > 0) {
    // few pose blocks (cameras / 256): workgroup s takes pose blocks s, s + SCAL_SLABS, ...
    const int t = tid - 16;
    double v = 0.0;
    for (int b = s; b < d.n_pose_blocks; b += SCAL_SLABS) {
      const double x = d.pose_part[5L * b + t];
      v = t == 4 ? fmax(v, x) : v + x;
    }
    const int slot = t == 0 ? SC_GDOT_C : t == 1 ? SC_DHD_C : t == 2 ? SC_DNORM2_C : t == 3 ? SC_XNORM2_C : SC_GINF_C;
    if (t == 4) { if (v > 0.0) sc[slot] = fmax(sc[slot], v); } else if (v != 0.0) sc[slot] += v;
  }
}

// ------------------------------------------------------------------------------------------------
// k_init_solve: everything a solve needs before its first iteration, in one launch instead of a control-block
// upload, three device copies and a memset: the initial guess into the current buffers (points also into the
// trial buffer: landmarks nobody observes are never written again), the LM control head, clean accumulators.
__global__ __launch_bounds__(256) void k_init_solve(DeviceProblem d, const double* __restrict__ poses_init,
                                                    const double* __restrict__ pts_init, LmHead head, long n_clear) {
  const long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  for (long t = t0; t < 7L * d.C; t += stride) d.poses_buf[0][t] = poses_init[t];
  for (long t = t0; t < 3L * d.L; t += stride) {
    const double v = pts_init[t];
    d.pts_buf[0][t] = v;
    d.pts_buf[1][t] = v;
  }
  for (long t = t0; t < n_clear; t += stride) d.Ugc_slab[t] = 0.0;  // [U / gc slabs | scalar copies | S slabs]
  if (t0 == 0) *static_cast<LmHead*>(d.ctrl) = head;
  if (t0 < LM_ACC_BITS / 32) d.ctrl->acc_bits[t0] = 0u;
}

// ------------------------------------------------------------------------------------------------
// k_cost: residuals only (trial-point evaluation, ba_evaluate).  24 B/obs (+ cached P, pose).
// Flat over observations; the landmark of observation k is found from o_lm[k].
template <bool ROBUST>
__global__ __launch_bounds__(256) void k_cost(DeviceProblem d,
                                              const double* __restrict__ poses, const double* __restrict__ pts,
                                              int cost_slot, int behind_slot, double* __restrict__ res_out) {
  __shared__ double red[4];
  if (poses == nullptr) {  // solver mode: evaluate the trial point
    const LmCtrl* c = d.ctrl;
    if (c->done) return;
    poses = d.poses_buf[c->sel ^ 1];
    pts = d.pts_buf[c->sel ^ 1];
  }
  double cost = 0.0;
  int nbehind = 0, nzero = 0;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < d.n; k += (long)gridDim.x * blockDim.x) {
    const int cam = d.o_cam[k];
    const long j = d.o_lm[k];
    const double P[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
    double R[9], X[3], r0, r1;
    quat_to_R(poses + 7 * cam + 3, R);
    to_camera(R, poses + 7 * cam, P, X);
    reproject(d.K, X, d.o_u[k], d.o_v[k], r0, r1);
    nbehind += (X[2] <= 0.0);
    nzero += (X[2] == 0.0);
    cost += !ROBUST ? 0.5 * (r0 * r0 + r1 * r1) : robust_cost(d.loss, r0) + robust_cost(d.loss, r1);
    if (res_out) {
      const long s = d.o_src[k];
      res_out[2 * s] = r0;
      res_out[2 * s + 1] = r1;
    }
  }
  const double bc = block_sum_256(cost, red);
  const double bb = block_sum_256((double)nbehind, red);
  const double bz = block_sum_256((double)nzero, red);
  if (threadIdx.x == 0) {
    double* sc = SCAL_MINE(d);
    atomic_add_f64(&sc[cost_slot], bc);
    if (bb != 0.0) atomic_add_f64(&sc[behind_slot], bb);
    if (bz != 0.0) atomic_add_f64(&sc[SC_ZEROZ], bz);
  }
}

// ------------------------------------------------------------------------------------------------
// k_residuals_static / k_reduce_static: the residual pass of ba_evaluate for ba_evaluate_static, the entry
// point that allocates nothing.  Same arithmetic as k_cost (quat_to_R, to_camera, reproject); everything
// beyond the caller's data lives in the library's static scratch array, whose regions arrive as kernel
// arguments.  Observations are taken in the caller's order (no track sort), the residuals go out in that
// order, and every block leaves its three partial sums (cost, z <= 0 count, z == 0 count) in its own slot:
// no atomics, so the second kernel folds them in a fixed order and the cost is bitwise reproducible.
// Algorithmic bytes: 24 B/obs read + 16 B/obs written (+ cached P, pose).
__global__ __launch_bounds__(256) void k_residuals_static(Intrinsics K, int n, const int* __restrict__ cam,
                                                          const int* __restrict__ pt, const double* __restrict__ uv,
                                                          const double* __restrict__ poses,
                                                          const double* __restrict__ pts, double* __restrict__ res,
                                                          double* __restrict__ partial) {
  __shared__ double red[4];
  double cost = 0.0;
  int nbehind = 0, nzero = 0;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) {
    const int c = cam[k];
    const long j = pt[k];
    const double P[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
    double R[9], X[3], r0, r1;
    quat_to_R(poses + 7 * c + 3, R);
    to_camera(R, poses + 7 * c, P, X);
    reproject(K, X, uv[2 * k], uv[2 * k + 1], r0, r1);
    nbehind = X[2] <= 0.0;
    nzero = X[2] == 0.0;
    cost = 0.5 * (r0 * r0 + r1 * r1);
    res[2 * k] = r0;
    res[2 * k + 1] = r1;
  }
  const double bc = block_sum_256(cost, red);
  const double bb = block_sum_256((double)nbehind, red);
  const double bz = block_sum_256((double)nzero, red);
  if (threadIdx.x == 0) {
    partial[3 * blockIdx.x] = bc;
    partial[3 * blockIdx.x + 1] = bb;
    partial[3 * blockIdx.x + 2] = bz;
  }
}

// one block: out[0..2] = column sums of partial[nblocks][3], summed in block order
__global__ __launch_bounds__(256) void k_reduce_static(const double* __restrict__ partial, int nblocks,
                                                       double* __restrict__ out) {
  __shared__ double red[4];
  double a[3] = {0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    a[0] += partial[3 * b]; a[1] += partial[3 * b + 1]; a[2] += partial[3 * b + 2];
  }
  for (int i = 0; i < 3; ++i) {
    const double s = block_sum_256(a[i], red);
    if (threadIdx.x == 0) out[i] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// k_triangulate: multi-view triangulation of every landmark track from known poses -- the step on the
// input side of the path.  One lane per landmark (the eigen-solve dominates: spreading
// a track over G lanes repeats it G times and was 2-4x slower).
//   1. linear (DLT) estimate, the algorithm of cv2.triangulatePoints that the reference calls for two
//      views (visual_odometry.py:546-556), for any number of views: rows u P_2 - P_0 and v P_2 - P_1 of
//      every view (P = K [R^T | -R^T c]) are accumulated into the 4x4 matrix A^T A, whose eigenvector of
//      the smallest eigenvalue (Rayleigh-quotient iteration in registers, cyclic Jacobi as the fallback) is
//      the homogeneous point.  Distorted pixels
//      are undistorted first (fixed-point iteration on the radial factor).
//   2. optional Gauss-Newton refinement of the reprojection error with the full camera model
//      (initialization.py:96-143 sketches it with finite differences; analytic 2x3 blocks here): at most
//      `refine_iters` steps, stops when |delta| < 1e-6.
//   3. mean reprojection distance over the views ((e1 + e2) / 2 of visual_odometry.py:565-568) and a
//      validity flag: finite, and deeper than min_depth in every view (initialization.py:148-155).
// Algorithmic bytes: 20 B/obs (camera index + pixel) + 33 B/landmark written, poses from cache;
// refinement re-reads the observations from L2.
// One Jacobi rotation annihilating M[p][q] of the packed symmetric 4x4 matrix (p < q, compile-time).
__host__ __device__ constexpr int sym4_index(int i, int j) {
  return i <= j ? 4 * i - i * (i - 1) / 2 + (j - i) : 4 * j - j * (j - 1) / 2 + (i - j);
}
__device__ __forceinline__ void jacobi_rotate4(double M[10], double V[16], int p, int q) {
  const double apq = M[sym4_index(p, q)];
  if (apq == 0.0) return;
  const double app = M[sym4_index(p, p)], aqq = M[sym4_index(q, q)];
  const double theta = (aqq - app) / (2.0 * apq);
  const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
  M[sym4_index(p, p)] = app - t * apq;
  M[sym4_index(q, q)] = aqq + t * apq;
  M[sym4_index(p, q)] = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r != p && r != q) {
      const double arp = M[sym4_index(r, p)], arq = M[sym4_index(r, q)];
      M[sym4_index(r, p)] = c * arp - s * arq;
      M[sym4_index(r, q)] = s * arp + c * arq;
    }
    const double vrp = V[4 * r + p], vrq = V[4 * r + q];
    V[4 * r + p] = c * vrp - s * vrq;
    V[4 * r + q] = s * vrp + c * vrq;
  }
}

// Eigenvector of the smallest eigenvalue of the packed symmetric PSD 4x4 matrix M (sym4_index order), for a
// point at finite distance.  Start: the inhomogeneous solution (w = 1) of the same rows; then Rayleigh-quotient
// iteration (cubic convergence: three steps) with an LDL^T factorisation of M - mu I without pivoting: its
// leading 3x3 block stays positive definite for a finite point (interlacing), only the last pivot is ~0.  The
// number of negative pivots counts the eigenvalues below mu: more than one, or a residual that is not small,
// means the iteration went elsewhere (degenerate geometry) and the cyclic Jacobi solver takes over.
__device__ __forceinline__ bool smallest_eigvec4_rqi(const double M[10], double x[4]) {
  const double v6[6] = {M[0], M[1], M[2], M[4], M[5], M[7]};
  double inv[6];
  damped_inv3(v6, 0.0, 0.0, inv);
  const double rhs[3] = {-M[3], -M[6], -M[8]};
  sym3_mul(inv, rhs, x);
  x[3] = 1.0;
  const double trace = M[0] + M[4] + M[7] + M[9];
  const double tiny = 1e-300 + 1e-30 * trace;
  bool ok = true;
  double mu = 0.0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    double y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      y[i] = 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) y[i] += M[sym4_index(i, c)] * x[c];
    }
    const double xx = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    mu = (x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3]) / xx;
    if (it == 3) {  // verdict: |M x - mu x| small against |M| |x|
      double res = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) res += (y[i] - mu * x[i]) * (y[i] - mu * x[i]);
      ok = ok && res <= 1e-22 * trace * trace * xx;
      break;
    }
    // LDL^T of M - mu I (unit lower L, diagonal D)
    double L10, L20, L30, L21, L31, L32, D0, D1, D2, D3;
    D0 = M[0] - mu;
    const double i0 = 1.0 / D0;
    L10 = M[1] * i0; L20 = M[2] * i0; L30 = M[3] * i0;
    D1 = (M[4] - mu) - L10 * M[1];
    const double i1 = 1.0 / D1;
    const double t21 = M[5] - L20 * M[1], t31 = M[6] - L30 * M[1];
    L21 = t21 * i1; L31 = t31 * i1;
    D2 = (M[7] - mu) - L20 * M[2] - L21 * t21;
    const double i2 = 1.0 / D2;
    const double t32 = M[8] - L30 * M[2] - L31 * t21;
    L32 = t32 * i2;
    D3 = (M[9] - mu) - L30 * M[3] - L31 * t31 - L32 * t32;
    ok = ok && D0 > 0.0 && D1 > 0.0 && D2 > 0.0;  // exactly the smallest eigenvalue may lie below mu
    if (fabs(D3) < tiny) D3 = tiny;
    // solve (M - mu I) z = x
    double z0 = x[0], z1 = x[1] - L10 * z0, z2 = x[2] - L20 * z0 - L21 * z1, z3 = x[3] - L30 * z0 - L31 * z1 - L32 * z2;
    z0 *= i0; z1 *= i1; z2 *= i2; z3 /= D3;
    z2 -= L32 * z3;
    z1 -= L21 * z2 + L31 * z3;
    z0 -= L10 * z1 + L20 * z2 + L30 * z3;
    const double nz = 1.0 / sqrt(z0 * z0 + z1 * z1 + z2 * z2 + z3 * z3);
    x[0] = z0 * nz; x[1] = z1 * nz; x[2] = z2 * nz; x[3] = z3 * nz;
    ok = ok && isfinite(nz);
  }
  return ok && x[3] != 0.0;
}

__device__ __noinline__ void smallest_eigvec4_jacobi(const double M_in[10], double x[4]) {
  double M[10], V[16];
#pragma unroll
  for (int i = 0; i < 10; ++i) M[i] = M_in[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = M[1] * M[1] + M[2] * M[2] + M[3] * M[3] + M[5] * M[5] + M[6] * M[6] + M[8] * M[8];
    const double dg = M[0] * M[0] + M[4] * M[4] + M[7] * M[7] + M[9] * M[9];
    if (off <= 1e-32 * dg) break;
    jacobi_rotate4(M, V, 0, 1); jacobi_rotate4(M, V, 0, 2); jacobi_rotate4(M, V, 0, 3);
    jacobi_rotate4(M, V, 1, 2); jacobi_rotate4(M, V, 1, 3); jacobi_rotate4(M, V, 2, 3);
  }
  const double ev[4] = {M[0], M[4], M[7], M[9]};
  int m = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) m = ev[i] < ev[m] ? i : m;
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = m == 0 ? V[4 * i] : m == 1 ? V[4 * i + 1] : m == 2 ? V[4 * i + 2] : V[4 * i + 3];
}

__global__ __launch_bounds__(64) void k_triangulate(DeviceProblem d, const double* __restrict__ poses, int refine_iters,
                                                    double min_depth, int force_jacobi, double* __restrict__ pts_out,
                                                    double* __restrict__ err_out, unsigned char* __restrict__ valid_out) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d.L_obs) return;
  const int kb = d.lm_ptr[j], ke = d.lm_ptr[j + 1];
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  if (ke - kb < 2) {
    err_out[j] = nan;
    valid_out[j] = 0;
    return;
  }
  const Intrinsics K = d.K;
  // A^T A, packed upper triangle: 00 01 02 03 11 12 13 22 23 33
  double M[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) M[i] = 0.0;
  for (int k = kb; k < ke; ++k) {
    const double* pose = poses + 7 * d.o_cam[k];
    double R[9];
    quat_to_R(pose + 3, R);
    double u = d.o_u[k], v = d.o_v[k];
    if (K.has_dist) {  // pixel of the undistorted ray
      const double xd = (u - K.cx) / K.fx, yd = (v - K.cy) / K.fy;
      double xn = xd, yn = yd;
      for (int it = 0; it < 20; ++it) {
        const double r2 = xn * xn + yn * yn;
        const double f = 1.0 + K.k1 * r2 + K.k2 * r2 * r2 + K.k3 * r2 * r2 * r2;
        xn = xd / f;
        yn = yd / f;
      }
      u = K.fx * xn + K.cx;
      v = K.fy * yn + K.cy;
    }
    // rows of [R^T | -R^T c]: row i = (R[0][i], R[1][i], R[2][i], -R[:, i] . c)
    double E[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      E[i][0] = R[i]; E[i][1] = R[3 + i]; E[i][2] = R[6 + i];
      E[i][3] = -(R[i] * pose[0] + R[3 + i] * pose[1] + R[6 + i] * pose[2]);
    }
    double a[4], b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double p0 = K.fx * E[0][c] + K.cx * E[2][c], p1 = K.fy * E[1][c] + K.cy * E[2][c], p2 = E[2][c];
      a[c] = u * p2 - p0;
      b[c] = v * p2 - p1;
    }
    int e = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = r; c < 4; ++c) M[e++] += a[r] * a[c] + b[r] * b[c];
  }
  double hx[4];
  if (force_jacobi || !smallest_eigvec4_rqi(M, hx)) smallest_eigvec4_jacobi(M, hx);
  double X[3] = {hx[0] / hx[3], hx[1] / hx[3], hx[2] / hx[3]};

  for (int it = 0; it < refine_iters; ++it) {
    double Hh[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int k = kb; k < ke; ++k) {
      double r[2], Jc[12], Jp[6], zc;
      linearize_obs(K, poses + 7 * d.o_cam[k], X, d.o_u[k], d.o_v[k], r, Jc, Jp, zc);
      Hh[0] += Jp[0] * Jp[0] + Jp[3] * Jp[3]; Hh[1] += Jp[0] * Jp[1] + Jp[3] * Jp[4]; Hh[2] += Jp[0] * Jp[2] + Jp[3] * Jp[5];
      Hh[3] += Jp[1] * Jp[1] + Jp[4] * Jp[4]; Hh[4] += Jp[1] * Jp[2] + Jp[4] * Jp[5]; Hh[5] += Jp[2] * Jp[2] + Jp[5] * Jp[5];
      g[0] += Jp[0] * r[0] + Jp[3] * r[1]; g[1] += Jp[1] * r[0] + Jp[4] * r[1]; g[2] += Jp[2] * r[0] + Jp[5] * r[1];
    }
    double Hi[6], dx[3];
    damped_inv3(Hh, 0.0, 0.0, Hi);
    sym3_mul(Hi, g, dx);
    X[0] -= dx[0]; X[1] -= dx[1]; X[2] -= dx[2];
    if (dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2] < 1e-12) break;
  }

  double err = 0.0;
  bool ok = isfinite(X[0]) && isfinite(X[1]) && isfinite(X[2]);
  for (int k = kb; k < ke; ++k) {
    const double* pose = poses + 7 * d.o_cam[k];
    double R[9], Xc[3], r0, r1;
    quat_to_R(pose + 3, R);
    to_camera(R, pose, X, Xc);
    reproject(K, Xc, d.o_u[k], d.o_v[k], r0, r1);
    err += sqrt(r0 * r0 + r1 * r1);
    ok = ok && Xc[2] > min_depth;
  }
  err /= (double)(ke - kb);
  pts_out[3 * j] = X[0]; pts_out[3 * j + 1] = X[1]; pts_out[3 * j + 2] = X[2];
  err_out[j] = err;
  valid_out[j] = (ok && isfinite(err)) ? 1 : 0;
}

// Export kernel for ba_jacobian_blocks (caller's observation order).
__global__ __launch_bounds__(256) void k_export_jac(DeviceProblem d,
                                                    const double* __restrict__ poses, const double* __restrict__ pts,
                                                    double* __restrict__ Jc_out, double* __restrict__ Jp_out) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < d.n; k += (long)gridDim.x * blockDim.x) {
    const int cam = d.o_cam[k];
    const long j = d.o_lm[k];
    const double P[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
    double r[2], Jc[12], Jp[6], zc;
    linearize_obs(d.K, poses + 7 * cam, P, d.o_u[k], d.o_v[k], r, Jc, Jp, zc);
    const bool is_free = d.cam2free[cam] >= 0;
    const long s = d.o_src[k];
    for (int i = 0; i < 12; ++i) Jc_out[12 * s + i] = is_free ? Jc[i] : 0.0;
    for (int i = 0; i < 6; ++i) Jp_out[6 * s + i] = Jp[i];
  }
}

// The factors of W of every observation at (poses, pts), in the layout of d.W: what the debug export reads when the
// solver keeps no W buffer (matrix-free path).
__global__ __launch_bounds__(256) void k_export_w(DeviceProblem d, const double* __restrict__ poses,
                                                  const double* __restrict__ pts, double* __restrict__ Wout) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < d.n; k += (long)gridDim.x * blockDim.x) {
    const int cam = d.o_cam[k];
    const long j = d.o_lm[k];
    const double P[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
    double r[2], Jc[12], Jp[6], zc, jw[W_ROWS];
    linearize_obs(d.K, poses + 7 * cam, P, d.o_u[k], d.o_v[k], r, Jc, Jp, zc);
    if (d.loss.kind != LOSS_LINEAR) {
      for (int i = 0; i < 2; ++i) {
        double sc;
        (void)robust_scale(d.loss, r[i], sc);
        for (int q = 0; q < 6; ++q) Jc[6 * i + q] *= sc;
        for (int q = 0; q < 3; ++q) Jp[3 * i + q] *= sc;
      }
    }
    w_pack(Jc, Jp, d.o_free[k] >= 0, jw);
    for (int i = 0; i < W_ROWS; ++i) Wout[(long)i * d.n + k] = jw[i];
  }
}

// ------------------------------------------------------------------------------------------------
// k_reduce_U: Ugc[c][e] = sum over slabs; inf-norm of the camera gradient.  One thread per entry.
__global__ __launch_bounds__(256) void k_reduce_U(DeviceProblem d, int solver_mode) {
  __shared__ double red[4];
  if (solver_mode && (d.ctrl->done || !d.ctrl->need_lin)) return;
  // 16 entries x 16 slab-lanes per block: lane l sums slabs l, l+16, ...; shuffle-reduce over lanes
  const int e_in_blk = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long t = (long)blockIdx.x * 16 + e_in_blk;
  const long tot = (long)d.C * 27;
  double sum = 0.0;
  if (t < tot)
    for (int k = sl; k < d.nslab_u; k += 16) sum += d.Ugc_slab[(long)k * tot + t];
  // lanes of one entry are 16 apart inside the block: combine through LDS
  __shared__ double part[256];
  part[threadIdx.x] = sum;
  __syncthreads();
  double gmax = 0.0;
  if (threadIdx.x < 16 && t < tot) {
    double tsum = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) tsum += part[threadIdx.x + 16 * k];
    d.Ugc_part[t] = tsum;
  }
  (void)red;
  (void)gmax;
}

// k_reduce_S: S (upper blocks) = sum of the slabs (+ partial U on the diagonal blocks), rhs = partial gc
// + sum of the slabs' rhs part, Ugc = partial U/gc.  One thread per entry of [nu*36 | 6*Cf | 27*C].
__global__ __launch_bounds__(256) void k_reduce_S(DeviceProblem d, int solver_mode) {
  if (solver_mode && d.ctrl->done) return;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nS = (long)d.nu * 36, nR = 6L * d.Cf, nU = 27L * d.C;
  if (t >= nS + nR + nU) return;
  if (t >= nS + nR) {  // this GPU's partial U / gc, into the buffer the all-reduce sums
    d.Ugc[t - nS - nR] = d.Ugc_part[t - nS - nR];
    return;
  }
  double sum = 0.0;
  for (int k = 0; k < d.nslab_s; ++k) sum += d.S_slab[(long)k * d.s_slab_stride + t];
  if (t < nS) {
    const int s = (int)(t / 36), e = (int)(t - 36L * s);
    if (d.low_of[s] < 0) {
      const int f = d.urow[s], i = e / 6, c = e - 6 * i;
      sum += d.Ugc_part[(long)d.free2cam[f] * 27 + (i <= c ? sym6_index(i, c) : sym6_index(c, i))];
    }
    d.S[t] = sum;
  } else {
    const long r = t - nS;
    const int f = (int)(r / 6), i = (int)(r - 6L * f);
    d.rhs[r] = sum + d.Ugc_part[(long)d.free2cam[f] * 27 + 21 + i];
  }
}

// diagonal entry i of U for free camera f (Marquardt scaling), from the (all-reduced) Ugc
__device__ __forceinline__ double u_diag(const DeviceProblem& d, int f, int i) {
  return d.Ugc[(long)d.free2cam[f] * 27 + sym6_index(i, i)];
}

// Find the upper-triangle slot of block (fa, fb), fa <= fb, by binary search in row fa.
__device__ __forceinline__ int upper_slot(const DeviceProblem& d, int fa, int fb) {
  int lo = d.urow_ptr[fa], hi = d.urow_ptr[fa + 1] - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (d.ucol[mid] < fb) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------------------------------------------
// k_schur: reduced camera system, built per landmark track.
//   S[ca][cb] -= (W_a Vd^-1) W_b^T   for every pair a <= b of the track,   rhs[ca] -= (W_a Vd^-1) gp
// One group of G lanes walks a contiguous range of tracks; lane p of the group owns pair
// (a, b) = column-major upper enumeration p = b (b + 1) / 2 + a, which does not depend on the track
// length, so along a run of tracks with the same camera signature a lane keeps adding into the same
// S block: the 6x6 product is accumulated in registers and flushed with one fp64 atomic per entry
// per run.  Algorithmic bytes: W 144 B/obs (each W is re-read by the lanes of its track from L1/L2),
// V 48 + gp 24 B/landmark, 4 B/obs camera index; S writes 288 B per (pair, run).
template <int G>
__global__ __launch_bounds__(64, 2) void k_schur(DeviceProblem d, double lam, double min_diag, int lm_per_group) {
  __shared__ double stage[64 * SCHUR_STRIDE];
  __shared__ double stage_r[64 * 7];
  if (lam < 0.0) {  // solver mode
    if (d.ctrl->done) return;
    lam = d.ctrl->lam;
  }
  const int lane = threadIdx.x;  // one wave per block
  const int gl = lane & (G - 1);
  constexpr int GPW = 64 / G;  // same track-to-group interleaving as k_linearize
  const long j0 = min((long)d.L_obs, (long)blockIdx.x * GPW * lm_per_group) + lane / G;
  const long j1 = min((long)d.L_obs, ((long)blockIdx.x + 1) * GPW * lm_per_group);
  const long n = d.n;
  double acc[36], accr[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) accr[i] = 0.0;
  int cur_a = -1, cur_b = -1;
  bool cur_sym = false;
  int nbad = 0;
  double* __restrict__ Sslab = d.S_slab + (long)(blockIdx.x % d.nslab_s) * d.s_slab_stride;
  double* __restrict__ Rslab = Sslab + (long)d.nu * 36;

  auto flush = [&]() {  // wave-uniform: every lane calls it
    const bool have = cur_a >= 0;
    const bool diag = have && cur_a == cur_b && !cur_sym;
    long key = 0;
    if (have) {
      key = (long)upper_slot(d, cur_a, cur_b) * 36;
      if (cur_sym) {  // two observations of one landmark in the same image: block + block^T
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int c = 0; c < 6; ++c) stage[lane * SCHUR_STRIDE + 6 * i + c] = -(acc[6 * i + c] + acc[6 * c + i]);
      } else {
#pragma unroll
        for (int i = 0; i < 36; ++i) stage[lane * SCHUR_STRIDE + i] = -acc[i];
      }
      if (diag) {
#pragma unroll
        for (int i = 0; i < 6; ++i) stage_r[lane * 7 + i] = -accr[i];
      }
    }
    __syncthreads();
    wave_scatter_add<36, SCHUR_STRIDE>(stage, have, key, Sslab);
    wave_scatter_add<6, 7>(stage_r, diag, 6L * cur_a, Rslab);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) accr[i] = 0.0;
    cur_a = -1;
    cur_b = -1;
    cur_sym = false;
  };

  // pair owned by this lane in the first pass: p = gl = b (b + 1) / 2 + a, a <= b (static per lane)
  int b0 = (int)((sqrt(8.0 * gl + 1.0) - 1.0) * 0.5);
  while ((b0 + 1) * (b0 + 2) / 2 <= gl) ++b0;
  while (b0 * (b0 + 1) / 2 > gl) --b0;
  const int a0 = gl - b0 * (b0 + 1) / 2;
  // software prefetch: the track bounds and this lane's camera pair of landmark j + 1 are loaded
  // while landmark j is being processed (removes two dependent round trips per landmark)
  int kb_n = 0, np_n = 0, fa_n = -1, fb_n = -1;
  auto prefetch = [&](long j) {
    kb_n = 0; np_n = 0; fa_n = -1; fb_n = -1;
    if (j < j1) {
      kb_n = d.lm_ptr[j];
      const int len = d.lm_ptr[j + 1] - kb_n;
      np_n = len * (len + 1) / 2;
      if (gl < np_n) { fa_n = d.o_free[kb_n + a0]; fb_n = d.o_free[kb_n + b0]; }
    }
  };
  prefetch(j0);
  for (int jj = 0; jj < lm_per_group; ++jj) {
    const long j = j0 + (long)jj * GPW;
    const bool lm_ok = j < j1;
    const int kb = kb_n, npairs = np_n, fa_first = fa_n, fb_first = fb_n;
    prefetch(j + GPW);
    double v[6] = {1, 0, 0, 1, 0, 1}, vi[6], g[3] = {0, 0, 0};
    if (lm_ok) {
#pragma unroll
      for (int i = 0; i < 6; ++i) v[i] = d.V[(long)i * d.L + j];
#pragma unroll
      for (int i = 0; i < 3; ++i) g[i] = d.gp[(long)i * d.L + j];
    }
    const double det = damped_inv3(v, lam, min_diag, vi);
    if (lm_ok && !(det > 0.0) && gl == 0) ++nbad;
    for (int p0 = 0; __any(p0 < npairs); p0 += G) {
      const int p = p0 + gl;
      bool act = p < npairs;
      int a = a0, b = b0, fa = fa_first, fb = fb_first;
      if (act && p0 > 0) {
        b = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
        while ((b + 1) * (b + 2) / 2 <= p) ++b;
        while (b * (b + 1) / 2 > p) --b;
        a = p - b * (b + 1) / 2;
        fa = d.o_free[kb + a];
        fb = d.o_free[kb + b];
      }
      act = act && fa >= 0 && fb >= 0;
      const bool sym = act && (fa == fb) && (a != b);
      if (__any(act && cur_a >= 0 && (fa != cur_a || fb != cur_b || sym != cur_sym))) flush();
      if (act) {
        cur_a = fa; cur_b = fb; cur_sym = sym;
        const int ka = kb + a, kbb = kb + b;
        double Y[18], wa[18], wb[18], ja[W_ROWS], jb2[W_ROWS];
#pragma unroll
        for (int i = 0; i < W_ROWS; ++i) { ja[i] = d.W[(long)i * n + ka]; jb2[i] = d.W[(long)i * n + kbb]; }
        w_from_j(ja, wa);
        w_from_j(jb2, wb);
#pragma unroll
        for (int i = 0; i < 6; ++i) sym3_mul(vi, &wa[3 * i], &Y[3 * i]);
        if (a == b) {
#pragma unroll
          for (int i = 0; i < 6; ++i) accr[i] = fma(Y[3 * i + 2], g[2], fma(Y[3 * i + 1], g[1], fma(Y[3 * i], g[0], accr[i])));
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const double w0 = wb[3 * c], w1 = wb[3 * c + 1], w2 = wb[3 * c + 2];
#pragma unroll
          for (int i = 0; i < 6; ++i)  // three chained FMAs into the accumulator (a sum formed first costs a fourth instruction)
            acc[6 * i + c] = fma(Y[3 * i + 2], w2, fma(Y[3 * i + 1], w1, fma(Y[3 * i], w0, acc[6 * i + c])));
        }
      }
    }
  }
  flush();
  if (nbad) atomic_add_f64(&SCAL_MINE(d)[SC_BADV], (double)nbad);
}

// ------------------------------------------------------------------------------------------------
// k_schur_staged: the reduced-system build for problems whose tracks have at most 10 observations
// (<= 64 pairs: every pair of a track has a lane of its own).  ba_set_problem cuts the landmark list
// into chunks of <= SCH_CAP observations / <= SCH_MAXLM landmarks and gives each block `cpb` of them.
// k_schur reads W straight from HBM inside a dependent chain and adds its blocks into S with fp64
// atomics (which execute at the memory side, ~1.3 TB/s chip-wide); here a wave
//   1. loads a whole chunk with coalesced, independent loads (lane t takes observation t: the 12 factors
//      of its W and its camera slot; lane l takes landmark l: V, gp, track start) into REGISTERS while the
//      pairs of the previous chunk are being multiplied (software pipeline: HBM latency is hidden),
//   2. inverts the damped V blocks (one per lane) and factors the inverse, Vd^-1 = L L^T; forms
//      Z = W L = Jc^T (Jp L) ONCE per observation (48 multiply-adds; W itself is never formed) and parks it in LDS
//      (SZ_STRIDE doubles per observation): W_a Vd^-1 W_b^T = Z_a Z_b^T and W_a Vd^-1 gp = Z_a (L^T gp),
//   3. runs the pair products Z_a Z_b^T out of LDS (16-byte reads; three FMAs chained into each accumulator),
//      accumulating in registers along a run of tracks with the same camera signature,
//   4. at the end of a run (the host marked those steps in sch_flush: the structure is static)
//      writes its 64 x 42 accumulators (6x6 block | 6 rhs entries per lane) as ONE contiguous
//      contribution tile with plain coalesced stores.  k_gather_S then sums, for every S block, the
//      rows that the host's inverted index lists for it: no atomics, no memset, and S is bitwise
//      reproducible from run to run.
// Chunk shape: 64 observations with one 18-double factor each (round 1 and most of round 2: 40 observations with
// Y = W Vd^-1 and W, 36 doubles each, in the same 19 KB of LDS) and 7-column flush passes: 235 VGPRs, two waves per
// SIMD (8 blocks per CU).  Measured, 40 x (Y | W) -> 64 x Z -> chained FMAs: 25.2 -> 24.6 -> 22.9 us on BASELINE
// configs[1], 58.3 -> 53.4 -> 49.3 us on configs[2], 529 -> 483 -> 447 us on configs[3] (matrix-free).  The counters
// say why the arithmetic matters more than the traffic here: VALU-active 65 % of the SIMD cycles, LDS waits 8 %.
constexpr int SCH_CAP = 64;     // observations per chunk (<= lanes of a wave)
constexpr int SCH_MAXLM = 32;   // landmarks per chunk
constexpr int SCH_MAXLEN = 10;  // longest track the staged kernel takes (55 pairs <= 64 lanes)
constexpr int SCH_MAXCPB = 32;  // chunks per block
constexpr int SZ_STRIDE = 22;   // Z[18] | 4 pad doubles: rows stay 16-byte aligned, and the 16-byte reads of the <= 10 rows
                                // of a track start 44 dwords apart: ten distinct 4-bank slots
constexpr int CTR_ROW = 42;     // doubles per contribution row: 6x6 block | 6 reduced-rhs entries
constexpr int FLUSH_NV = 7, FLUSH_STRIDE = FLUSH_NV;  // columns per flush pass (divides CTR_ROW; odd: no LDS padding needed)

// MF ("matrix free"): W is not read from memory but rebuilt from the observation record, the pose and the point
// (linearize_obs: the same arithmetic as the linearisation kernels), with the loads in two stages so that none is
// waited for: the records of chunk ci + 2 and the poses (gathered by the camera index of the records loaded one stage
// earlier), V, gp and points of chunk ci + 1 are in flight while the pairs of chunk ci are multiplied.
template <int G, bool MF>
__global__ __launch_bounds__(64, 2) void k_schur_staged(DeviceProblem d, double lam, double min_diag, int cpb,
                                                        const double* __restrict__ poses, const double* __restrict__ pts,
                                                        int n_fold) {
  __shared__ double stage[64 * FLUSH_STRIDE];
  __shared__ __align__(16) double sZ[SCH_CAP * SZ_STRIDE];  // Z = W L per observation, Vd^-1 = L L^T
  __shared__ double sVi[SCH_MAXLM * 9];  // L (6: l00 l10 l20 l11 l21 l22) | L^T gp (3) per landmark of the chunk
  __shared__ int sFree[SCH_CAP];
  __shared__ int sKb[SCH_MAXLM + 1];
  __shared__ int sTab[3][SCH_MAXCPB + 1];  // this block's chunks: first landmark | first observation | flush bits
  __shared__ double sPts[MF ? SCH_MAXLM * 3 : 1];
  // the last n_fold blocks of the grid fold the linearisation's U | gc rows into the slabs that the gather sums
  // (fold_u_rows: short, memory-bound waves that fill the tail of the launch, where the Schur blocks run out; no launch
  // of their own.  As the FIRST blocks of the grid they delayed the Schur blocks: 57.8 against 50.0 us on BASELINE configs[2])
  const int n_schur = (int)gridDim.x - n_fold;
  if ((int)blockIdx.x >= n_schur) {
    if (lam < 0.0 && (d.ctrl->done || !d.ctrl->need_lin)) return;  // a re-solve with another damping: U | gc are kept
    fold_u_rows(d, blockIdx.x - n_schur, threadIdx.x);
    return;
  }
  const int sblk = blockIdx.x;  // this block's number among the Schur blocks
  if (MF && lam < 0.0) {
    poses = d.poses_buf[d.ctrl->sel];
    pts = d.pts_buf[d.ctrl->sel];
  }
  const bool per_comp = lam < 0.0 && d.clm.lam_lm != nullptr;  // solver mode with a damping per component
  if (lam < 0.0) {  // solver mode
    if (d.ctrl->done) return;
    lam = d.ctrl->lam;
  }
  const int c0 = sblk * cpb, c1 = min(d.n_chunks, c0 + cpb);
  if (c0 >= c1) return;
  const int lane = threadIdx.x;  // one wave per block
  if (lane <= c1 - c0) {
    sTab[0][lane] = d.sch_lm[c0 + lane];
    sTab[1][lane] = d.sch_k[c0 + lane];
    sTab[2][lane] = lane < c1 - c0 ? (int)d.sch_flush[c0 + lane] : 0;
  }
  const int gl = lane & (G - 1), grp = lane / G;
  constexpr int GPW = 64 / G;
  const long n = d.n, L = d.L;
  double acc[36], accr[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) accr[i] = 0.0;
  bool cur_sym = false;
  int nbad = 0;
  double* __restrict__ out = d.contrib + (long)d.sch_blk_row[sblk] * CTR_ROW;
  __syncthreads();

  auto flush = [&]() {  // wave-uniform: every lane calls it
#pragma unroll
    for (int q = 0; q < CTR_ROW / FLUSH_NV; ++q) {
#pragma unroll
      for (int e = 0; e < FLUSH_NV; ++e) {
        const int t = FLUSH_NV * q + e;
        double v;
        if (t < 36) {
          const int i = t / 6, c = t - 6 * i;
          // two observations of one landmark in the same image: block + block^T
          v = cur_sym ? acc[6 * i + c] + acc[6 * c + i] : acc[t];
        } else {
          v = accr[t - 36];
        }
        stage[lane * FLUSH_STRIDE + e] = -v;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < FLUSH_NV; ++k) {  // 64 rows x FLUSH_NV entries, consecutive lanes -> consecutive addresses
        const int idx = lane + 64 * k, row = idx / FLUSH_NV, e = idx - FLUSH_NV * row;
        out[row * CTR_ROW + FLUSH_NV * q + e] = stage[row * FLUSH_STRIDE + e];
      }
      __syncthreads();
    }
    out += 64 * CTR_ROW;
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) accr[i] = 0.0;
    cur_sym = false;
  };

  // pair owned by this lane: p = gl = b (b + 1) / 2 + a, a <= b (static per lane)
  int b0 = (int)((sqrt(8.0 * gl + 1.0) - 1.0) * 0.5);
  while ((b0 + 1) * (b0 + 2) / 2 <= gl) ++b0;
  while (b0 * (b0 + 1) / 2 > gl) --b0;
  const int a0 = gl - b0 * (b0 + 1) / 2;

  // registers of the chunk in flight (raw loaded values only: arithmetic on them would make the
  // compiler wait for the loads right here)
  double pw[MF ? 7 : W_ROWS], pv[6], pg[3], plam = 0.0;  // MF: pw is the pose of the observation's camera
  double pP[3] = {0.0, 0.0, 1.0}, pu = 0.0, pvv = 0.0;   // MF: the landmark's point (landmark lanes), the pixel
  int a_cam = 0, a_free = -1, a_lm = 0;                  // MF, stage A: the record of chunk ci + 2
  double a_u = 0.0, a_v = 0.0;
  int pfree = -1, plm = 0, pkb = 0, p_j0 = 0, p_k0 = 0, p_nobs = 0, p_nc = 0;
  auto issue_records = [&](int ci) {  // MF stage A
    const int k0 = sTab[1][ci - c0], k1 = sTab[1][ci - c0 + 1];
    if (lane < k1 - k0) {
      a_cam = d.o_cam[k0 + lane]; a_free = d.o_free[k0 + lane]; a_lm = d.o_lm[k0 + lane];
      a_u = d.o_u[k0 + lane]; a_v = d.o_v[k0 + lane];
    }
  };
  auto issue = [&](int ci) {
    const int j0 = sTab[0][ci - c0], j1 = sTab[0][ci - c0 + 1], k0 = sTab[1][ci - c0], k1 = sTab[1][ci - c0 + 1];
    p_j0 = j0;
    p_k0 = k0;
    p_nobs = k1 - k0;
    p_nc = j1 - j0;
    if (lane < p_nobs) {
      if constexpr (MF) {  // stage B: the records of this chunk are in the stage-A registers
#pragma unroll
        for (int r = 0; r < 7; ++r) pw[r] = poses[7 * a_cam + r];
        pfree = a_free; plm = a_lm; pu = a_u; pvv = a_v;
      } else {
#pragma unroll
        for (int r = 0; r < W_ROWS; ++r) pw[r] = d.W[(long)r * n + k0 + lane];
        pfree = d.o_free[k0 + lane];
        plm = d.o_lm[k0 + lane];
      }
    }
    if (lane < p_nc) {
      if constexpr (MF) {
#pragma unroll
        for (int i = 0; i < 3; ++i) pP[i] = pts[3 * (long)(j0 + lane) + i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) pv[i] = d.V[(long)i * L + j0 + lane];
#pragma unroll
      for (int i = 0; i < 3; ++i) pg[i] = d.gp[(long)i * L + j0 + lane];
      if (per_comp) plam = d.clm.lam_lm[j0 + lane];
    }
    if (lane <= p_nc) pkb = d.lm_ptr[j0 + lane];
  };
#pragma unroll
  for (int r = 0; r < (MF ? 7 : W_ROWS); ++r) pw[r] = MF && r == 3 ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) pv[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) pg[i] = 0.0;
  if constexpr (MF) issue_records(c0);
  issue(c0);
  if constexpr (MF) {
    if (c0 + 1 < c1) issue_records(c0 + 1);
  }
  for (int ci = c0; ci < c1; ++ci) {
    const int nobs = p_nobs, nc = p_nc, j0 = p_j0, k0 = p_k0;
    const unsigned fbits = (unsigned)sTab[2][ci - c0];
    __syncthreads();  // the pairs of the previous chunk are done with the LDS tiles
    if (lane < nc) {
      // Vd^-1 = L L^T (Cholesky of the 3x3 inverse): the pair products are then Z_a Z_b^T with ONE 6x3 factor
      // Z = W L per observation, and the reduced right-hand side is Z (L^T gp)
      double vi[6];
      const double det = damped_inv3(pv, per_comp ? plam : lam, min_diag, vi);
      const double d00 = vi[0], il00 = 1.0 / sqrt(d00);
      const double l10 = vi[1] * il00, l20 = vi[2] * il00;
      const double d11 = vi[3] - l10 * l10, il11 = 1.0 / sqrt(d11);
      const double l21 = (vi[4] - l20 * l10) * il11;
      const double d22 = vi[5] - l20 * l20 - l21 * l21;
      const bool ok = det > 0.0 && d00 > 0.0 && d11 > 0.0 && d22 > 0.0;
      if (!ok) ++nbad;  // the landmark then contributes nothing (finite S), and the step is refused by k_lm_decide
      const double L6[6] = {ok ? d00 * il00 : 0.0, ok ? l10 : 0.0, ok ? l20 : 0.0, ok ? d11 * il11 : 0.0, ok ? l21 : 0.0,
                            ok ? sqrt(d22) : 0.0};
#pragma unroll
      for (int i = 0; i < 6; ++i) sVi[lane * 9 + i] = L6[i];
      sVi[lane * 9 + 6] = L6[0] * pg[0] + L6[1] * pg[1] + L6[2] * pg[2];
      sVi[lane * 9 + 7] = L6[3] * pg[1] + L6[4] * pg[2];
      sVi[lane * 9 + 8] = L6[5] * pg[2];
      if constexpr (MF) {
#pragma unroll
        for (int i = 0; i < 3; ++i) sPts[lane * 3 + i] = pP[i];
      }
    }
    if (lane <= nc) sKb[lane] = pkb - k0;
    if (lane < nobs) sFree[lane] = pfree;
    __syncthreads();
    if (lane < nobs) {
      const int li = plm - j0;
      double jw[W_ROWS];
      if constexpr (MF) {
        const double P[3] = {sPts[li * 3], sPts[li * 3 + 1], sPts[li * 3 + 2]};
        double r[2], Jc[12], Jp[6], zc;
        linearize_obs(d.K, pw, P, pu, pvv, r, Jc, Jp, zc);
        if (d.loss.kind != LOSS_LINEAR) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            double sc;
            (void)robust_scale(d.loss, r[i], sc);
#pragma unroll
            for (int q = 0; q < 6; ++q) Jc[6 * i + q] *= sc;
#pragma unroll
            for (int q = 0; q < 3; ++q) Jp[3 * i + q] *= sc;
          }
        }
        w_pack(Jc, Jp, pfree >= 0, jw);
      } else {
#pragma unroll
        for (int i = 0; i < W_ROWS; ++i) jw[i] = pw[i];  // the stored factors
      }
      // Z = W L = Jc^T (Jp L), Jc = [-Jp | Jr]: 12 + 36 multiply-adds, W itself is never formed
      const double l00 = sVi[li * 9], l10 = sVi[li * 9 + 1], l20 = sVi[li * 9 + 2], l11 = sVi[li * 9 + 3],
                   l21 = sVi[li * 9 + 4], l22 = sVi[li * 9 + 5];
      const double q0[3] = {jw[0] * l00 + jw[1] * l10 + jw[2] * l20, jw[1] * l11 + jw[2] * l21, jw[2] * l22};
      const double q1[3] = {jw[3] * l00 + jw[4] * l10 + jw[5] * l20, jw[4] * l11 + jw[5] * l21, jw[5] * l22};
      double* __restrict__ zrow = &sZ[lane * SZ_STRIDE];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          zrow[3 * i + c] = -(jw[i] * q0[c] + jw[3 + i] * q1[c]);
          zrow[9 + 3 * i + c] = jw[6 + i] * q0[c] + jw[9 + i] * q1[c];
        }
    }
    __syncthreads();
    if (ci + 1 < c1) issue(ci + 1);  // in flight while the pairs below are multiplied
    if constexpr (MF) {
      if (ci + 2 < c1) issue_records(ci + 2);
    }
    for (int l0 = 0, step = 0; l0 < nc; l0 += GPW, ++step) {
      if ((fbits >> step) & 1u) flush();
      const int li = l0 + grp;
      int kb = 0, npairs = 0;
      double g[3] = {0, 0, 0};
      if (li < nc) {
        kb = sKb[li];
        const int len = sKb[li + 1] - kb;
        npairs = len * (len + 1) / 2;
#pragma unroll
        for (int i = 0; i < 3; ++i) g[i] = sVi[li * 9 + 6 + i];
      }
      bool act = gl < npairs;
      const int fa = act ? sFree[kb + a0] : -1, fb = act ? sFree[kb + b0] : -1;
      act = act && fa >= 0 && fb >= 0;
      if (act) {
        cur_sym = (fa == fb) && (a0 != b0);
        const double* __restrict__ Ya = &sZ[(kb + a0) * SZ_STRIDE];
        const double* __restrict__ Wb = &sZ[(kb + b0) * SZ_STRIDE];
        double Y[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) Y[i] = Ya[i];
        if (a0 == b0) {
#pragma unroll
          for (int i = 0; i < 6; ++i) accr[i] = fma(Y[3 * i + 2], g[2], fma(Y[3 * i + 1], g[1], fma(Y[3 * i], g[0], accr[i])));
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const double w0 = Wb[3 * c], w1 = Wb[3 * c + 1], w2 = Wb[3 * c + 2];
#pragma unroll
          for (int i = 0; i < 6; ++i)  // three chained FMAs into the accumulator (a sum formed first costs a fourth instruction)
            acc[6 * i + c] = fma(Y[3 * i + 2], w2, fma(Y[3 * i + 1], w1, fma(Y[3 * i], w0, acc[6 * i + c])));
        }
      }
    }
  }
  flush();
  if (nbad) atomic_add_f64(&SCAL_MINE(d)[SC_BADV], (double)nbad);
}

// (eight waves per S block since late round 3: the launch lasts as long as its heaviest block, a diagonal one, whose rows
// are twice the average; with four waves config 2 took 13.4 us)
constexpr int GATHER_WAVES = 8, GATHER_THREADS = 64 * GATHER_WAVES;
// k_gather_S (staged path): S (upper blocks) = sum of the contribution rows that the inverted index
// lists for each block (+ partial U on the diagonal blocks), rhs = partial gc + the rows' rhs part,
// Ugc = partial U/gc (the diagonal blocks also sum their camera's k_linearize slabs: no k_reduce_U
// launch on this path).  One 512-thread block per S block: wave w takes rows w, w + 8, ... of the
// list (lanes 0..41 = the 42 entries of a row, 16 rows in flight per wave), the eight partial sums are
// combined in a fixed order.  Reads every contribution row once: 336 B per (pair, run, block of
// k_schur_staged).
__global__ __launch_bounds__(GATHER_THREADS) void k_gather_S(DeviceProblem d, int solver_mode, double lam, double min_diag,
                                                  int precond, double* __restrict__ csum) {
  __shared__ double part[GATHER_WAVES][64];
  __shared__ double sblk[36];
  __shared__ double upart[8][32];
  __shared__ double ugc[32];
  int from_slabs = 1;  // U / gc were accumulated by a k_linearize since the last gather
  if (solver_mode) {
    if (d.ctrl->done) return;
    from_slabs = d.ctrl->need_lin;
    lam = d.ctrl->lam;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool extra = (int)blockIdx.x >= d.nu;  // blocks past nu: U / gc of the fixed cameras (exported, never solved for)
  const int s = extra ? 0 : blockIdx.x;
  const bool diag = extra || d.low_of[s] < 0;
  const int f = d.urow[s];
  const long cam27 = (long)(extra ? d.fixed_cams[blockIdx.x - d.nu] : d.free2cam[f]) * 27;
  if (diag && threadIdx.x < 256) {  // (eight groups of 32 threads, as when the workgroup had 256)
    // this camera's U (21) | gc (6): sum of the k_linearize slabs (k_reduce_U's job, fused here), or
    // the values kept from the last linearisation when this is a re-solve with another damping
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    double u = 0.0;
    if (e < 27) {
      if (from_slabs) {
        const long tot = (long)d.C * 27;
        // four slabs of this group per round: the loads are issued before the stores that clear them (the compiler
        // cannot move a load of the array over a store to it: one round trip per slab otherwise)
        for (int k = g; k < d.nslab_u; k += 32) {
          double* p0 = d.Ugc_slab + (long)k * tot + cam27 + e;
          const bool h1 = k + 8 < d.nslab_u, h2 = k + 16 < d.nslab_u, h3 = k + 24 < d.nslab_u;
          const double a0 = p0[0], a1 = h1 ? p0[8 * tot] : 0.0, a2 = h2 ? p0[16 * tot] : 0.0, a3 = h3 ? p0[24 * tot] : 0.0;
          p0[0] = 0.0;  // consumed: clean for the next linearisation (no memset per iteration)
          if (h1) p0[8 * tot] = 0.0;
          if (h2) p0[16 * tot] = 0.0;
          if (h3) p0[24 * tot] = 0.0;
          u += a0; u += a1; u += a2; u += a3;
        }
      } else if (g == 0) {
        u = d.Ugc_part[cam27 + e];
      }
    }
    upart[g][e] = u;
  }
  const int r0 = d.ctr_ptr[s], r1 = extra ? r0 : d.ctr_ptr[s + 1];
  double acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.0;
  // wave w takes rows r0 + w + GATHER_WAVES k.  Lane l first fetches the row number of k = kb + l (one load for 64
  // rows), then the rows are read 16 at a time: no contribution load waits for an index load.
  const int cnt = (r1 - r0 - w + GATHER_WAVES - 1) / GATHER_WAVES;  // rows of this wave
  for (int kb = 0; kb < cnt; kb += 64) {
    const int mine = (kb + lane < cnt) ? d.ctr_row[r0 + w + GATHER_WAVES * (kb + lane)] : 0;
    const int m = min(64, cnt - kb);
    for (int k0 = 0; k0 < m; k0 += 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const long q = __builtin_amdgcn_readlane(mine, (k0 + u) & 63);
        v[u] = (k0 + u < m && lane < CTR_ROW) ? d.contrib[q * CTR_ROW + lane] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 7] += v[u];
    }
  }
  part[w][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (diag && threadIdx.x < 27) {
    double u = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) u += upart[g][threadIdx.x];
    ugc[threadIdx.x] = u;
    d.Ugc_part[cam27 + threadIdx.x] = u;
    d.Ugc[cam27 + threadIdx.x] = u;  // in the buffer the multi-GPU all-reduce sums
  }
  __syncthreads();
  if (extra || w != 0 || lane >= (diag ? CTR_ROW : 36)) return;
  static_assert(GATHER_WAVES == 8, "the fold below is written out for eight waves");
  double sum = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) +
               ((part[4][lane] + part[5][lane]) + (part[6][lane] + part[7][lane]));
  if (lane < 36) {
    if (diag) {
      const int i = lane / 6, c = lane - 6 * i;
      sum += ugc[i <= c ? sym6_index(i, c) : sym6_index(c, i)];
    }
    d.S[(long)s * 36 + lane] = sum;
    if (diag && precond) sblk[lane] = sum;
    if (csum) {  // the damped block into the coarse sums P^T (S + lam D) P of ba_coarse.h: both triangles
      const int r = lane / 6, c = lane - 6 * r, a = f / d.c_ag, b = d.ucol[s] / d.c_ag;
      const double v = (diag && r == c) ? sum + lam * fmax(ugc[sym6_index(r, r)], min_diag) : sum;
      atomic_add_f64(&csum[(6 * a + r) * d.c_nc + 6 * b + c], v);
      if (!diag) atomic_add_f64(&csum[(6 * b + c) * d.c_nc + 6 * a + r], v);
    }
  } else {
    d.rhs[6L * f + lane - 36] = sum + ugc[21 + lane - 36];
  }
  // single GPU: the block-Jacobi preconditioner of this camera, Minv = (S_ff + lam max(diag U_f, min_diag))^-1,
  // right here (k_block_precond's job; with several GPUs S is only final after the all-reduce)
  if (diag && precond) {
    __builtin_amdgcn_wave_barrier();  // wave 0 only: LDS keeps one wave's accesses in order
    if (lane == 0) {
      double A[36];
#pragma unroll
      for (int i = 0; i < 36; ++i) A[i] = sblk[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) A[7 * i] += lam * fmax(ugc[sym6_index(i, i)], min_diag);
      if (!spd_inverse6(A)) atomic_add_f64(&SCAL_MINE(d)[SC_BADV], 1.0);
#pragma unroll
      for (int i = 0; i < 36; ++i) d.Minv[(long)f * 36 + i] = A[i];
    }
  }
}

// k_gather_S_wave: the same sums with ONE WAVE per S block (four blocks of S per workgroup, no barrier, no cross-wave
// fold), for systems whose S blocks collect few contribution rows each: at BASELINE configs[3] an S block has 21 rows
// of 42 doubles, and a 256-thread workgroup per block spends its time on index loads, two barriers and the fold.
// Fixed summation order (rows in index order into eight partial sums): S stays bitwise reproducible.  No fused
// preconditioner (the one-workgroup PCG's problems are small and keep k_gather_S).
// `bcr` (solver mode, one GPU, no prior: S is final here): the damped blocks and -rhs also go straight into the level-0
// arrays of the cyclic reduction (what k_bcr_assemble does from S in a launch of its own: ba_bcr.h has the layout).
struct BcrScatter {
  double* B;                  // null: off
  const long *dst, *dst2, *rdst;
  const unsigned char* kind;
  int m;
  double min_diag;
};

__global__ __launch_bounds__(256) void k_gather_S_wave(DeviceProblem d, int solver_mode, BcrScatter bcr) {
  int from_slabs = 1;
  double lam = 0.0;
  if (solver_mode) {
    if (d.ctrl->done) return;
    from_slabs = d.ctrl->need_lin;
    lam = d.ctrl->lam;
  }
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= d.nu + (d.C - d.Cf)) return;
  const bool extra = idx >= d.nu;
  const int s = extra ? 0 : idx;
  const bool diag = extra || d.low_of[s] < 0;
  const int f = d.urow[s];
  const long cam27 = (long)(extra ? d.fixed_cams[idx - d.nu] : d.free2cam[f]) * 27;
  const int r0 = d.ctr_ptr[s], cnt = extra ? 0 : d.ctr_ptr[s + 1] - r0;
  double u = 0.0;  // lanes 0..26 of a diagonal block: this camera's U (21) | gc (6)
  if (diag && lane < 27) {
    if (from_slabs) {
      const long tot = (long)d.C * 27;
      double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
      for (int k = 0; k < d.nslab_u; k += 4) {  // (nslab_u is 1 or a multiple of 4 up to 32; guard the tail anyway)
        double* p0 = d.Ugc_slab + (long)k * tot + cam27 + lane;
        const double a0 = p0[0], a1 = k + 1 < d.nslab_u ? p0[tot] : 0.0, a2 = k + 2 < d.nslab_u ? p0[2 * tot] : 0.0,
                     a3 = k + 3 < d.nslab_u ? p0[3 * tot] : 0.0;
        p0[0] = 0.0;  // consumed: clean for the next linearisation
        if (k + 1 < d.nslab_u) p0[tot] = 0.0;
        if (k + 2 < d.nslab_u) p0[2 * tot] = 0.0;
        if (k + 3 < d.nslab_u) p0[3 * tot] = 0.0;
        u0 += a0; u1 += a1; u2 += a2; u3 += a3;
      }
      u = (u0 + u1) + (u2 + u3);
    } else {
      u = d.Ugc_part[cam27 + lane];
    }
    d.Ugc_part[cam27 + lane] = u;
    d.Ugc[cam27 + lane] = u;  // in the buffer the multi-GPU all-reduce sums
  }
  double acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = 0.0;
  for (int kb = 0; kb < cnt; kb += 64) {
    const int mine = (kb + lane < cnt) ? d.ctr_row[r0 + kb + lane] : 0;
    const int m = min(64, cnt - kb);
    for (int k0 = 0; k0 < m; k0 += 16) {
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const long row = __builtin_amdgcn_readlane(mine, (k0 + q) & 63);
        v[q] = (k0 + q < m && lane < CTR_ROW) ? d.contrib[row * CTR_ROW + lane] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q & 7] += v[q];
    }
  }
  double sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  // the diagonal blocks add U (entry (i, c) of the symmetric 6x6 sits in lane sym6_index) and the rhs adds gc
  int src = 0;
  if (lane < 36) { const int i = lane / 6, c = lane - 6 * i; src = i <= c ? sym6_index(i, c) : sym6_index(c, i); }
  else if (lane < CTR_ROW) src = 21 + lane - 36;
  const double uu = __shfl(u, src);
  if (extra || lane >= (diag ? CTR_ROW : 36)) return;
  if (lane < 36) d.S[(long)s * 36 + lane] = diag ? sum + uu : sum;
  else d.rhs[6L * f + lane - 36] = sum + uu;
  if (bcr.B) {
    if (lane < 36) {
      const int r = lane / 6, c = lane - 6 * r, k = bcr.kind[s], m = bcr.m;
      double v = diag ? sum + uu : sum;
      double* blk = bcr.B + bcr.dst[s];
      if (k == 0) {  // a camera's diagonal block: uu of a diagonal entry is U's diagonal entry (Marquardt scaling)
        if (c <= r) {
          if (r == c) v += lam * fmax(uu, bcr.min_diag);
          blk[r * m + c] = v;
          blk[c * m + r] = v;
        }
      } else if (k == 1) {
        blk[r * m + c] = v;
        bcr.B[bcr.dst2[s] + c * m + r] = v;
      } else if (k == 2) {
        blk[r * m + c] = v;
      } else {
        blk[c * m + r] = v;
      }
    } else {
      bcr.B[bcr.rdst[f] + lane - 36] = -(sum + uu);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_damp_mirror: (after the optional all-reduce) add lam * max(diag U, min_diag) to the diagonal of S,
// invert the diagonal blocks into the block-Jacobi preconditioner, copy upper off-diagonal blocks,
// transposed, into their lower mirror slots.  One thread per upper slot.
__global__ __launch_bounds__(256) void k_damp_mirror(DeviceProblem d, double lam, double min_diag) {
  if (lam < 0.0) {
    if (d.ctrl->done) return;
    lam = d.ctrl->lam;
  }
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.nu) return;
  double* Sb = d.S + (long)s * 36;
  const int low = d.low_of[s];
  if (low >= 0) {
    double* Sl = d.S + (long)low * 36;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int c = 0; c < 6; ++c) Sl[6 * c + i] = Sb[6 * i + c];
  } else {
    const int f = d.urow[s];
    double A[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) A[i] = Sb[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) A[7 * i] += lam * fmax(u_diag(d, f, i), min_diag);
#pragma unroll
    for (int i = 0; i < 36; ++i) Sb[i] = A[i];
    if (!spd_inverse6(A)) atomic_add_f64(&d.scal[SC_BADV], 1.0);
#pragma unroll
    for (int i = 0; i < 36; ++i) d.Minv[(long)f * 36 + i] = A[i];
  }
}

// ------------------------------------------------------------------------------------------------
// Block-Jacobi PCG on S dc = -rhs.  Vectors are [Cf][6]; one thread per scalar row.
// All step lengths are computed on the device from reduced scalars; the host only polls SC_DONE.
__global__ __launch_bounds__(256) void k_pcg_init(DeviceProblem d, int solver_mode) {
  __shared__ double red[4];
  if (solver_mode && d.ctrl->done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) d.scal[SC_DONE] = 1.0;
    return;
  }
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double rz = 0.0;
  if (t < 6 * d.Cf) {
    const int f = t / 6, i = t - 6 * f;
    double zi = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) zi += d.Minv[(long)f * 36 + 6 * i + c] * (-d.rhs[6 * f + c]);
    const double ri = -d.rhs[t];
    d.dc[t] = 0.0;
    d.r[t] = ri;
    d.z[t] = zi;
    d.p[t] = zi;
    rz = ri * zi;
  }
  const double b = block_sum_256(rz, red);
  if (threadIdx.x == 0) {
    atomic_add_f64(&d.scal[SC_RZ], b);
    atomic_add_f64(&d.scal[SC_RZ0], b);
  }
}

__global__ __launch_bounds__(256) void k_pcg_spmv(DeviceProblem d) {
  __shared__ double red[4];
  if (d.scal[SC_DONE] != 0.0) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double pq = 0.0;
  if (t < 6 * d.Cf) {
    const int f = t / 6, i = t - 6 * f;
    double s = 0.0;
    for (int e = d.brow_ptr[f]; e < d.brow_ptr[f + 1]; ++e) {
      const double* B = d.S + (long)d.bslot[e] * 36 + 6 * i;
      const double* x = d.p + 6 * d.bcol[e];
      s += B[0] * x[0] + B[1] * x[1] + B[2] * x[2] + B[3] * x[3] + B[4] * x[4] + B[5] * x[5];
    }
    d.q[t] = s;
    pq = s * d.p[t];
  }
  const double b = block_sum_256(pq, red);
  if (threadIdx.x == 0) atomic_add_f64(&d.scal[SC_PQ], b);
}

__global__ __launch_bounds__(256) void k_pcg_update(DeviceProblem d) {
  __shared__ double red[4];
  if (d.scal[SC_DONE] != 0.0) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const double pq = d.scal[SC_PQ];
  const double alpha = pq > 0.0 ? d.scal[SC_RZ] / pq : 0.0;
  double rz = 0.0;
  if (t < 6 * d.Cf) {
    const int f = t / 6, i = t - 6 * f;
    double rn[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) rn[c] = d.r[6 * f + c] - alpha * d.q[6 * f + c];
    double zi = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) zi += d.Minv[(long)f * 36 + 6 * i + c] * rn[c];
    d.dc[t] += alpha * d.p[t];
    d.z[t] = zi;
    rz = rn[i] * zi;
    // r is rewritten by k_pcg_dir (all threads of this kernel still read the old r)
  }
  const double b = block_sum_256(rz, red);
  if (threadIdx.x == 0) atomic_add_f64(&d.scal[SC_RZNEW], b);
}

__global__ __launch_bounds__(256) void k_pcg_dir(DeviceProblem d, double tol2, int max_iter) {
  if (d.scal[SC_DONE] != 0.0) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const double pq = d.scal[SC_PQ];
  const double rz = d.scal[SC_RZ], rzn = d.scal[SC_RZNEW];
  const double alpha = pq > 0.0 ? rz / pq : 0.0;
  const double beta = rz > 0.0 ? rzn / rz : 0.0;
  if (t < 6 * d.Cf) {
    d.r[t] -= alpha * d.q[t];
    d.p[t] = d.z[t] + beta * d.p[t];
  }
  // the scalar roll-over is done by the LAST block to get here, so nobody still reads the old values
  __shared__ int last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned int ticket = atomicAdd(reinterpret_cast<unsigned int*>(&d.scal[SC_COUNT - 1]), 1u);
    last = (ticket == gridDim.x - 1);
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    const double it = d.scal[SC_PCGIT] + 1.0;
    d.scal[SC_PCGIT] = it;
    d.scal[SC_RZ] = rzn;
    d.scal[SC_RZNEW] = 0.0;
    d.scal[SC_PQ] = 0.0;
    *reinterpret_cast<unsigned int*>(&d.scal[SC_COUNT - 1]) = 0u;
    if (!(rzn > tol2 * d.scal[SC_RZ0]) || !(pq > 0.0) || it >= (double)max_iter) d.scal[SC_DONE] = 1.0;
  }
}

// ------------------------------------------------------------------------------------------------
// k_solve_small: the whole reduced solve for a small camera system in ONE workgroup and ONE launch:
// LM damping of the diagonal and the complete block-Jacobi PCG loop, S in registers (each thread
// owns up to NB upper 6x6 blocks), vectors and preconditioner in LDS: an iteration touches no global
// memory.  Built for latency, the only bound here (S is read once, 288 B per block):
//   * Chronopoulos-Gear form of PCG: ONE fused reduction (r.z and z.Sz) and 3 barriers per iteration
//     instead of two reductions and 4 barriers (same iterates in exact arithmetic);
//   * SpMV by symmetry without atomics: thread (i, j) stores B z_j and B^T z_i into its own LDS
//     slot, row threads then gather the <= (blocks per row) slots listed by sm_gidx in a fixed order
//     (ds_add_f64 costs ~20 cycles per wave-instruction and made the sum order-dependent);
//   * wave sums by DPP, not ds_bpermute shuffles;
//   * the 6 entries of a camera sit in one wave (60 rows per wave), so z = Minv r exchanges r through
//     LDS without a block barrier.
// k_block_precond: Minv_f = (S_ff + lam * max(diag U_f, min_diag))^-1, one thread per free camera.
__global__ __launch_bounds__(64) void k_block_precond(DeviceProblem d, double lam, double min_diag) {
  if (lam < 0.0) {
    if (d.ctrl->done) return;
    lam = d.ctrl->lam;
  }
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= d.Cf) return;
  double A[36];
  const double* Sd = d.S + (long)d.diag_slot[f] * 36;
#pragma unroll
  for (int i = 0; i < 36; ++i) A[i] = Sd[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) A[7 * i] += lam * fmax(u_diag(d, f, i), min_diag);
  if (!spd_inverse6(A)) atomic_add_f64(&d.scal[SC_BADV], 1.0);
#pragma unroll
  for (int i = 0; i < 36; ++i) d.Minv[(long)f * 36 + i] = A[i];
}

constexpr int SMALL_THREADS = 512;
constexpr int SMALL_VPT = 2;             // vector entries per thread
constexpr int SMALL_ROWS_PER_WAVE = 60;  // 10 cameras per wave
constexpr int SMALL_MAX_ROWS = SMALL_VPT * (SMALL_THREADS / 64) * SMALL_ROWS_PER_WAVE;  // 960 = 160 free cameras
constexpr int SMALL_MAX_NB = 2;          // upper blocks per thread: nu <= 1024
constexpr int SMALL_GREG = 20;           // gather offsets a row thread keeps in registers
constexpr int SMALL_ZSTRIDE = 7;         // doubles per camera of z and r in LDS: 6 | pad (a stride of 6 doubles makes the 64-bit reads of 32 lanes share banks two by two)
constexpr int SMALL_CSTRIDE = 13;        // doubles per SpMV slot: B z_j (6) | B^T z_i (6) | pad (bank spread)
constexpr int SMALL_COARSE_MAX = 64;     // == COARSE_MAX (ba_coarse.h): 8 threads per row fill the workgroup
constexpr int SMALL_COARSE_PARTS = 8;    // threads per row of the coarse inverse
constexpr int SMALL_COARSE_K = SMALL_COARSE_MAX / SMALL_COARSE_PARTS;  // its entries per thread

__device__ __forceinline__ void small_load_block(const DeviceProblem& d, int t, double lam, double min_diag,
                                                 double B[36], int& i, int& j) {
  const int4 ds = reinterpret_cast<const int4*>(d.sm_desc)[t];
  const int s = ds.x;
  i = ds.y;
  j = ds.z;
#pragma unroll
  for (int e = 0; e < 36; ++e) B[e] = d.S[(long)s * 36 + e];
  if (i == j) {
#pragma unroll
    for (int e = 0; e < 6; ++e) B[7 * e] += lam * fmax(d.Ugc[ds.w + sym6_index(e, e)], min_diag);
  }
}

// One stored upper block B = S_ij, both directions of the symmetric product into the block's own slot
// Returns the block's share of z^T S z: z_i^T B z_j, twice for an off-diagonal block.
__device__ __forceinline__ double small_block_products(const double B[36], int i, int j, const double* z, double* out) {
  double zj[6], zi[6], q = 0.0;
#pragma unroll
  for (int c = 0; c < 6; ++c) { zj[c] = z[SMALL_ZSTRIDE * j + c]; zi[c] = z[SMALL_ZSTRIDE * i + c]; }
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) a += B[6 * r + c] * zj[c];
    out[r] = a;
    q += zi[r] * a;
  }
  if (i != j) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double a = 0.0;
#pragma unroll
      for (int r = 0; r < 6; ++r) a += B[6 * r + c] * zi[r];
      out[6 + c] = a;
    }
    q += q;
  }
  return q;
}

// COARSE: z = Minv r + P Sc^-1 P^T r with the aggregation coarse space of ba_coarse.h (cinv: Sc^-1 of the previous LM
// iteration, [nc][nc] + a validity flag; null: block-Jacobi alone).  The restriction P^T r is wave-local (an aggregate
// is 5 or 10 of the 10 cameras of a wave), the nc x nc product is spread over 8 nc threads with the matrix in
// registers, the prolongation is a read of the aggregate's 6-vector: two more barriers per iteration.
// -DSMALL_CLOCKS: wall_clock64 cuts (10 ns ticks) inside k_solve_small, printed by thread 0 (a debug build, never the
// library that ships): set-up, then for the first PCG iterations the block products up to the barrier, the scalars
// and the gather of w, the vector updates and the preconditioner up to the second barrier
#ifdef SMALL_CLOCKS
#define SMALL_CUT(i) do { if (threadIdx.x == 0 && (i) < 32) small_cut[i] = wall_clock64(); } while (0)
#else
#define SMALL_CUT(i) do {} while (0)
#endif
template <int NB, bool COARSE>
__global__ __launch_bounds__(SMALL_THREADS) void k_solve_small(DeviceProblem d, double lam, double min_diag,
                                                               double tol2, int max_iter, const double* __restrict__ cinv) {
  extern __shared__ double sm[];
  if (lam < 0.0) {
    if (d.ctrl->done) return;
    lam = d.ctrl->lam;
  }
  const int N = 6 * d.Cf;
  double* zv = sm;                              // [Cf][7]  z = Minv r: the vector the SpMV multiplies (6 | pad)
  double* rv = zv + SMALL_ZSTRIDE * d.Cf;       // [Cf][7]  r, for the wave-local exchange
  double* red = rv + SMALL_ZSTRIDE * d.Cf;      // [8][2]
  double* rcv = red + 16;                       // [SMALL_COARSE_MAX]  P^T r
  double* zcv = rcv + SMALL_COARSE_MAX;         // [SMALL_COARSE_MAX]  Sc^-1 P^T r
  double* slot = zcv + SMALL_COARSE_MAX;        // [nu][SMALL_CSTRIDE]
  int* gl = reinterpret_cast<int*>(slot + (size_t)(d.nu + 1) * SMALL_CSTRIDE);  // [nb] gather list
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nc = d.c_nc, ag = d.c_ag;
  const bool coarse = COARSE && cinv != nullptr && cinv[nc * nc] > 0.5;  // uniform
  double cw[COARSE ? SMALL_COARSE_K : 1];  // row tid / 8 of Sc^-1, columns tid % 8 + 8 k
  if (COARSE) {
#pragma unroll
    for (int k = 0; k < SMALL_COARSE_K; ++k) {
      const int b = (tid & (SMALL_COARSE_PARTS - 1)) + SMALL_COARSE_PARTS * k;
      cw[k] = (coarse && tid < SMALL_COARSE_PARTS * nc && b < nc) ? cinv[(tid / SMALL_COARSE_PARTS) * nc + b] : 0.0;
    }
  }

  for (int t = tid; t < d.nb; t += SMALL_THREADS) gl[t] = d.sm_gidx[t];
  if (tid < SMALL_CSTRIDE) slot[(size_t)d.nu * SMALL_CSTRIDE + tid] = 0.0;
  double B[NB][36];
  int ij[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int t = tid + k * SMALL_THREADS;
    ij[k] = -1;
    if (t < d.nu) {
      int i, j;
      small_load_block(d, t, lam, min_diag, B[k], i, j);
      ij[k] = (i << 16) | j;
    }
  }
  // rows of this thread: wave wv (+ 8 per extra entry), lanes 0..59
  double mrow[SMALL_VPT][6];  // the preconditioner row of each entry (registers: no LDS copy of Minv)
  int row6[SMALL_VPT];
  int ent[SMALL_VPT], g0[SMALL_VPT], g1[SMALL_VPT], zoff[SMALL_VPT];  // zoff: the entry's place in zv / rv
  double x[SMALL_VPT], r[SMALL_VPT], p[SMALL_VPT], sv[SMALL_VPT], z[SMALL_VPT];
#pragma unroll
  for (int v = 0; v < SMALL_VPT; ++v) {
    const int t = (wv + (SMALL_THREADS / 64) * v) * SMALL_ROWS_PER_WAVE + lane;
    ent[v] = (lane < SMALL_ROWS_PER_WAVE && t < N) ? t : -1;
    zoff[v] = t + t / 6;
    row6[v] = t % 6;
    x[v] = 0.0; p[v] = 0.0; sv[v] = 0.0; z[v] = 0.0; r[v] = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) mrow[v][c] = 0.0;
    g0[v] = 0; g1[v] = 0;
    if (ent[v] >= 0) {
      r[v] = -d.rhs[t];
      rv[zoff[v]] = r[v];
      g0[v] = d.brow_ptr[t / 6];
      g1[v] = d.brow_ptr[t / 6 + 1];
#pragma unroll
      for (int c = 0; c < 6; ++c) mrow[v][c] = d.Minv[6 * t + c];  // row (t mod 6) of camera (t / 6)'s 6x6 inverse
    }
  }
  __syncthreads();
  int gi[SMALL_VPT][SMALL_GREG];  // slot offsets of the row's first blocks (a zeroed slot pads short rows)
#pragma unroll
  for (int v = 0; v < SMALL_VPT; ++v)
#pragma unroll
    for (int k = 0; k < SMALL_GREG; ++k)
      gi[v][k] = (ent[v] >= 0 && g0[v] + k < g1[v]) ? gl[g0[v] + k] + ent[v] % 6 : d.nu * SMALL_CSTRIDE;
  auto precondition = [&]() {  // z = Minv r for this thread's rows (r of the camera is in rv), published in zv
    if (COARSE && coarse) {
      const int per_wave = (SMALL_ROWS_PER_WAVE / 6) / ag;  // aggregates among the 10 cameras of a wave: 2 or 1
#pragma unroll
      for (int v = 0; v < SMALL_VPT; ++v) {
        const int cam0 = (wv + (SMALL_THREADS / 64) * v) * (SMALL_ROWS_PER_WAVE / 6);
        const int al = lane / 6, comp = lane - 6 * al, a = cam0 / ag + al;
        if (al < per_wave && 6 * a < nc) {
          double s = 0.0;
          for (int q = 0; q < ag; ++q) {
            const int cam = a * ag + q;
            if (cam < d.Cf) s += rv[SMALL_ZSTRIDE * cam + comp];
          }
          rcv[6 * a + comp] = s;
        }
      }
      __syncthreads();
      {  // every thread: the shuffles below need all eight lanes of a row (8 x 64 = the workgroup)
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < SMALL_COARSE_K; ++k) {
          const int b = (tid & (SMALL_COARSE_PARTS - 1)) + SMALL_COARSE_PARTS * k;
          s += cw[k] * rcv[b < nc ? b : 0];  // cw is zero beyond nc
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if ((tid & (SMALL_COARSE_PARTS - 1)) == 0 && tid < SMALL_COARSE_PARTS * nc) zcv[tid / SMALL_COARSE_PARTS] = s;
      }
      __syncthreads();
    }
#pragma unroll
    for (int v = 0; v < SMALL_VPT; ++v)
      if (ent[v] >= 0) {
        const double* rc = rv + zoff[v] - row6[v];  // r of this row's camera
        double zi = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) zi += mrow[v][c] * rc[c];
        if (COARSE && coarse) zi += zcv[6 * ((ent[v] / 6) / ag) + row6[v]];
        z[v] = zi;
        zv[zoff[v]] = zi;
      }
  };
#ifdef SMALL_CLOCKS
  long long small_cut[32];
  for (int i = 0; i < 32; ++i) small_cut[i] = 0;
  SMALL_CUT(0);
#endif
  precondition();
  __syncthreads();
  SMALL_CUT(1);
  double gamma = 0.0, gamma0 = 0.0, delta_ = 0.0, inv_gamma_old = 1.0, denom_old = 1.0;
  static_assert(SMALL_THREADS == 512, "the partial sums below are written out for 8 waves");
  int it = 0;
  while (true) {
    // (A) every stored block: B z_j and B^T z_i into its slot, and the block's share of delta = z^T S z; the row
    // threads add their share of gamma = r.z: both sums are complete at the barrier that publishes the slots (two
    // barriers per iteration, not three: no reduction after the gather)
    double pg = 0.0, pd = 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (ij[k] >= 0)
        pd += small_block_products(B[k], ij[k] >> 16, ij[k] & 0xffff, zv, slot + (size_t)(tid + k * SMALL_THREADS) * SMALL_CSTRIDE);
#pragma unroll
    for (int v = 0; v < SMALL_VPT; ++v)
      if (ent[v] >= 0) pg += r[v] * z[v];
    pg = wave_sum_dpp(pg);
    pd = wave_sum_dpp(pd);
    if (lane == 0) { red[2 * wv] = pg; red[2 * wv + 1] = pd; }
    SMALL_CUT(2 + 4 * it);
    __syncthreads();
    SMALL_CUT(3 + 4 * it);
    {
      const double g01 = red[0] + red[2], g23 = red[4] + red[6], g45 = red[8] + red[10], g67 = red[12] + red[14];
      const double d01 = red[1] + red[3], d23 = red[5] + red[7], d45 = red[9] + red[11], d67 = red[13] + red[15];
      gamma = (g01 + g23) + (g45 + g67);
      delta_ = (d01 + d23) + (d45 + d67);
    }
    if (it == 0) gamma0 = gamma;
    if (it >= max_iter || !(gamma > tol2 * gamma0) || !(gamma > 0.0)) break;
    // beta = gamma / gamma_old; alpha = gamma / (delta - beta gamma / alpha_old) = gamma / (delta - beta^2 denom_old):
    // one reciprocal on the critical path, 1 / gamma for the next iteration computed beside it
    const double beta = it ? gamma * inv_gamma_old : 0.0;
    const double denom = it ? delta_ - beta * beta * denom_old : delta_;
    if (!(denom > 0.0)) break;
    const double alpha = gamma * fast_rcp(denom);
    inv_gamma_old = fast_rcp(gamma);
    denom_old = denom;
    // (B) w = S z gathered per row, (C) p = z + beta p; s = w + beta s (= S p); x += alpha p; r -= alpha s; z = Minv r
    double w[SMALL_VPT];
#pragma unroll
    for (int v = 0; v < SMALL_VPT; ++v) {
      w[v] = 0.0;
      if (ent[v] >= 0) {
        // the first SMALL_GREG slots of the row come from register-held offsets: independent LDS reads
        double w0 = 0.0, w1 = 0.0;
#pragma unroll
        for (int k = 0; k < SMALL_GREG; k += 2) {
          w0 += slot[gi[v][k]];
          w1 += slot[gi[v][k + 1]];
        }
        for (int e = g0[v] + SMALL_GREG; e < g1[v]; ++e) w0 += slot[gl[e] + row6[v]];
        w[v] = w0 + w1;
      }
    }
#pragma unroll
    for (int v = 0; v < SMALL_VPT; ++v)
      if (ent[v] >= 0) {
        p[v] = z[v] + beta * p[v];
        sv[v] = w[v] + beta * sv[v];
        x[v] += alpha * p[v];
        r[v] -= alpha * sv[v];
        rv[zoff[v]] = r[v];
      }
    SMALL_CUT(4 + 4 * it);
    __builtin_amdgcn_wave_barrier();  // the 6 rows of a camera belong to this wave: LDS keeps a wave's accesses in order
    precondition();
    SMALL_CUT(5 + 4 * it);
    ++it;
    __syncthreads();
  }
#ifdef SMALL_CLOCKS
  if (threadIdx.x == 0) {
    printf("k_solve_small its %d: setup->loop %lld |", it, small_cut[1] - small_cut[0]);
    for (int q = 0; q < 6; ++q)
      printf(" [prod %lld bar %lld scal+gather+upd %lld precond %lld next %lld]", small_cut[2 + 4 * q] - (q ? small_cut[1 + 4 * q] : small_cut[1]),
             small_cut[3 + 4 * q] - small_cut[2 + 4 * q], small_cut[4 + 4 * q] - small_cut[3 + 4 * q], small_cut[5 + 4 * q] - small_cut[4 + 4 * q], small_cut[6 + 4 * q] - small_cut[5 + 4 * q]);
    printf(" (x10 ns)\n");
  }
#endif
#pragma unroll
  for (int v = 0; v < SMALL_VPT; ++v)
    if (ent[v] >= 0) d.dc[ent[v]] = x[v];
  if (tid == 0) {
    d.scal[SC_PCGIT] = (double)it;
    d.scal[SC_RZ] = gamma;
    d.scal[SC_RZ0] = gamma0;
    d.scal[SC_DONE] = 1.0;
  }
}

// ------------------------------------------------------------------------------------------------
// k_solve_components (ba_components.h): dense Cholesky per connected component of the camera graph, a wave each.
constexpr int COMP_MAX_CAMS = 12;


// ------------------------------------------------------------------------------------------------
// pose update: trial poses = retract(poses, dc); camera part of the step statistics
// (g.d, d^T U d, |d|^2, |x|^2).  One thread per camera; runs as the trailing blocks of k_backsub.
// t <- t + dt, q <- normalise(q (x) [1, dtheta / 2])   (Hamilton product, reference src/geometry.py:504-512).
// One function for the trial poses that the pose blocks write and for the trial poses that k_backsub_flat forms per
// observation for the trial cost: the same bits.
__device__ __forceinline__ void retract_pose(const double* __restrict__ ps, const double dl[6], double po[7]) {
  po[0] = ps[0] + dl[0]; po[1] = ps[1] + dl[1]; po[2] = ps[2] + dl[2];
  const double w1 = ps[3], x1 = ps[4], y1 = ps[5], z1 = ps[6];
  const double x2 = 0.5 * dl[3], y2 = 0.5 * dl[4], z2 = 0.5 * dl[5];
  const double w = w1 - x1 * x2 - y1 * y2 - z1 * z2;
  const double x = w1 * x2 + x1 + y1 * z2 - z1 * y2;
  const double y = w1 * y2 - x1 * z2 + y1 + z1 * x2;
  const double z = w1 * z2 + x1 * y2 - y1 * x2 + z1;
  const double inv = 1.0 / sqrt(w * w + x * x + y * y + z * z);
  po[3] = w * inv; po[4] = x * inv; po[5] = y * inv; po[6] = z * inv;
}

__device__ __forceinline__ void pose_update_block(const DeviceProblem& d, const double* __restrict__ poses,
                                                  double* __restrict__ poses_t, int cam_block, double* red /*[4]*/) {
  const int cam = cam_block * 256 + threadIdx.x;
  double gd = 0.0, dhd = 0.0, dn = 0.0, xn = 0.0, gmax = 0.0;
  if (cam < d.C) {
    const int f = d.cam2free[cam];
    const double* ps = poses + 7 * cam;
    double* pt = poses_t + 7 * cam;
    if (f < 0) {
#pragma unroll
      for (int i = 0; i < 7; ++i) pt[i] = ps[i];
    } else {
      double dl[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) dl[i] = d.dc[6 * f + i];
      const double* u = d.Ugc + (long)cam * 27;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        gd += u[21 + i] * dl[i];
        gmax = fmax(gmax, fabs(u[21 + i]));
        dn += dl[i] * dl[i];
#pragma unroll
        for (int c = 0; c < 6; ++c) dhd += dl[i] * u[i <= c ? sym6_index(i, c) : sym6_index(c, i)] * dl[c];
      }
      double po[7];
      retract_pose(ps, dl, po);
#pragma unroll
      for (int i = 0; i < 7; ++i) pt[i] = po[i];
      xn = ps[0] * ps[0] + ps[1] * ps[1] + ps[2] * ps[2] + 1.0;
      if (d.clm.lam) d.clm.pose_val[f] = gd + 0.5 * dhd;  // (summed per component, in camera order, by k_lm_decide_comp)
    }
  }
  // one exchange for the five statistics (a block_sum_256 each is two barriers apiece)
  __shared__ double pred[4][5];
  {
    const double w0 = wave_sum_dpp(gd), w1 = wave_sum_dpp(dhd), w2 = wave_sum_dpp(dn), w3 = wave_sum_dpp(xn),
                 w4 = group_max<64>(gmax);
    if ((threadIdx.x & 63) == 0) {
      double* r5 = pred[threadIdx.x >> 6];
      r5[0] = w0; r5[1] = w1; r5[2] = w2; r5[3] = w3; r5[4] = w4;
    }
  }
  __syncthreads();
  if (threadIdx.x < 5) {  // a plain store per statistic: k_fold_stats sums the pose blocks in a fixed order
    const int t = threadIdx.x;
    d.pose_part[5L * cam_block + t] = t < 4 ? (pred[0][t] + pred[1][t]) + (pred[2][t] + pred[3][t])
                                            : fmax(fmax(pred[0][4], pred[1][4]), fmax(pred[2][4], pred[3][4]));
  }
  (void)red;
}

__global__ __launch_bounds__(256) void k_pose_update(DeviceProblem d, const double* __restrict__ poses,
                                                     double* __restrict__ poses_t) {
  __shared__ double red[4];
  if (poses == nullptr) {
    const LmCtrl* c = d.ctrl;
    if (c->done) return;
    poses = d.poses_buf[c->sel];
    poses_t = d.poses_buf[c->sel ^ 1];
  }
  pose_update_block(d, poses, poses_t, blockIdx.x, red);
}

// ------------------------------------------------------------------------------------------------
// k_backsub: dp_j = -Vd_j^-1 (gp_j + sum_k W_k^T dc_cam(k)),  trial point = P_j + dp_j, and the
// point part of the step statistics (g.d, d^T V d + 2 dc^T W dp, |d|^2, |x|^2).
// Same group-per-track walk as k_linearize.  Algorithmic bytes: W 144 B/obs + 4 B/obs,
// per landmark V 48 + gp 24 + P 24 read, P 24 written.
constexpr double GUARD_KEEP = 0.1;  // a guarded landmark step keeps every depth above this fraction of its value
template <int G>
__global__ __launch_bounds__(256) void k_backsub(DeviceProblem d, double lam, double min_diag,
                                                 const double* __restrict__ pts, double* __restrict__ pts_t,
                                                 int lm_per_group, const double* __restrict__ poses,
                                                 double* __restrict__ poses_t, int pose_blocks, int guard) {
  __shared__ double red[4];
  if (lam < 0.0) {
    const LmCtrl* c = d.ctrl;
    if (c->done) return;
    guard = c->guard_cheirality;
    lam = c->lam;
    pts = d.pts_buf[c->sel];
    pts_t = d.pts_buf[c->sel ^ 1];
    poses = d.poses_buf[c->sel];
    poses_t = d.poses_buf[c->sel ^ 1];
  }
  // the last pose_blocks blocks of the grid retract the poses (k_pose_update's work, no launch of its own)
  if ((int)blockIdx.x >= (int)gridDim.x - pose_blocks) {
    pose_update_block(d, poses, poses_t, blockIdx.x - (gridDim.x - pose_blocks), red);
    return;
  }
  const int lane = threadIdx.x & (G - 1);
  const long group = ((long)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const long j0 = group * lm_per_group;
  const long j1 = min((long)d.L_obs, j0 + lm_per_group);
  const long n = d.n;
  double gd = 0.0, dhd = 0.0, dn = 0.0, xn = 0.0;
  int nshort = 0;
  for (long j = j0; j < j1; ++j) {
    const int kb = d.lm_ptr[j], ke = d.lm_ptr[j + 1];
    double t[3] = {0, 0, 0};
    for (int k = kb + lane; k < ke; k += G) {
      const int f = d.o_free[k];
      if (f < 0) continue;
      double dl[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) dl[i] = d.dc[6 * f + i];
      double jw[W_ROWS], w18[18];
#pragma unroll
      for (int i = 0; i < W_ROWS; ++i) jw[i] = d.W[(long)i * n + k];
      w_from_j(jw, w18);
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) t[c] += w18[3 * i + c] * dl[i];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c] = group_sum<G>(t[c]);
    // every lane of the group forms the step (the operands are the same addresses for all of them: broadcast loads)
    double v[6], vi[6], g[3], s[3], dp[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = d.V[(long)i * d.L + j];
#pragma unroll
    for (int i = 0; i < 3; ++i) g[i] = d.gp[(long)i * d.L + j];
    damped_inv3(v, lam, min_diag, vi);
    s[0] = g[0] + t[0]; s[1] = g[1] + t[1]; s[2] = g[2] + t[2];
    sym3_mul(vi, s, dp);
    dp[0] = -dp[0]; dp[1] = -dp[1]; dp[2] = -dp[2];
    const double P[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
    if (guard) {
      // Cheirality, landmark by landmark: a step that would take the landmark through the principal plane of one of
      // its cameras (depth z -> <= 0) is shortened so that no depth drops below a tenth of its value; the poses are
      // the current ones (the cameras' own steps are small against such a jump; what still slips through is caught
      // by the whole-step test in k_lm_decide).  The model decrease below is that of the step actually taken.
      double alpha = 1.0;
      for (int k = kb + lane; k < ke; k += G) {
        const double* ps = poses + 7 * d.o_cam[k];
        const double qw = ps[3], qx = ps[4], qy = ps[5], qz = ps[6];
        const double r3[3] = {2 * qx * qz + 2 * qw * qy, 2 * qy * qz - 2 * qw * qx, 1 - 2 * qx * qx - 2 * qy * qy};
        const double z0 = (P[0] - ps[0]) * r3[0] + (P[1] - ps[1]) * r3[1] + (P[2] - ps[2]) * r3[2];
        const double dz = dp[0] * r3[0] + dp[1] * r3[1] + dp[2] * r3[2];
        if (z0 > 0.0 && z0 + dz < GUARD_KEEP * z0) alpha = fmin(alpha, (1.0 - GUARD_KEEP) * z0 / -dz);
      }
      alpha = -group_max<G>(-alpha);
      if (alpha < 1.0) {
        dp[0] *= alpha; dp[1] *= alpha; dp[2] *= alpha;
        nshort += (lane == 0);
      }
    }
    if (lane == 0) {
      pts_t[3 * j] = P[0] + dp[0]; pts_t[3 * j + 1] = P[1] + dp[1]; pts_t[3 * j + 2] = P[2] + dp[2];
      gd += g[0] * dp[0] + g[1] * dp[1] + g[2] * dp[2];
      const double vdp0 = v[0] * dp[0] + v[1] * dp[1] + v[2] * dp[2];
      const double vdp1 = v[1] * dp[0] + v[3] * dp[1] + v[4] * dp[2];
      const double vdp2 = v[2] * dp[0] + v[4] * dp[1] + v[5] * dp[2];
      dhd += dp[0] * vdp0 + dp[1] * vdp1 + dp[2] * vdp2 + 2.0 * (t[0] * dp[0] + t[1] * dp[1] + t[2] * dp[2]);
      dn += dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
      xn += P[0] * P[0] + P[1] * P[1] + P[2] * P[2];
    }
  }
  const double b0 = block_sum_256(gd, red), b1 = block_sum_256(dhd, red), b2 = block_sum_256(dn, red),
               b3 = block_sum_256(xn, red), b4 = block_sum_256((double)nshort, red);
  if (threadIdx.x == 0) {
    double* sc = SCAL_MINE(d);
    atomic_add_f64(&sc[SC_GDOT], b0);
    atomic_add_f64(&sc[SC_DHD], b1);
    atomic_add_f64(&sc[SC_DNORM2], b2);
    atomic_add_f64(&sc[SC_XNORM2], b3);
    if (b4 != 0.0) atomic_add_f64(&sc[SC_NSHORT], b4);
  }
}

// ------------------------------------------------------------------------------------------------
// k_backsub_flat: k_backsub with ONE LANE PER OBSERVATION (all 64 lanes load W, against 5 of 8 in the group-per-track
// kernel on tracks of five), plus the trial cost, so that solver iterations need no k_cost launch and no second pass
// over the observations.  A wave takes one chunk of whole tracks (the chunk tables of k_linearize_flat, built for any
// camera count): every load is issued in two rounds (W, indices, pixels, V, gp, P, zmin; then the camera steps), the
// W^T dc rows are parked in LDS and lane l sums the rows of track l in track order, forms dp, applies the cheirality
// guard and parks the trial point; the observation lanes then retract their camera's pose (retract_pose: the bits that
// the pose blocks write) and evaluate the trial residual.  The guard's exact test needs the pose of every observing
// camera; it is skipped whenever |dp| < (1 - GUARD_KEEP) zmin (zmin: the landmark's smallest depth, left by the
// linearisation), which cannot cross a principal plane: the test then costs 8 B per landmark.
// Algorithmic bytes: those of k_backsub + k_cost minus the second read of the points: 172 B/obs + 152 B/landmark.
// Per-component statistics of a chunk (a wave) without atomics: for every component on the chunk's list (cp_ptr / cp_comp:
// the distinct components of its landmarks, as ba_set_problem found them) the sum of v over the active lanes of that
// component, a wave sum in a fixed order, stored to the entry's slot.  Wave-uniform call; every entry is written in
// every launch, so nothing needs clearing.
__device__ __forceinline__ void comp_store(const CompLm& q, int chunk, int slot, int comp, bool act, double v) {
  const int e0 = q.cp_ptr[chunk], e1 = q.cp_ptr[chunk + 1];
  for (int e = e0; e < e1; ++e) {
    const int ce = q.cp_comp[e];
    const double sum = wave_sum_dpp(act && comp == ce ? v : 0.0);
    if ((threadIdx.x & 63) == 0) q.cp_val[3L * e + slot] = sum;
  }
}

template <bool ROBUST, bool MF>
__global__ __launch_bounds__(256, MF ? 4 : 5) void k_backsub_flat(DeviceProblem d, double lam, double min_diag,
                                                      const double* __restrict__ pts, double* __restrict__ pts_t,
                                                      const double* __restrict__ poses, double* __restrict__ poses_t,
                                                      int pose_blocks, int guard, int with_cost) {
  __shared__ double red[4];
  __shared__ double sT[4][64 * 3];
  __shared__ double sPt[4][LINF_MAXLM * 3];
  __shared__ int sComp[4][LINF_MAXLM];
  const bool per_comp = lam < 0.0 && d.clm.lam != nullptr;  // solver mode with a damping and a decision per component
  if (lam < 0.0) {
    const LmCtrl* c = d.ctrl;
    if (c->done) return;
    guard = c->guard_cheirality;
    lam = c->lam;
    pts = d.pts_buf[c->sel];
    pts_t = d.pts_buf[c->sel ^ 1];
    poses = d.poses_buf[c->sel];
    poses_t = d.poses_buf[c->sel ^ 1];
  }
  if ((int)blockIdx.x >= (int)gridDim.x - pose_blocks) {
    pose_update_block(d, poses, poses_t, blockIdx.x - (gridDim.x - pose_blocks), red);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  const bool have = c < d.n_lin_chunks;
  int kb = 0, ke = 0, jb = 0, je = 0;
  if (have) { jb = d.lin_lm[c]; je = d.lin_lm[c + 1]; kb = d.lin_k[c]; ke = d.lin_k[c + 1]; }
  const long n = d.n;
  // the lanes beyond the chunk's observations repeat its last one (as in k_linearize_rows: no branch around the
  // arithmetic, no defaults to merge): their rows of the exchange are read by nobody, their cost is not counted
  const bool act = kb + lane < ke;
  const int k = have ? (act ? kb + lane : ke - 1) : 0;
  // round 1: everything whose address is known from the chunk bounds
  constexpr bool mf = MF;  // matrix free (d.W == nullptr): W is rebuilt from the record, the pose and the point (no stored factors)
  int cam = 0;
  double jw[W_ROWS], u = 0.0, v = 0.0;
  const int f = d.o_free[k], lm = d.o_lm[k];
  if (!mf) {
#pragma unroll
    for (int i = 0; i < W_ROWS; ++i) jw[i] = d.W[(long)i * n + k];
  }
  if (with_cost || mf) { cam = d.o_cam[k]; u = d.o_u[k]; v = d.o_v[k]; }
  const bool lm_lane = lane < je - jb;
  const long j = jb + lane;
  int tb = 0, te = 0;
  double V6[6] = {1, 0, 0, 1, 0, 1}, g[3] = {0, 0, 0}, P[3] = {0, 0, 0}, zmn = 0.0;
  int cj = 0;
  if (lm_lane && per_comp) { lam = d.clm.lam_lm[j]; cj = d.clm.lm_comp[j]; }
  if (lm_lane) {
    tb = d.lm_ptr[j] - kb; te = d.lm_ptr[j + 1] - kb;
#pragma unroll
    for (int i = 0; i < 6; ++i) V6[i] = d.V[(long)i * d.L + j];
#pragma unroll
    for (int i = 0; i < 3; ++i) g[i] = d.gp[(long)i * d.L + j];
#pragma unroll
    for (int i = 0; i < 3; ++i) P[i] = pts[3 * j + i];
    if (guard) zmn = d.zmin[j];
  }
  // round 2: the camera step of this observation (and, for the trial cost, its pose)
  double dl[6] = {0, 0, 0, 0, 0, 0}, ps[7] = {0, 0, 0, 1, 0, 0, 0};
  if (f >= 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) dl[i] = d.dc[6 * f + i];
  }
  if (mf) {  // (stored W: the pose is only needed for the trial cost, and is fetched there: 14 registers less across the middle)
#pragma unroll
    for (int i = 0; i < 7; ++i) ps[i] = poses[7 * cam + i];
  }
  double* myT = sT[wave];
  double* myP = sPt[wave];
  if (mf) {  // the current points to the observation lanes (the same rows take the trial points later, after a barrier)
    if (lm_lane) { myP[3 * lane] = P[0]; myP[3 * lane + 1] = P[1]; myP[3 * lane + 2] = P[2]; }
    __syncthreads();
    {
      const int li = have ? lm - jb : 0;
      const double Pc[3] = {myP[3 * li], myP[3 * li + 1], myP[3 * li + 2]};
      double r[2], Jc[12], Jp[6], zc;
      linearize_obs(d.K, ps, Pc, u, v, r, Jc, Jp, zc);
      if (ROBUST) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          double sc;
          (void)robust_scale(d.loss, r[i], sc);
#pragma unroll
          for (int q = 0; q < 6; ++q) Jc[6 * i + q] *= sc;
#pragma unroll
          for (int q = 0; q < 3; ++q) Jp[3 * i + q] *= sc;
        }
      }
      w_pack(Jc, Jp, f >= 0, jw);
    }
  }
  {  // W^T dl = Jp^T (Jc dl), Jc = [-Jp | Jr]: the 6x3 block is never formed
    const double a0 = -(jw[0] * dl[0] + jw[1] * dl[1] + jw[2] * dl[2]) + jw[6] * dl[3] + jw[7] * dl[4] + jw[8] * dl[5];
    const double a1 = -(jw[3] * dl[0] + jw[4] * dl[1] + jw[5] * dl[2]) + jw[9] * dl[3] + jw[10] * dl[4] + jw[11] * dl[5];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) myT[3 * lane + cc] = jw[cc] * a0 + jw[3 + cc] * a1;
  }
  __syncthreads();  // (each wave only reads its own rows; every wave of the block reaches the barriers)
  double gd = 0.0, dhd = 0.0, dn = 0.0, xn = 0.0;
  int nshort = 0;
  if (lm_lane) {
    double t[3] = {0, 0, 0};
    for (int r = tb; r < te; ++r) {
      t[0] += myT[3 * r]; t[1] += myT[3 * r + 1]; t[2] += myT[3 * r + 2];
    }
    double vi[6], s3[3], dp[3];
    damped_inv3(V6, lam, min_diag, vi);
    s3[0] = g[0] + t[0]; s3[1] = g[1] + t[1]; s3[2] = g[2] + t[2];
    sym3_mul(vi, s3, dp);
    dp[0] = -dp[0]; dp[1] = -dp[1]; dp[2] = -dp[2];
    if (guard) {
      const double keep = (1.0 - GUARD_KEEP) * zmn;
      if (!(zmn > 0.0) || dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2] >= keep * keep) {  // rare: the exact test of k_backsub
        double alpha = 1.0;
        for (int r = tb; r < te; ++r) {
          const double* pc = poses + 7 * d.o_cam[kb + r];
          const double qw = pc[3], qx = pc[4], qy = pc[5], qz = pc[6];
          const double r3[3] = {2 * qx * qz + 2 * qw * qy, 2 * qy * qz - 2 * qw * qx, 1 - 2 * qx * qx - 2 * qy * qy};
          const double z0 = (P[0] - pc[0]) * r3[0] + (P[1] - pc[1]) * r3[1] + (P[2] - pc[2]) * r3[2];
          const double dz = dp[0] * r3[0] + dp[1] * r3[1] + dp[2] * r3[2];
          if (z0 > 0.0 && z0 + dz < GUARD_KEEP * z0) alpha = fmin(alpha, (1.0 - GUARD_KEEP) * z0 / -dz);
        }
        if (alpha < 1.0) { dp[0] *= alpha; dp[1] *= alpha; dp[2] *= alpha; nshort = 1; }
      }
    }
    const double Pt[3] = {P[0] + dp[0], P[1] + dp[1], P[2] + dp[2]};
    pts_t[3 * j] = Pt[0]; pts_t[3 * j + 1] = Pt[1]; pts_t[3 * j + 2] = Pt[2];
    myP[3 * lane] = Pt[0]; myP[3 * lane + 1] = Pt[1]; myP[3 * lane + 2] = Pt[2];
    sComp[wave][lane] = cj;
    gd = g[0] * dp[0] + g[1] * dp[1] + g[2] * dp[2];
    const double vdp0 = V6[0] * dp[0] + V6[1] * dp[1] + V6[2] * dp[2];
    const double vdp1 = V6[1] * dp[0] + V6[3] * dp[1] + V6[4] * dp[2];
    const double vdp2 = V6[2] * dp[0] + V6[4] * dp[1] + V6[5] * dp[2];
    dhd = dp[0] * vdp0 + dp[1] * vdp1 + dp[2] * vdp2 + 2.0 * (t[0] * dp[0] + t[1] * dp[1] + t[2] * dp[2]);
    dn = dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
    xn = P[0] * P[0] + P[1] * P[1] + P[2] * P[2];
  }
  if (per_comp && have) comp_store(d.clm, c, 0, cj, lm_lane, gd + 0.5 * dhd);
  double cost = 0.0;
  int nbehind = 0, nzero = 0;
  if (with_cost) {
    if (!mf) {
#pragma unroll
      for (int i = 0; i < 7; ++i) ps[i] = poses[7 * cam + i];
    }
    __syncthreads();
    {
      const int li = have ? lm - jb : 0;
      const double Pt[3] = {myP[3 * li], myP[3 * li + 1], myP[3 * li + 2]};
      double po[7];
      if (f >= 0) retract_pose(ps, dl, po);
      else {
#pragma unroll
        for (int i = 0; i < 7; ++i) po[i] = ps[i];
      }
      double R[9], X[3], r0, r1;
      quat_to_R(po + 3, R);
      to_camera(R, po, Pt, X);
      reproject(d.K, X, u, v, r0, r1);
      nbehind = act && X[2] <= 0.0;
      nzero = act && X[2] == 0.0;
      const double cost_k = !ROBUST ? 0.5 * (r0 * r0 + r1 * r1) : robust_cost(d.loss, r0) + robust_cost(d.loss, r1);
      cost = act ? cost_k : 0.0;
    }
    if (per_comp && have) {
      const int co = act ? sComp[wave][lm - jb] : 0;
      comp_store(d.clm, c, 1, co, act, cost);
      comp_store(d.clm, c, 2, co, act, (double)nbehind);
    }
  }
  // the block's eight statistics in ONE exchange: wave sums (the three counters by ballot), one barrier, eight atomics
  // (one block_sum_256 each cost two barriers apiece: sixteen at the end of every block)
  __shared__ double sred[4][8];
  {
    const double w0 = wave_sum_dpp(gd), w1 = wave_sum_dpp(dhd), w2 = wave_sum_dpp(dn), w3 = wave_sum_dpp(xn),
                 w5 = with_cost ? wave_sum_dpp(cost) : 0.0;
    const double w4 = (double)__popcll(__ballot(nshort != 0)), w6 = (double)__popcll(__ballot(nbehind != 0)),
                 w7 = (double)__popcll(__ballot(nzero != 0));
    if (lane == 0) {
      double* r8 = sred[wave];
      r8[0] = w0; r8[1] = w1; r8[2] = w2; r8[3] = w3; r8[4] = w4; r8[5] = w5; r8[6] = w6; r8[7] = w7;
    }
  }
  __syncthreads();
  if (tid < 8)  // a plain store per statistic: k_fold_stats sums the blocks in a fixed order (no atomics)
    d.bs_part[8L * blockIdx.x + tid] = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
}

// Copy the untouched (unobserved) landmarks [L_obs, L) into the trial buffer.
__global__ void k_copy_tail(const double* __restrict__ src, double* __restrict__ dst, long begin, long end) {
  for (long i = begin + (long)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (long)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// k_fold_scalars (multi-GPU only): fold the privatised scalar copies into out[0, SC_NSUM) (sum; max
// for the two inf-norm slots) so that ONE small sum all-reduce can combine the ranks.  Slots whose
// value is replicated (camera part) are zeroed on every rank but the root, so that the sum over ranks
// is the root's value.  The point-gradient inf-norm needs a max over ranks: rank r puts its value in
// out[SC_COUNT + r] and zeros in the other ranks' slots; after the sum every rank holds all of them.
constexpr int FOLD_MAX_RANKS = 64;
__global__ __launch_bounds__(128) void k_fold_scalars(DeviceProblem d, double* __restrict__ out, int rank, int world) {
  const int i = threadIdx.x;
  if (i >= SC_COUNT) {
    if (i - SC_COUNT < world) out[i] = 0.0;
    if (i - SC_COUNT != rank) return;
  }
  const int slot = i >= SC_COUNT ? SC_GINF : i;
  double v = 0.0;
  if (slot < SC_NSUM) {
    const bool is_max = (slot == SC_GINF || slot == SC_GINF_C);
    for (int k = 0; k < SCAL_SLABS; ++k) {
      const double x = d.scal[(size_t)k * SC_COUNT + slot];
      v = is_max ? fmax(v, x) : v + x;
    }
    const bool replicated = (slot == SC_GDOT_C || slot == SC_DHD_C || slot == SC_DNORM2_C || slot == SC_XNORM2_C ||
                             slot == SC_GINF_C);
    if (replicated && rank != 0) v = 0.0;
    if (i == SC_GINF) v = 0.0;  // travels in the per-rank slots
  }
  out[i] = v;
}

// ------------------------------------------------------------------------------------------------
// k_lm_decide: the Levenberg-Marquardt bookkeeping of one iteration, on the device (one wave).
// Folds the privatised scalar copies (or takes the rank-combined values from `folded`), then does
// exactly what the host used to do after a read-back: gain ratio, accept / reject, Nielsen's lambda
// update, scipy-style termination tests (scipy/optimize/_lsq/common.py:705-717 vocabulary).
__device__ __forceinline__ void lm_publish(LmMirror* mirror, const LmHead& h, int seq) {
  if (!mirror) return;
  mirror->head = h;
  __threadfence_system();
  __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_lm_decide(DeviceProblem d, const double* __restrict__ folded, int world,
                                                   LmMirror* mirror, int seq) {
  LmCtrl* c = d.ctrl;
  if (c->done) {  // launched after termination: only tell the host that this launch is through
    if (threadIdx.x == 0) lm_publish(mirror, *static_cast<const LmHead*>(c), seq);
    return;
  }
  __shared__ double part[8][SC_COUNT];
  __shared__ double s[SC_COUNT];
  // the control head and the solver's iteration count are requested with the statistics: one round trip, not two
  const LmHead h_in = *static_cast<const LmHead*>(c);
  const double pcg_in = d.scal[SC_PCGIT];
  const int t = threadIdx.x, i = t & (SC_COUNT - 1), g = t / SC_COUNT;  // 8 groups x 32 slots
  {
    const bool is_max = (i == SC_GINF || i == SC_GINF_C);
    double v = 0.0;
    if (!folded) {
#pragma unroll
      for (int k = 0; k < SCAL_SLABS / 8; ++k) {
        const double x = d.scal[(size_t)(g * (SCAL_SLABS / 8) + k) * SC_COUNT + i];
        v = is_max ? fmax(v, x) : v + x;
      }
    }
    // the statistics are consumed here: leave the accumulators clean for the next iteration (the PCG's own
    // slots, >= SC_NSUM, are written by the solver itself)
    if (i < SC_NSUM) {
#pragma unroll
      for (int k = 0; k < SCAL_SLABS / 8; ++k) d.scal[(size_t)(g * (SCAL_SLABS / 8) + k) * SC_COUNT + i] = 0.0;
    }
    part[g][i] = v;
  }
  __syncthreads();
  if (t < SC_NSUM) {
    double v;
    if (folded) {
      v = folded[t];
      if (t == SC_GINF)
        for (int r = 0; r < world; ++r) v = fmax(v, folded[SC_COUNT + r]);
    } else {
      const bool is_max = (t == SC_GINF || t == SC_GINF_C);
      v = part[0][t];
#pragma unroll
      for (int k = 1; k < 8; ++k) v = is_max ? fmax(v, part[k][t]) : v + part[k][t];
    }
    s[t] = v;
  }
  __syncthreads();
  if (t != 0) return;
  // one snapshot of the control head (wide loads, one round trip), written back once at the end
  LmHead h = h_in;
  const int pcg_it = (int)pcg_in;
  const int it = ++h.it;
  bool early = false;
  if (h.need_lin) {
    h.cost = s[SC_COST];
    h.nbehind = s[SC_NBEHIND];
    h.ginf_pts = s[SC_GINF];
    h.linearizations++;
    h.need_lin = 0;
    if (it == 1) {
      h.cost0 = h.cost;
      if (s[SC_ZEROZ] > 0.0) { h.error = 1; h.done = 1; early = true; }
      else if (!isfinite(h.cost)) { h.error = 2; h.done = 1; early = true; }
    }
  }
  if (!early) {
    const double ginf = fmax(h.ginf_pts, s[SC_GINF_C]);
    h.ginf = ginf;
    h.pcg_total += pcg_it;
    if (ginf <= h.gtol) { h.status = 1; h.done = 1; early = true; }
  }
  if (!early) {
    const double cost = h.cost;
    const double cost_t = s[SC_COST_T];
    const double pred = -(s[SC_GDOT] + s[SC_GDOT_C]) - 0.5 * (s[SC_DHD] + s[SC_DHD_C]);
    double rho = (pred > 0.0 && isfinite(cost_t)) ? (cost - cost_t) / pred : -1.0;
    if (h.guard_cheirality && s[SC_NBEHIND_T] > h.nbehind) { rho = -1.0; h.guard_rejections++; }
    h.shortened_steps += (int)fmin(s[SC_NSHORT], 1e9);
    // a factorisation that met a non-positive pivot (or a landmark / camera block that was not positive definite)
    // solved a perturbed system: its step is not trusted, and more damping is the cure
    if (s[SC_BADV] > 0.0) { rho = -1.0; h.bad_pivots += (int)fmin(s[SC_BADV], 1e9); }
    const double step_norm = sqrt(s[SC_DNORM2] + s[SC_DNORM2_C]);
    const double x_norm = sqrt(s[SC_XNORM2] + s[SC_XNORM2_C]);
    const bool accept = rho > h.accept_ratio;
    const int hi = h.n_hist < LM_HISTORY ? h.n_hist++ : LM_HISTORY - 1;
    c->lam_hist[hi] = h.lam;
    c->pcg_hist[hi] = pcg_it;
    c->acc_hist[hi] = accept;
    if (accept && it - 1 < LM_ACC_BITS) c->acc_bits[(it - 1) >> 5] |= 1u << ((it - 1) & 31);
    c->trial_hist[hi] = cost_t;
    c->rho_hist[hi] = rho;
    c->ginf_hist[hi] = h.ginf;
    c->step_hist[hi] = step_norm;
    bool stop = false;
    if (accept) {
      const double dF = cost - cost_t;
      h.sel ^= 1;
      h.cost = cost_t;
      h.nbehind = s[SC_NBEHIND_T];
      h.accepted++;
      const double tt = 2.0 * rho - 1.0;
      h.lam = fmax(h.lambda_min, h.lam * fmax(1.0 / 3.0, 1.0 - tt * tt * tt));
      h.nu = 2.0;
      h.need_lin = 1;
      if (dF <= h.ftol * cost_t) { h.status = 2; stop = true; }
      else if (step_norm <= h.xtol * (x_norm + h.xtol)) { h.status = 3; stop = true; }
    } else {
      h.lam = fmin(h.lambda_max, h.lam * h.nu);
      h.nu *= 2.0;
      if (step_norm <= h.xtol * (x_norm + h.xtol)) { h.status = 3; stop = true; }
    }
    c->cost_hist[hi] = h.cost;
    if (stop || it >= h.max_iters) h.done = 1;  // status stays 0 (max_iters) unless a test fired
  }
  *static_cast<LmHead*>(c) = h;
  lm_publish(mirror, h, seq);
}

// ------------------------------------------------------------------------------------------------
// Levenberg-Marquardt per connected component (CompLm): batched windows no longer share one damping sequence and one
// accept / reject decision.  Per LM iteration, after the trial kernels: k_lm_decide_comp (one workgroup: every
// component's gain ratio, decision and Nielsen update from its own statistics; the problem is done when every
// component is) and k_commit_comp (accepted components copy their trial poses / points into the current buffers, every
// landmark gets its component's new damping).  The current estimate stays in buffer 0 and every iteration linearises.
// A component is frozen once an accepted step gains less than ftol of its cost (or its damping hits lambda_max).

__global__ __launch_bounds__(256) void k_comp_init(DeviceProblem d, double lambda0) {
  const CompLm& c = d.clm;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < max(d.L_obs, d.n_comp); t += gridDim.x * blockDim.x) {
    if (t < d.L_obs) c.lam_lm[t] = lambda0;
    if (t < d.n_comp) {
      c.lam[t] = lambda0; c.nu[t] = 2.0; c.cost[t] = 0.0; c.nbehind[t] = 0.0; c.accept[t] = 0; c.done[t] = 0;
#pragma unroll
      for (int i = 0; i < CL_STRIDE; ++i) c.stat[(long)CL_STRIDE * t + i] = 0.0;
    }
  }
}

// cost and cheirality count of every component at (poses, pts): a wave per chunk of the flat kernels, a lane per
// observation, the chunk's per-component sums into cp_val (slots 1 and 2); k_comp_fold_cost adds up a component's entries
template <bool ROBUST>
__global__ __launch_bounds__(256) void k_comp_cost(DeviceProblem d, const double* __restrict__ poses,
                                                   const double* __restrict__ pts) {
  const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (c >= d.n_lin_chunks) return;
  const int k = d.lin_k[c] + (threadIdx.x & 63);
  const bool act = k < d.lin_k[c + 1];
  double cost = 0.0, nb = 0.0;
  int comp = 0;
  if (act) {
    const int cam = d.o_cam[k], lm = d.o_lm[k];
    comp = d.clm.lm_comp[lm];
    const double* ps = poses + 7 * cam;
    const double P[3] = {pts[3L * lm], pts[3L * lm + 1], pts[3L * lm + 2]};
    double R[9], X[3], r0, r1;
    quat_to_R(ps + 3, R);
    to_camera(R, ps, P, X);
    reproject(d.K, X, d.o_u[k], d.o_v[k], r0, r1);
    nb = X[2] <= 0.0 ? 1.0 : 0.0;
    cost = !ROBUST ? 0.5 * (r0 * r0 + r1 * r1) : robust_cost(d.loss, r0) + robust_cost(d.loss, r1);
  }
  comp_store(d.clm, c, 1, comp, act, cost);
  comp_store(d.clm, c, 2, comp, act, nb);
}

__global__ __launch_bounds__(256) void k_comp_fold_cost(DeviceProblem d) {  // a wave per component, as comp_fold_stats_block
  const CompLm& q = d.clm;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= d.n_comp) return;
  double cost = 0.0, nb = 0.0;
  for (int i = q.cc_ptr[w] + lane; i < q.cc_ptr[w + 1]; i += 64) {
    const long e = q.cc_ent[i];
    cost += q.cp_val[3 * e + 1];
    nb += q.cp_val[3 * e + 2];
  }
  cost = wave_sum_dpp(cost); nb = wave_sum_dpp(nb);
  if (lane == 0) { q.cost[w] = cost; q.nbehind[w] = nb; }
}

constexpr int DECIDE_COMP_THREADS = 1024;
__global__ __launch_bounds__(DECIDE_COMP_THREADS) void k_lm_decide_comp(DeviceProblem d, LmMirror* mirror, int seq) {
  LmCtrl* c = d.ctrl;
  if (c->done) {
    if (threadIdx.x == 0) lm_publish(mirror, *static_cast<const LmHead*>(c), seq);
    return;
  }
  __shared__ double part[8][SC_COUNT];
  __shared__ double s[SC_COUNT];
  __shared__ double red[5][DECIDE_COMP_THREADS / 64];
  const int t = threadIdx.x;
  const CompLm& q = d.clm;
  // everything this thread's first component needs and the control head, requested before the fold of the global
  // statistics: one round trip for all of it instead of three in a row
  struct CompIn { double st0, cost_t, behind_t, badp, behind, cost, lam, nu; int done; };
  auto load_comp = [&](int w) {
    const double* st = q.stat + (long)CL_STRIDE * w;
    return CompIn{st[0], st[2], st[3], st[4], q.nbehind[w], q.cost[w], q.lam[w], q.nu[w], q.done[w]};
  };
  CompIn in0{};
  if (t < d.n_comp) in0 = load_comp(t);
  const LmHead h0 = *static_cast<const LmHead*>(c);
  if (t < 256) {  // the global statistics (gradient norms, step norms, counters): folded and cleared as k_lm_decide does
    const int i = t & (SC_COUNT - 1), g = t / SC_COUNT;
    const bool is_max = (i == SC_GINF || i == SC_GINF_C);
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < SCAL_SLABS / 8; ++k) {
      const double x = d.scal[(size_t)(g * (SCAL_SLABS / 8) + k) * SC_COUNT + i];
      v = is_max ? fmax(v, x) : v + x;
    }
    if (i < SC_NSUM) {
#pragma unroll
      for (int k = 0; k < SCAL_SLABS / 8; ++k) d.scal[(size_t)(g * (SCAL_SLABS / 8) + k) * SC_COUNT + i] = 0.0;
    }
    part[g][i] = v;
  }
  __syncthreads();
  if (t < SC_NSUM) {
    const bool is_max = (t == SC_GINF || t == SC_GINF_C);
    double v = part[0][t];
#pragma unroll
    for (int k = 1; k < 8; ++k) v = is_max ? fmax(v, part[k][t]) : v + part[k][t];
    s[t] = v;
  }
  __syncthreads();
  const bool landmark_bad = s[SC_BADV] > 0.0;  // a landmark block that was not positive definite: nobody's step is trusted
  double cost_sum = 0.0, cost_before = 0.0, n_acc = 0.0, n_done = 0.0;
  int rejected_guard = 0;
  for (int w = t; w < d.n_comp; w += DECIDE_COMP_THREADS) {
    double* st = q.stat + (long)CL_STRIDE * w;
    const CompIn in = w == t ? in0 : load_comp(w);
    const double st0 = in.st0, cost_t = in.cost_t, behind_t = in.behind_t, badp = in.badp, behind = in.behind;
    double cost = in.cost, lam = in.lam, nu = in.nu;
    int done = in.done, accept = 0;
    cost_before += cost;
    if (!done) {
      const double pred = -st0;
      double rho = (pred > 0.0 && isfinite(cost_t)) ? (cost - cost_t) / pred : -1.0;
      if (h0.guard_cheirality && behind_t > behind) { rho = -1.0; ++rejected_guard; }
      if (landmark_bad || badp > 0.0) rho = -1.0;
      accept = rho > h0.accept_ratio;
      if (accept) {
        const double dF = cost - cost_t;
        cost = cost_t;
        q.nbehind[w] = behind_t;
        const double tt = 2.0 * rho - 1.0;
        lam = fmax(h0.lambda_min, lam * fmax(1.0 / 3.0, 1.0 - tt * tt * tt));
        nu = 2.0;
        if (dF <= h0.ftol * cost_t) done = 1;
      } else {
        if (lam >= h0.lambda_max) done = 1;  // no damping is large enough: give the component up
        lam = fmin(h0.lambda_max, lam * nu);
        nu *= 2.0;
      }
      q.lam[w] = lam; q.nu[w] = nu; q.cost[w] = cost; q.done[w] = done;
    }
    q.accept[w] = accept;
#pragma unroll
    for (int i = 0; i < CL_STRIDE; ++i) st[i] = 0.0;
    cost_sum += cost; n_acc += accept; n_done += done;
  }
  cost_sum = wave_sum_dpp(cost_sum); n_acc = wave_sum_dpp(n_acc); n_done = wave_sum_dpp(n_done);
  cost_before = wave_sum_dpp(cost_before);
  const double gr = wave_sum_dpp((double)rejected_guard);
  if ((t & 63) == 0) {
    red[0][t >> 6] = cost_sum; red[1][t >> 6] = n_acc; red[2][t >> 6] = n_done; red[3][t >> 6] = gr; red[4][t >> 6] = cost_before;
  }
  __syncthreads();
  if (t != 0) return;
  double tot = 0.0, acc = 0.0, dn = 0.0, grj = 0.0, before = 0.0;
  for (int k = 0; k < DECIDE_COMP_THREADS / 64; ++k) {
    tot += red[0][k]; acc += red[1][k]; dn += red[2][k]; grj += red[3][k]; before += red[4][k];
  }
  LmHead h = h0;
  const int it = ++h.it;
  if (it == 1) {
    h.cost0 = before;
    if (s[SC_ZEROZ] > 0.0) { h.error = 1; h.done = 1; }
    else if (!isfinite(before)) { h.error = 2; h.done = 1; }
  }
  h.linearizations++;
  h.nbehind = s[SC_NBEHIND];
  h.ginf_pts = s[SC_GINF];
  h.ginf = fmax(s[SC_GINF], s[SC_GINF_C]);
  h.shortened_steps += (int)fmin(s[SC_NSHORT], 1e9);
  h.guard_rejections += grj > 0.0;
  if (landmark_bad) h.bad_pivots += (int)fmin(s[SC_BADV], 1e9);
  const int hi = h.n_hist < LM_HISTORY ? h.n_hist++ : LM_HISTORY - 1;
  c->lam_hist[hi] = q.lam[0];
  c->pcg_hist[hi] = 0;
  c->acc_hist[hi] = acc > 0.0;
  if (it - 1 < LM_ACC_BITS) c->acc_bits[(it - 1) >> 5] |= 1u << ((it - 1) & 31);  // every iteration linearises
  c->trial_hist[hi] = tot;
  c->rho_hist[hi] = acc;  // (components that accepted)
  c->ginf_hist[hi] = h.ginf;
  c->step_hist[hi] = sqrt(s[SC_DNORM2] + s[SC_DNORM2_C]);
  if (!h.done) {
    if (acc > 0.0) h.accepted++;
    h.cost = tot;
    h.lam = q.lam[0];
    h.need_lin = 1;
    if (h.ginf <= h.gtol) { h.status = 1; h.done = 1; }
    else if (dn >= (double)d.n_comp) { h.status = 2; h.done = 1; }
    else if (it >= h.max_iters) h.done = 1;
  }
  c->cost_hist[hi] = h.cost;
  *static_cast<LmHead*>(c) = h;
  lm_publish(mirror, h, seq);
}

__global__ __launch_bounds__(256) void k_commit_comp(DeviceProblem d) {
  const CompLm& q = d.clm;
  // No look at ctrl->done: the decision that ends the solve still has accepted steps to commit.  A launch enqueued
  // after that finds the same flags and the same trial buffers (every kernel before it exits at once): the same copy.
  double* __restrict__ P0 = d.poses_buf[0];
  const double* __restrict__ P1 = d.poses_buf[1];
  double* __restrict__ X0 = d.pts_buf[0];
  const double* __restrict__ X1 = d.pts_buf[1];
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < max(d.L_obs, d.Cf); t += (long)gridDim.x * blockDim.x) {
    if (t < d.L_obs) {
      const int w = q.lm_comp[t];
      if (q.accept[w]) { X0[3 * t] = X1[3 * t]; X0[3 * t + 1] = X1[3 * t + 1]; X0[3 * t + 2] = X1[3 * t + 2]; }
      q.lam_lm[t] = q.lam[w];
    }
    if (t < d.Cf && q.accept[q.free_comp[t]]) {
      const long cam = d.free2cam[t];
#pragma unroll
      for (int i = 0; i < 7; ++i) P0[7 * cam + i] = P1[7 * cam + i];
    }
  }
}
