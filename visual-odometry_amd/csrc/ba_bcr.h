// Block cyclic reduction of the reduced camera system of a trajectory (gfx950, fp64).
//
// Cameras along a trajectory share landmarks only with their neighbours: with the free cameras grouped, in order,
// into N super-blocks of at least `bwr` cameras each (bwr = the cyclic half-bandwidth of S in blocks), S is block
// TRIDIAGONAL in super-blocks -- cyclic when the trajectory closes into a loop:
//     D_i   (m x m, m = 6 x cameras per super-block, SPD)      the diagonal block of super-block i
//     E_i   (m x m)                                              the block (i, i + 1 mod N)
// ba_banded.h factors such a system column by column: a chain of C (one level) or C / P + P bw (two levels) dependent
// 6-wide column steps of ~4.6 us each -- 0.85 ms per solve at 1000 cameras, 2.7 ms at 10 000, most of an LM iteration
// and, with several GPUs, work that every rank repeats.  Cyclic reduction eliminates every other super-block at once:
//     level l:  for every odd i (one workgroup each, k_bcr_factor):  D_i = L L^T,
//                   Fa = E_{i-1} L^-T,  Fb = E_i^T L^-T,  y = L^-1 r_i,  L^-T itself
//               for every even e (one workgroup each, k_bcr_update), the system of level l + 1 (half the blocks):
//                   D'_{e/2} = D_e - Fb_{e-1} Fb_{e-1}^T - Fa_{e+1} Fa_{e+1}^T
//                   E'_{e/2} = -Fa_{e+1} Fb_{e+1}^T            (= block (e, e + 2))
//                   r'_{e/2} = r_e - Fb_{e-1} y_{e-1} - Fa_{e+1} y_{e+1}
// until two blocks (or BCR_TAIL_ROWS / m, if more) are left, which ONE workgroup factors as a dense matrix in LDS
// (k_bcr_tail: the last two levels of ~28 us each collapse into one kernel); then back up (k_bcr_backsolve):
// x_i = L^-T (y - Fa^T x_{i-1} - Fb^T x_{i+1}).
// The chain is 3 log2(N m / BCR_TAIL_ROWS) + 2 short kernels deep instead of hundreds of column steps, every level is as wide as
// the chip, and all sums run in a fixed order (no atomics): the step is bitwise reproducible, which the multi-GPU
// solve relies on.  The elimination is a block Cholesky in nested-dissection order: Schur complements of an SPD
// matrix, as stable as the banded factorisation it replaces.
//
// Block layout: every m x m block row-major with row stride m; super-blocks with fewer than m / 6 cameras are padded
// with identity rows (zero coupling, zero right-hand side).
#pragma once
#include "ba_banded.h"
#include "ba_dense.h"

// -DBCR_CLOCKS: wall_clock64 cuts (10 ns ticks) inside the kernels of the chain, printed by one thread (a debug build,
// tools/bcr_clocks.sh: never the library that ships)
#ifdef BCR_CLOCKS
#define BCR_CUT(i) do { if (threadIdx.x == 0) bcr_cut[i] = wall_clock64(); } while (0)
#define BCR_CUTS_DECL long long bcr_cut[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define BCR_CUTS_PRINT(name, lvl, who) do { if (threadIdx.x == 0 && blockIdx.x == (who)) printf("%s level %d: %lld %lld %lld %lld %lld %lld (x10 ns)\n", name, lvl, bcr_cut[1] - bcr_cut[0], bcr_cut[2] - bcr_cut[1], bcr_cut[3] - bcr_cut[2], bcr_cut[4] - bcr_cut[3], bcr_cut[5] - bcr_cut[4], bcr_cut[6] - bcr_cut[5]); } while (0)
#else
#define BCR_CUT(i) do {} while (0)
#define BCR_CUTS_DECL do {} while (0)
#define BCR_CUTS_PRINT(name, lvl, who) do {} while (0)
#endif

constexpr int BCR_THREADS = 256;
constexpr int BCR_BCAST_WAVE = 2 * 64;              // doubles per wave: two column buffers (bcr_reg_step)
constexpr int BCR_BCAST = 4 * BCR_BCAST_WAVE + 2;   // the workgroup's share of LDS for them (+ 2: room to align to 16 bytes)
constexpr int BCR_MAX_M = 48;      // 8 cameras per super-block
constexpr int BCR_MAX_LEVELS = 24;
constexpr int BCR_TAIL_ROWS = 64;   // the last max(2, BCR_TAIL_ROWS / m) blocks are solved densely by one workgroup (a step of its
                                    // block Cholesky costs ~1 us at 60 rows but ~4 us at 120: larger tails lose to the levels they replace)

struct BcrLevel {
  int n;        // blocks at this level
  int n_elim;   // blocks eliminated here: the odd ones (n / 2); 0 at the tail level (lv[n_levels])
  long D, E, r;            // offsets (doubles) of D [n][m*m], E [n][m*m], r [n][m]
  long LiT, Fa, Fb, y;     // offsets of the factors of the eliminated blocks [n_elim][...]
};

// The top back-substitution levels that k_bcr_tail runs itself, after its dense solve (a wave per eliminated block, x of
// the blocks involved in an LDS copy): a level of one to eight blocks is not worth a launch of its own.
constexpr int BCR_TOP_MAX = 4;
struct BcrTop {
  int n;                     // levels folded in (0: none), highest level first
  int xshift;                // the LDS copy of x is indexed by (level-0 block >> xshift): xshift = the lowest folded level
  int nx;                    // its blocks (= blocks of the lowest folded level)
  int level[BCR_TOP_MAX];
  BcrLevel lv[BCR_TOP_MAX];
};

struct BcrPlan {
  int m, n_levels, N;      // n_levels elimination levels lv[0 .. n_levels); lv[n_levels] holds the tail system (D, E, r only)
  long X;                  // offset of x [N][m], indexed by the level-0 block
  BcrLevel lv[BCR_MAX_LEVELS];
};

// ------------------------------------------------------------------------------------------------
// k_bcr_assemble: the damped S (upper 6x6 blocks) and -rhs into the level-0 arrays, which must be zero wherever this
// kernel does not write (it writes the same entries every time: the arrays are cleared once, when they are allocated).
//   kind[s] 0: diagonal block of a camera (lower triangle mirrored), 1: block inside a D (written twice, mirrored),
//           2: block of an E, as stored, 3: block of an E, transposed (the loop closure: block (N - 1, 0)).
//   dst[s]  offset of the target of entry (0, 0) (for kind 1: of the upper copy; dst2[s] the mirrored copy)
//   rdst[f] offset of the right-hand-side entries of free camera f;  pad[q]: offsets of padded diagonal entries (= 1).
__global__ __launch_bounds__(256) void k_bcr_assemble(DeviceProblem d, double* __restrict__ B, int m,
                                                      const long* __restrict__ dst, const long* __restrict__ dst2,
                                                      const unsigned char* __restrict__ kind, const long* __restrict__ rdst,
                                                      const long* __restrict__ pad, int n_pad, double lam, double min_diag,
                                                      int solver_mode) {
  if (solver_mode) {
    if (d.ctrl->done) return;
    lam = d.ctrl->lam;
  }
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nS = (long)d.nu * 36;
  if (t < nS) {
    const int s = (int)(t / 36), e = (int)(t - 36L * s), r = e / 6, c = e - 6 * r;
    double v = d.S[t];
    const int k = kind[s];
    double* blk = B + dst[s];
    if (k == 0) {
      if (c > r) return;
      if (r == c) v += lam * fmax(u_diag(d, d.urow[s], r), min_diag);
      blk[r * m + c] = v;
      blk[c * m + r] = v;
    } else if (k == 1) {
      blk[r * m + c] = v;
      B[dst2[s] + c * m + r] = v;
    } else if (k == 2) {
      blk[r * m + c] = v;
    } else {
      blk[c * m + r] = v;
    }
  } else if (t < nS + 6L * d.Cf) {
    const long p = t - nS;
    const int f = (int)(p / 6), e = (int)(p - 6L * f);
    B[rdst[f] + e] = -d.rhs[p];
  } else if (t < nS + 6L * d.Cf + n_pad) {
    B[pad[t - nS - 6L * d.Cf]] = 1.0;
  }
}

// L^-1 (lower, row-major 6x6) of the Cholesky factor of the SPD block at A (row stride `ld`, lower part used).
__device__ __forceinline__ bool chol6_inverse_ld(const double* A, int ld, double Li[36]) {
  double a[36];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) a[6 * i + j] = A[i * ld + j];
  return chol6_inverse(a, Li);
}

constexpr int BCR_EPT = (BCR_MAX_M * BCR_MAX_M + BCR_THREADS - 1) / BCR_THREADS;  // entries of a block per thread
// first 16-byte aligned double at or after p, both inside the 16-byte aligned dynamic LDS array `base` (index arithmetic
// only: a cast through an integer would lose the LDS address space and turn every access into a flat one)
__device__ __forceinline__ double* bcr_align16(double* base, double* p) { return base + (((p - base) + 1) & ~1L); }

// One item of a trailing update: out[0..6) -= sum_t a[t] * b[c * ld + t].  a, b and out point into the same LDS array,
// so the compiler must assume that a store to out changes b: every load is issued before the first store here (one
// LDS round trip per item instead of one per column).
__device__ __forceinline__ void bcr_update_item(const double* a, const double* b, double* out, int ld) {
  double x[6], bb[36], o[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) x[t] = a[t];
#pragma unroll
  for (int c = 0; c < 6; ++c)
#pragma unroll
    for (int t = 0; t < 6; ++t) bb[6 * c + t] = b[c * ld + t];
#pragma unroll
  for (int c = 0; c < 6; ++c) o[c] = out[c];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 6; ++t) s += x[t] * bb[6 * c + t];
    o[c] -= s;
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) out[c] = o[c];
}

// Round 3: the factorisation of T out of REGISTERS, without a single barrier or LDS access inside its loop.  The blocked
// version below walks m / 6 column blocks with two workgroup barriers, a redundant 6 x 6 factorisation per thread and
// two LDS round trips each: 8.2 of the 16 us of a k_bcr_level launch at m = 30 (wall_clock64 cuts), and that launch is
// the unit of the critical path of the whole reduced solve (log2 N of them in a row).  Here a lane holds one ROW of T:
// every wave keeps the m rows of D in its lanes 0 .. m - 1 (all four waves factor D redundantly: the other SIMDs would
// idle anyway) and a quarter of the 3 m + 1 rows below it in the lanes after them; column j of the factor reaches
// the other lanes through v_readlane (a scalar operand of the multiply-adds), one column at a time:
//     p = T[j][j];  T[.][j] *= 1 / sqrt(p);  T[.][c] -= T[.][j] T[c][j]  for c > j.
// The chain is m steps of (reciprocal square root, a row of independent multiply-adds); fully unrolled, row arrays in
// registers (m doubles per lane).  m <= 36: the 64 lanes of a wave must hold m + ceil((3 m + 1) / 4) rows.
// One elimination step of the register-resident factorisations below.  Column j - 1 of the factor (tp: this lane's
// entry; the whole column sits in `prev`, an LDS buffer that the wave wrote in the step before) updates the columns
// after j; column j gets its last update and its pivot chain.  The column reaches the other lanes in two ways: the ONE
// entry on the critical path (L[j][j - 1], then the pivot itself) through v_readlane, everything else as 16-byte LDS
// reads of a uniform address (a broadcast: two entries per instruction, issued beside the arithmetic and a whole step
// ahead of their use), because four v_readlane / s_nop / v_fma instructions per entry made the loop issue-bound
// (1.5 k instructions at ~5 cycles for the 24 columns of a level: 3.5 us measured; 15 us for the 48 of the tail).
constexpr int BCR_GAPS = 8;  // places in the pivot chain of a column where updates of other columns are issued
template <int M, int J>
__device__ __forceinline__ void bcr_reg_step(double (&row)[M], double2 (&bc)[M / 2], double& tp, bool& bad,
                                             double* __restrict__ cur, int lane) {
  // bc: column J - 1 of the factor, two entries per element (entries 2 k and 2 k + 1), requested from LDS at the end of
  // the step before; pairs from C0 / 2 on update the columns after J (the pair that starts at J holds the entry that
  // went in through v_readlane)
  constexpr int C0 = (J + 1) & ~1, K0 = C0 / 2, NP = J > 0 ? M / 2 - K0 : 0, PER = (NP + BCR_GAPS - 1) / BCR_GAPS;
  auto updates = [&](int gap) {  // the gap-th share of the column updates (compile-time bounds after unrolling)
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int k = K0 + gap * PER + u;
      if (k < M / 2) {
        if (2 * k > J) row[2 * k] = fma(-tp, bc[k].x, row[2 * k]);
        row[2 * k + 1] = fma(-tp, bc[k].y, row[2 * k + 1]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // The schedule is written out by hand, a fence after every link of the chain: left alone, the scheduler either sinks
  // every update to where its column is needed (column j then waits for j dependent multiply-adds) or, inside one
  // region per column, hoists the updates in front of the chain, where they wait for the LDS reads instead of hiding
  // the chain's latencies.
  // The chain of a column, as short as it gets: the pivot is row J's diagonal entry after its last update, formed from
  // TWO independent v_readlane pairs (the entry before the update and L[J][J - 1]) instead of update -> v_readlane; a
  // non-positive pivot is clamped with one v_max (no compare / select through the scalar unit) and counted off the
  // chain; 1 / sqrt(p) is the hardware estimate and ONE third-order step (e = 1 - p y^2; y += y e (1/2 + 3/8 e): full
  // fp64 from an estimate good to 2^-20 or better), five dependent operations instead of six.
  double p = readlane_f64(row[J], J);
  if constexpr (J > 0) {
    const double s = readlane_f64(tp, J);
    p = fma(-s, s, p);
    row[J] = fma(-tp, s, row[J]);
  }
  if (!(p > 0.0)) bad = true;
  p = fmax(p, 1e-300);
  __builtin_amdgcn_sched_barrier(0);
  double inv = __builtin_amdgcn_rsq(p);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NP > 0) updates(0);
  double a = p * inv;
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NP > 0) updates(1);
  const double e = fma(-a, inv, 1.0);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NP > 0) updates(2);
  a = fma(0.375, e, 0.5);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NP > 0) updates(3);
  a = a * e;
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NP > 0) updates(4);
  inv = fma(inv, a, inv);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NP > 0) { updates(5); updates(6); }
  const double t = row[J] * inv;
  row[J] = t;
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NP > 0) updates(7);  // (the last share: bc is about to be refilled with column J)
  cur[lane] = t;  // column J: for the broadcast reads of the next step, issued right here
  if constexpr (J + 1 < M) {
    constexpr int K1 = ((J + 2) & ~1) / 2;
    const double2* __restrict__ pv = reinterpret_cast<const double2*>(cur);
#pragma unroll
    for (int k = K1; k < M / 2; ++k) bc[k] = pv[k];
  }
  __builtin_amdgcn_sched_barrier(0);
  tp = t;
}
template <int M, int J = 0>
__device__ __forceinline__ void bcr_reg_steps(double (&row)[M], double2 (&bc)[M / 2], double& tp, bool& bad,
                                              double* __restrict__ bcast, int lane) {
  if constexpr (J < M) {
    bcr_reg_step<M, J>(row, bc, tp, bad, bcast + (J & 1) * 64, lane);
    bcr_reg_steps<M, J + 1>(row, bc, tp, bad, bcast, lane);
  }
}

template <int M>
__device__ __forceinline__ bool bcr_factor_T_regs(double* T, int ld, int tid, double* bcast, const double* sub) {
  static_assert(M % 2 == 0, "the broadcast reads take the columns two at a time");
  constexpr int EXTRA = 3 * M + 1, EPW = (EXTRA + 3) / 4;  // rows below D, and their share per wave
  static_assert(M + EPW <= 64, "a wave holds the rows of D and its share of the rows below in its 64 lanes");
  const int lane = tid & 63, wv = tid >> 6;
  const int e = wv * EPW + (lane - M);                               // index of this lane's row among the rows below D
  const bool mine = lane >= M && lane - M < EPW && e < EXTRA;        // it has one
  const int rho = lane < M ? lane : (mine ? M + e : 0);               // the row of T in this lane (idle lanes: row 0, never written back)
  double row[M];
#pragma unroll
  for (int c = 0; c < M; ++c) row[c] = T[rho * ld + c];
  if (sub != nullptr && lane < M) {  // a part of D that another wave left beside T (k_bcr_level, m = 24)
#pragma unroll
    for (int c = 0; c < M; ++c) row[c] -= sub[rho * ld + c];
  }
  bool bad = false;
  double tp = 0.0;
  double2 bc[M / 2];
  bcr_reg_steps<M>(row, bc, tp, bad, bcast + wv * BCR_BCAST_WAVE, lane);  // (every wave its own two column buffers)
  if (mine) {
#pragma unroll
    for (int c = 0; c < M; ++c) T[rho * ld + c] = row[c];
  }
  __syncthreads();
  return bad;
}

// The block Cholesky of the tall matrix T = [D ; Ea ; Eb^T ; I ; r^T] ((4 m + 1) x m in LDS, row stride ld): see
// k_bcr_factor.  Called by every thread of the workgroup after a barrier that made T visible; returns with a barrier.
__device__ __forceinline__ bool bcr_factor_T(double* T, int m, int ld, int tid, double* bcast, const double* sub = nullptr) {
  switch (m) {  // (the rows below D come back in T; the rows of D themselves are not needed afterwards)
    case 12: return bcr_factor_T_regs<12>(T, ld, tid, bcast, sub);
    case 18: return bcr_factor_T_regs<18>(T, ld, tid, bcast, sub);
    case 24: return bcr_factor_T_regs<24>(T, ld, tid, bcast, sub);
    case 30: return bcr_factor_T_regs<30>(T, ld, tid, bcast, sub);
    case 36: return bcr_factor_T_regs<36>(T, ld, tid, bcast, sub);
    default: break;
  }
  const int nb = m / 6, R = 4 * m + 1;
  bool bad = false;
  for (int j = 0; j < nb; ++j) {
    const int c0 = 6 * j;
    double Li[36];
    if (!chol6_inverse_ld(T + c0 * ld + c0, ld, Li)) bad = true;  // (the panel below never writes the diagonal block)
    // panel: row <- row * L_jj^-T for the rows below the diagonal block (D rows > block j, all coupling rows, the
    // identity rows that can be non-zero here, the right-hand side)
    for (int rho = tid; rho < R; rho += BCR_THREADS) {
      const bool live = rho < m ? rho >= c0 + 6 : (rho >= 3 * m && rho < 4 * m ? rho - 3 * m < c0 + 6 : true);
      if (!live) continue;
      double* row = T + rho * ld + c0;
      double x[6], o[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) x[t] = row[t];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t <= c; ++t) s += x[t] * Li[6 * c + t];
        o[c] = s;
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) row[c] = o[c];
    }
    __syncthreads();
    // trailing update: T[rho][c] -= sum_t T[rho][c0 + t] * L[c][c0 + t] for the columns c right of the panel;
    // item = (row rho, column block kb)
    const int nbc = nb - 1 - j;
    for (int w = tid; w < R * nbc; w += BCR_THREADS) {
      const int rho = w / nbc, kb = j + 1 + (w - rho * nbc), cb = 6 * kb;
      if (rho < m) { if (rho < cb) continue; }                                  // D: blocks on or below the diagonal
      else if (rho >= 3 * m && rho < 4 * m && rho - 3 * m >= c0 + 6) continue;  // identity rows still zero here
      bcr_update_item(T + rho * ld + c0, T + cb * ld + c0, T + rho * ld + cb, ld);
    }
    __syncthreads();
  }
  return bad;
}

// Fa, Fb, L^-T, y of eliminated block q of a level, out of the factored T.
__device__ __forceinline__ void bcr_store_factors(double* __restrict__ B, const BcrLevel& lv, int q, const double* T, int m,
                                                  int ld, int tid) {
  const int mm = m * m;
  double* __restrict__ Fa = B + lv.Fa + (long)q * mm;
  double* __restrict__ Fb = B + lv.Fb + (long)q * mm;
  double* __restrict__ LiT = B + lv.LiT + (long)q * mm;
  for (int e = tid; e < mm; e += BCR_THREADS) {
    const int r = e / m, c = e - r * m;
    Fa[e] = T[(m + r) * ld + c];
    Fb[e] = T[(2 * m + r) * ld + c];
    // L^-T goes out TRANSPOSED (entry [r][c] of the stored block is L^-T[c][r] = L^-1[r][c]): the back-substitution
    // keeps row `lane` of L^-T in lane `lane`, and reads it with coalesced loads this way
    LiT[e] = r >= c ? T[(3 * m + c) * ld + r] : 0.0;
  }
  if (tid < m) B[lv.y + (long)q * m + tid] = T[4 * m * ld + tid];
}



// ------------------------------------------------------------------------------------------------
// k_bcr_factor: one workgroup per eliminated block i of a level.  The tall matrix
//     T = [ D_i ; E_{i-1} ; E_i^T ; I ; r_i^T ]        ((4 m + 1) x m, in LDS)
// goes through a right-looking block Cholesky of its top m x m part, 6 columns at a time (every thread factors the
// 6x6 diagonal block itself, as k_banded_factor does; a thread then scales one row of the panel; in the trailing
// update a thread takes one row and one 6-wide column block: 6 panel entries and a 6x6 block of L out of LDS for 36
// multiply-adds): what is left below D_i is Fa = E_{i-1} L^-T, Fb = E_i^T L^-T, L^-T and y^T = (L^-1 r_i)^T.
// All global loads are issued before the first use (one round trip, not one per loop iteration).
// (Levels have n >= 3 blocks: smaller systems go to k_bcr_tail.)
__global__ __launch_bounds__(BCR_THREADS) void k_bcr_factor(double* __restrict__ B, BcrLevel lv, int m, const LmCtrl* ctrl,
                                                            double* __restrict__ badv) {
  extern __shared__ __align__(16) double lds[];
  // (the "solve is over" flag is requested with the first round of loads and looked at after it: a launch that follows
  // termination still exits before its first store, and a live one does not pay a round trip of its own for the flag)
  const int done = ctrl ? ctrl->done : 0;
  // (the level's offsets arrive as kernel arguments, by value: indexing a plan structure in the argument block with
  // the level number cost every launch of the chain a dependent scalar load before its first global load)
  const int ld = m + 1, mm = m * m;
  const int tid = threadIdx.x;
  const int q = blockIdx.x;                    // q-th eliminated block
  const int i = 2 * q + 1;                     // its index at this level
  double* T = lds;                             // [4 m + 1][ld]
  const double* __restrict__ Di = B + lv.D + (long)i * mm;
  const double* __restrict__ Ea = B + lv.E + (long)(i - 1) * mm;   // block (i - 1, i)
  const double* __restrict__ Eb = B + lv.E + (long)i * mm;         // block (i, i + 1)
  {
    double vd[BCR_EPT], va[BCR_EPT], vb[BCR_EPT];
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int e = tid + u * BCR_THREADS;
      vd[u] = va[u] = vb[u] = 0.0;
      if (e < mm) { vd[u] = Di[e]; va[u] = Ea[e]; vb[u] = Eb[e]; }
    }
    const double rv = tid < m ? B[lv.r + (long)i * m + tid] : 0.0;
    if (done) return;
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int e = tid + u * BCR_THREADS;
      if (e < mm) {
        const int r = e / m, c = e - r * m;
        T[r * ld + c] = vd[u];
        T[(3 * m + r) * ld + c] = r == c ? 1.0 : 0.0;
        T[(m + r) * ld + c] = va[u];          // row r of block i - 1, column c of block i
        T[(2 * m + c) * ld + r] = vb[u];      // E_i^T: E_i[r][c] is row c of block i + 1, column r of block i
      }
    }
    if (tid < m) T[4 * m * ld + tid] = rv;
  }
  __syncthreads();
  const bool bad = bcr_factor_T(T, m, ld, tid, bcr_align16(lds, T + (4 * m + 1) * ld));
  bcr_store_factors(B, lv, q, T, m, ld, tid);
  if (bad && tid == 0) atomic_add_f64(badv, 1.0);
}

// ------------------------------------------------------------------------------------------------
// Products of two m x m blocks held row-major in LDS (row stride ld), C = X Y^T, one 16 x 16 tile per wave and call on
// v_mfma_f64_16x16x4_f64.  A thread-per-2x2-tile loop reads 1 LDS operand per multiply-add (a workgroup's four waves
// share one 128 B / clk LDS port: 5.4 us for the products of a level at m = 30, measured with wall_clock64); the
// matrix instruction reads 2 operands per lane for 16 multiply-adds.
//   A operand: lane supplies X[i0 + (lane & 15)][k0 + (lane >> 4)];  B operand: Y[j0 + (lane & 15)][k0 + (lane >> 4)]
//   result:    acc[q] = C[i0 + (lane >> 4) + 4 q][j0 + (lane & 15)]
// Rows beyond m only feed outputs that the caller drops (the row index is clamped to stay inside the block); columns
// k >= m (m is a multiple of 6, not of 4) are fed as zeros.
// A wave runs FOUR tiles at a time as independent accumulator chains, interleaved k group by k group (one chain alone
// waits out the latency of every matrix instruction: 143 ns per instruction measured).  No branch inside the loop (a
// conditional matrix instruction makes the compiler copy the accumulators out after each one): a slot that is off
// multiplies zeros.
struct BcrTile {
  const double *X, *Y;  // blocks in LDS
  int i0, j0;           // first row of X / of Y (= column of C) of the tile
  int kind;             // 0: off, else the caller's tag (wave-uniform)
};
constexpr int BCR_CHAINS = 4;
__device__ __forceinline__ void bcr_tiles_xyt(const BcrTile (&job)[BCR_CHAINS], int m, int ld, int lane, dvec4 (&acc)[BCR_CHAINS]) {
  const int l15 = lane & 15, l4 = lane >> 4;
  const double *xr[BCR_CHAINS], *yr[BCR_CHAINS];
#pragma unroll
  for (int s = 0; s < BCR_CHAINS; ++s) {
    acc[s] = dvec4{0.0, 0.0, 0.0, 0.0};
    xr[s] = job[s].X + min(job[s].i0 + l15, m - 1) * ld;
    yr[s] = job[s].Y + min(job[s].j0 + l15, m - 1) * ld;
  }
  double a[BCR_CHAINS], b[BCR_CHAINS];
#pragma unroll
  for (int s = 0; s < BCR_CHAINS; ++s) { a[s] = xr[s][l4]; b[s] = yr[s][l4]; }   // m >= 6 > l4
  for (int k0 = 0; k0 < m; k0 += 4) {
    // the operands of the next k group are requested before this group's matrix instructions are issued
    const bool in = k0 + l4 < m;
    const int kn = k0 + 4 + l4 < m ? k0 + 4 + l4 : 0;
    double an[BCR_CHAINS], bn[BCR_CHAINS];
#pragma unroll
    for (int s = 0; s < BCR_CHAINS; ++s) { an[s] = xr[s][kn]; bn[s] = yr[s][kn]; }
#pragma unroll
    for (int s = 0; s < BCR_CHAINS; ++s) {
      const bool use = in && job[s].kind != 0;
      acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(use ? a[s] : 0.0, use ? b[s] : 0.0, acc[s], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < BCR_CHAINS; ++s) { a[s] = an[s]; b[s] = bn[s]; }
  }
}
// q-th tile of the lower triangle of an MT x MT grid of tiles -> (ti, tj), tj <= ti
__device__ __forceinline__ void bcr_lower_tile(int q, int& ti, int& tj) {
  ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= q) ++ti;
  tj = q - ti * (ti + 1) / 2;
}
constexpr int BCR_MT = (BCR_MAX_M + 15) / 16;                       // tiles per block side
constexpr int BCR_LPW = (BCR_MT * (BCR_MT + 1) / 2 + 3) / 4;        // lower tiles per wave (4 waves)
constexpr int BCR_FPW1 = (BCR_MT * BCR_MT + 3) / 4;                 // tiles of one full block per wave
constexpr int BCR_FPW2 = (2 * BCR_MT * BCR_MT + 3) / 4;             // tiles of two full blocks per wave
constexpr int BCR_ROUNDS1 = (2 * BCR_LPW + BCR_FPW1 + BCR_CHAINS - 1) / BCR_CHAINS;
constexpr int BCR_ROUNDS2 = (2 * BCR_LPW + BCR_FPW2 + BCR_CHAINS - 1) / BCR_CHAINS;
static_assert(BCR_THREADS == 256, "the tile schedule below assumes four waves");

// ------------------------------------------------------------------------------------------------
// m = 24 (tracks of five: four cameras per super-block): a whole 24 x 24 product C += X Y^T by ONE wave on the vector
// unit, a 3 x 3 tile of C per lane (lane = 8 ti + tj: rows 3 ti .. 3 ti + 2 of X against rows 3 tj .. of Y), two k at a
// time: six 16-byte LDS reads feed 18 multiply-adds, eight distinct rows per read instruction (stride 75 doubles: no
// bank conflict; the lanes of a row get a broadcast).  The fp64 matrix instruction needs 16-wide tiles: 24 = 16 + 8
// wastes 44 % of its work, and it issues once per ~140 cycles on this part: the four products of a level took 3.4 us on
// it (wall_clock64), with one wave per product here they take under 1 us.
__device__ __forceinline__ void bcr_gemm24_xyt(const double* X, const double* Y, int lane, double (&acc)[9]) {
  constexpr int M = 24, LD = M + 1;
  const double* xr = X + 3 * (lane >> 3) * LD;
  const double* yr = Y + 3 * (lane & 7) * LD;
#pragma unroll
  for (int k = 0; k < M; k += 2) {
    double xa[3][2], yb[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      xa[a][0] = xr[a * LD + k]; xa[a][1] = xr[a * LD + k + 1];
      yb[a][0] = yr[a * LD + k]; yb[a][1] = yr[a * LD + k + 1];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[3 * a + b] = fma(xa[a][1], yb[b][1], fma(xa[a][0], yb[b][0], acc[3 * a + b]));
  }
}

// ------------------------------------------------------------------------------------------------
// k_bcr_update: one workgroup per even block e of level l = block e / 2 of level l + 1 (formulas in the header).
// The products run as 16 x 16 tiles on the fp64 matrix instruction (bcr_tile_xyt), a tile per wave and turn; D' is
// symmetric: only its lower tiles are computed.
__global__ __launch_bounds__(BCR_THREADS, 2) void k_bcr_update(double* __restrict__ B, BcrLevel lv, BcrLevel nx, int m, const LmCtrl* ctrl) {
  extern __shared__ __align__(16) double lds[];
  const int done = ctrl ? ctrl->done : 0;  // looked at after the first round of loads (see k_bcr_factor)
  const int ld = m + 1, mm = m * m;
  const int tid = threadIdx.x;
  const int k = blockIdx.x, e = 2 * k, n = lv.n;
  // odd neighbours (indices into the factor arrays), -1: none
  int ql = -1, qr = -1;
  if (e >= 1) ql = (e - 1) >> 1; else if ((n & 1) == 0) ql = (n - 1) >> 1;   // block 0's left neighbour is n - 1
  if (e + 1 < n) qr = (e + 1) >> 1;
  double* FbL = lds;                 // Fb of the left odd block  [m][ld]  (rows: block e)
  double* FaR = FbL + m * ld;        // Fa of the right odd block [m][ld]  (rows: block e)
  double* FbR = FaR + m * ld;        // Fb of the right odd block [m][ld]  (rows: block e + 2)
  double* yL = FbR + m * ld;         // [m]
  double* yR = yL + m;               // [m]
  double* Dn_s = yR + m;             // [m][ld] D' staged for the mirrored write-out
  const double* __restrict__ De = B + lv.D + (long)e * mm;
  const double* __restrict__ Ee = B + lv.E + (long)e * mm;
  const int lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int MT = (m + 15) >> 4, nlow = MT * (MT + 1) / 2, nfull = MT * MT;
  double de[BCR_LPW][4], re = 0.0;
  const int rrow = tid - 3 * 64;  // r' is formed by the last wave, which gets the fewest tiles: row rrow by lane rrow
  {
    double v1[BCR_EPT], v2[BCR_EPT], v3[BCR_EPT];
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      v1[u] = v2[u] = v3[u] = 0.0;
      if (w < mm) {
        if (ql >= 0) v1[u] = B[lv.Fb + (long)ql * mm + w];
        if (qr >= 0) { v2[u] = B[lv.Fa + (long)qr * mm + w]; v3[u] = B[lv.Fb + (long)qr * mm + w]; }
      }
    }
    double y1 = 0.0, y2 = 0.0;
    if (tid < m) {
      if (ql >= 0) y1 = B[lv.y + (long)ql * m + tid];
      if (qr >= 0) y2 = B[lv.y + (long)qr * m + tid];
    }
    if (rrow >= 0 && rrow < m) re = B[lv.r + (long)e * m + rrow];
#pragma unroll
    for (int p = 0; p < BCR_LPW; ++p) {  // the entries of D_e that this thread's tiles update, in flight with the rest
      int ti, tj;
      bcr_lower_tile(wv + 4 * p, ti, tj);
      const int c = 16 * tj + l15;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * ti + l4 + 4 * q;
        de[p][q] = (wv + 4 * p < nlow && r < m && c < m) ? De[r * m + c] : 0.0;
      }
    }
    if (done) return;
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      if (w < mm) {
        const int r = w / m, c = w - r * m;
        FbL[r * ld + c] = v1[u]; FaR[r * ld + c] = v2[u]; FbR[r * ld + c] = v3[u];
      }
    }
    if (tid < m) { yL[tid] = y1; yR[tid] = y2; }
  }
  __syncthreads();
  double* __restrict__ Dn = B + nx.D + (long)k * mm;
  double* __restrict__ En = B + nx.E + (long)k * mm;
  const bool copy_E = qr < 0;  // n odd, e = n - 1: blocks n - 1 and 0 stay coupled directly
  {
    // the jobs of this wave, in order: the two products of each of its lower tiles of D' (tiles wv, wv + 4, ...), then
    // its tiles of E' = block (e, e + 2) = -FaR FbR^T, which continue the round robin over the waves
    const int np = wv < nlow ? (nlow - wv + 3) / 4 : 0, q0 = (wv + 4 - nlow % 4) % 4;
    const int nq = copy_E || q0 >= nfull ? 0 : (nfull - q0 + 3) / 4, njobs = 2 * np + nq;
    auto make_job = [&](int n) {
      BcrTile j{FbL, FbL, 0, 0, 0};
      if (n < 2 * np) {
        int ti, tj;
        bcr_lower_tile(wv + 4 * (n >> 1), ti, tj);
        j = (n & 1) ? BcrTile{FaR, FaR, 16 * ti, 16 * tj, 1} : BcrTile{FbL, FbL, 16 * ti, 16 * tj, 1};
      } else if (n < njobs) {
        const int q = q0 + 4 * (n - 2 * np), ti = q / MT, tj = q - ti * MT;
        j = BcrTile{FaR, FbR, 16 * ti, 16 * tj, 2};
      }
      return j;
    };
#pragma unroll
    for (int rd = 0; rd < BCR_ROUNDS1; ++rd) {
      if (BCR_CHAINS * rd >= njobs) break;
      BcrTile job[BCR_CHAINS];
      dvec4 acc[BCR_CHAINS];
#pragma unroll
      for (int s = 0; s < BCR_CHAINS; ++s) job[s] = make_job(BCR_CHAINS * rd + s);
      bcr_tiles_xyt(job, m, ld, lane, acc);
#pragma unroll
      for (int s = 0; s < BCR_CHAINS; ++s) {
        const int n = BCR_CHAINS * rd + s, c = job[s].j0 + l15;
        if (job[s].kind == 1 && (s & 1) == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = job[s].i0 + l4 + 4 * q;
            if (r < m && c < m) Dn_s[r * ld + c] = de[(n >> 1) < BCR_LPW ? (n >> 1) : 0][q] - (acc[s][q] + acc[s | 1][q]);
          }
        } else if (job[s].kind == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = job[s].i0 + l4 + 4 * q;
            if (r < m && c < m) En[r * m + c] = -acc[s][q];
          }
        }
      }
    }
    if (copy_E)
      for (int w = tid; w < mm; w += BCR_THREADS) En[w] = Ee[w];
  }
  if (rrow >= 0 && rrow < m) {
    const int r = rrow;
    double s = 0.0;
    for (int t0 = 0; t0 < m; t0 += 6) {  // m is a multiple of 6
#pragma unroll
      for (int u = 0; u < 6; ++u) { const int t = t0 + u; s += FbL[r * ld + t] * yL[t] + FaR[r * ld + t] * yR[t]; }
    }
    B[nx.r + (long)k * m + r] = re - s;
  }
  __syncthreads();
  for (int w = tid; w < mm; w += BCR_THREADS) {  // lower triangle mirrored
    const int r = w / m, c = w - r * m;
    Dn[w] = c <= r ? Dn_s[r * ld + c] : Dn_s[c * ld + r];
  }
}

// ------------------------------------------------------------------------------------------------
// k_bcr_backsolve: one workgroup per eliminated block i of a level: x_i = L^-T (y - Fa^T x_{i-1} - Fb^T x_{i+1}); x of
// the even neighbours is already in X (indexed by the level-0 block: block i of level l is block i << l of level 0).
// Also scatters x_i to the camera step dc ([Cf][6]; cam0[blk] = first free camera of a level-0 block, cnt[blk] its
// camera count).  The three factor blocks come in with one round of loads.
__global__ __launch_bounds__(BCR_THREADS) void k_bcr_backsolve(double* __restrict__ B, BcrLevel lv, int m, long Xoff, int level,
                                                               const int* __restrict__ cam0, const int* __restrict__ cnt,
                                                               const LmCtrl* ctrl, double* __restrict__ dc) {
  __shared__ double sF[3 * BCR_MAX_M * (BCR_MAX_M + 1)];
  __shared__ double xa[BCR_MAX_M], xb[BCR_MAX_M], tt[BCR_MAX_M], part[4][BCR_MAX_M];
  BCR_CUTS_DECL;
  BCR_CUT(0);
  const int done = ctrl ? ctrl->done : 0;  // looked at after the round of loads (see k_bcr_factor)
  const int ld = m + 1, mm = m * m, tid = threadIdx.x;
  const int q = blockIdx.x, n = lv.n;
  const int i = 2 * q + 1;
  const long o = (long)i << level;
  double* __restrict__ X = B + Xoff;
  double* sFa = sF;
  double* sFb = sFa + m * ld;
  double* sLi = sFb + m * ld;
  {
    double v1[BCR_EPT], v2[BCR_EPT], v3[BCR_EPT];
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      v1[u] = v2[u] = v3[u] = 0.0;
      if (w < mm) {
        v1[u] = B[lv.Fa + (long)q * mm + w]; v2[u] = B[lv.Fb + (long)q * mm + w]; v3[u] = B[lv.LiT + (long)q * mm + w];
      }
    }
    double va = 0.0, vb = 0.0, vy = 0.0;
    if (tid < m) {
      const long oa = (long)(i - 1) << level, ob = (long)(i + 1 == n ? 0 : i + 1) << level;
      va = X[oa * m + tid];
      vb = X[ob * m + tid];
      vy = B[lv.y + (long)q * m + tid];
    }
    if (done) return;
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      if (w < mm) {
        const int r = w / m, c = w - r * m;
        sFa[r * ld + c] = v1[u]; sFb[r * ld + c] = v2[u]; sLi[c * ld + r] = v3[u];  // (L^-T is stored transposed)
      }
    }
    if (tid < m) { xa[tid] = va; xb[tid] = vb; tt[tid] = vy; }
  }
  __syncthreads();
  BCR_CUT(1);  // loads
  // t_c = y_c - sum_r Fa[r][c] xa[r] + Fb[r][c] xb[r]: four row groups per column, folded in a fixed order
  const int c = tid & 63, g = tid >> 6;
  if (c < m) {
    double s = 0.0;
#pragma unroll 4
    for (int r = g; r < m; r += 4) s += sFa[r * ld + c] * xa[r] + sFb[r * ld + c] * xb[r];
    part[g][c] = s;
  }
  __syncthreads();
  if (tid < m) tt[tid] -= (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
  __syncthreads();
  if (tid < m) {
    double x = 0.0;
    for (int c0 = 0; c0 < m; c0 += 6) {  // L^-T is stored with its zeros: whole rows, six loads in flight at a time
#pragma unroll
      for (int u = 0; u < 6; ++u) x += sLi[tid * ld + c0 + u] * tt[c0 + u];
    }
    X[o * m + tid] = x;
    if (tid < 6 * cnt[o]) dc[6L * cam0[o] + tid] = x;
  }
  BCR_CUT(2);
  BCR_CUTS_PRINT("k_bcr_backsolve", level, 0);
}

// ------------------------------------------------------------------------------------------------
// The same back-substitution by ONE WAVE per eliminated block, without LDS and without a barrier (round 3; m <= 36):
// lane c holds COLUMN c of Fa and Fb (coalesced loads straight into registers) and row c of L^-T (stored transposed for
// exactly this), all requested in one round; x of the two neighbours and y sit one entry per lane and travel by
// v_readlane:  t_c = y_c - sum_r Fa[r][c] xa_r - sum_r Fb[r][c] xb_r,   x_c = sum_k L^-T[c][k] t_k  (fixed order).
// XS: the neighbours' x come from an LDS copy indexed by (level-0 block >> xshift) (the levels that k_bcr_tail folds
// into its own launch), and x_i goes there as well as to X.
template <int M, bool XS>
__device__ __forceinline__ void bcr_backsolve_wave(double* __restrict__ B, const BcrLevel& lv, long Xoff, int level, int q,
                                                   int lane, const int* __restrict__ cam0, const int* __restrict__ cnt,
                                                   double* __restrict__ dc, double* xs, int xshift, int done = 0) {
  constexpr int mm = M * M;
  const int n = lv.n, i = 2 * q + 1;
  const long o = (long)i << level, oa = (long)(i - 1) << level, ob = (long)(i + 1 == n ? 0 : i + 1) << level;
  const int c = lane < M ? lane : 0;
  double* __restrict__ X = B + Xoff;
  const double* __restrict__ Fa = B + lv.Fa + (long)q * mm + c;
  const double* __restrict__ Fb = B + lv.Fb + (long)q * mm + c;
  const double* __restrict__ Lt = B + lv.LiT + (long)q * mm + c;
  double fa[M], fb[M], li[M];
#pragma unroll
  for (int r = 0; r < M; ++r) { fa[r] = Fa[r * M]; fb[r] = Fb[r * M]; li[r] = Lt[r * M]; }  // li[k] = L^-T[c][k]
  const double y = B[lv.y + (long)q * M + c];
  const double xa = XS ? xs[(oa >> xshift) * M + c] : X[oa * M + c];
  const double xb = XS ? xs[(ob >> xshift) * M + c] : X[ob * M + c];
  double t0 = y, t1 = 0.0;
#pragma unroll
  for (int r = 0; r < M; ++r) {
    t0 = fma(-fa[r], readlane_f64(xa, r), t0);
    t1 = fma(-fb[r], readlane_f64(xb, r), t1);
  }
  const double t = t0 + t1;
  double x0 = 0.0, x1 = 0.0;
#pragma unroll
  for (int k = 0; k < M; k += 2) {
    x0 = fma(li[k], readlane_f64(t, k), x0);
    x1 = fma(li[k + 1], readlane_f64(t, k + 1), x1);
  }
  const double x = x0 + x1;
  if (lane < M && !done) {  // (a launch that follows termination stores nothing)
    X[o * M + lane] = x;
    if (XS) xs[(o >> xshift) * M + lane] = x;
    if (lane < 6 * cnt[o]) dc[6L * cam0[o] + lane] = x;
  }
}
template <bool XS>
__device__ __forceinline__ bool bcr_backsolve_wave_m(int m, double* __restrict__ B, const BcrLevel& lv, long Xoff, int level,
                                                     int q, int lane, const int* __restrict__ cam0, const int* __restrict__ cnt,
                                                     double* __restrict__ dc, double* xs, int xshift, int done = 0) {
  switch (m) {
    case 12: bcr_backsolve_wave<12, XS>(B, lv, Xoff, level, q, lane, cam0, cnt, dc, xs, xshift, done); return true;
    case 18: bcr_backsolve_wave<18, XS>(B, lv, Xoff, level, q, lane, cam0, cnt, dc, xs, xshift, done); return true;
    case 24: bcr_backsolve_wave<24, XS>(B, lv, Xoff, level, q, lane, cam0, cnt, dc, xs, xshift, done); return true;
    case 30: bcr_backsolve_wave<30, XS>(B, lv, Xoff, level, q, lane, cam0, cnt, dc, xs, xshift, done); return true;
    case 36: bcr_backsolve_wave<36, XS>(B, lv, Xoff, level, q, lane, cam0, cnt, dc, xs, xshift, done); return true;
    default: return false;
  }
}
constexpr int BCR_WAVE_MAX_M = 36;  // block sizes that the wave-per-block back-substitution takes (multiples of 6 from 12)
__host__ __device__ inline bool bcr_wave_backsolve_ok(int m) { return m >= 12 && m <= BCR_WAVE_MAX_M && m % 6 == 0; }

// four eliminated blocks of a level per workgroup, a wave each
__global__ __launch_bounds__(BCR_THREADS) void k_bcr_backsolve_w(double* __restrict__ B, BcrLevel lv, int m, long Xoff, int level,
                                                                 const int* __restrict__ cam0, const int* __restrict__ cnt,
                                                                 const LmCtrl* ctrl, double* __restrict__ dc) {
  const int done = ctrl ? ctrl->done : 0;  // requested with the loads, looked at before the stores
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= lv.n_elim) return;
  (void)bcr_backsolve_wave_m<false>(m, B, lv, Xoff, level, q, threadIdx.x & 63, cam0, cnt, dc, nullptr, 0, done);
}

// ------------------------------------------------------------------------------------------------
// k_bcr_level: k_bcr_update of level l and k_bcr_factor of level l + 1 in ONE launch (one workgroup per block k of level
// l + 1 = even block e = 2 k of level l).  A block that is even again at level l + 1 only needs its updated D', r' (and
// E' where it is the last block of an odd count) in memory.  A block that is ODD at level l + 1 is eliminated right
// here: besides D', r' it forms the two couplings it needs from the factors of ITS OWN odd neighbours at level l,
//     block (k - 1, k) = -Fa_{e-1} Fb_{e-1}^T,     block (k, k + 1) = -Fa_{e+1} Fb_{e+1}^T   (or E_e where e + 1 does not exist),
// lays them out as the tall matrix of k_bcr_factor in LDS and factors it: no round trip through memory and one launch
// per level instead of two.
__global__ __launch_bounds__(BCR_THREADS, 2) void k_bcr_level(double* __restrict__ B, BcrLevel lv, BcrLevel nx, int m, int level, const LmCtrl* ctrl,
                                                           double* __restrict__ badv) {
  extern __shared__ __align__(16) double lds[];
  BCR_CUTS_DECL;
  BCR_CUT(0);
  const int done = ctrl ? ctrl->done : 0;  // looked at after the first round of loads (see k_bcr_factor)
  const int ld = m + 1, mm = m * m;
  const int tid = threadIdx.x;
  const int k = blockIdx.x, e = 2 * k, n = lv.n;
  const bool elim = (k & 1) != 0;  // eliminated at level l + 1
  int ql = -1, qr = -1;
  if (e >= 1) ql = (e - 1) >> 1; else if ((n & 1) == 0) ql = (n - 1) >> 1;
  if (e + 1 < n) qr = (e + 1) >> 1;
  double* FbL = lds;                 // [m][ld] Fb of the left odd block  (rows: block e)
  double* FaR = FbL + m * ld;        // [m][ld] Fa of the right odd block (rows: block e)
  double* FbR = FaR + m * ld;        // [m][ld] Fb of the right odd block (rows: block e + 2)
  double* FaL = FbR + m * ld;        // [m][ld] Fa of the left odd block  (rows: block e - 2)      (elim only)
  double* yL = FaL + m * ld;         // [m]
  double* yR = yL + m;               // [m]
  double* T = yR + m;                // [(4 m + 1)][ld] (elim), or [m][ld] as the staging copy of D'
  const double* __restrict__ De = B + lv.D + (long)e * mm;
  const double* __restrict__ Ee = B + lv.E + (long)e * mm;
  const int lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int MT = (m + 15) >> 4, nlow = MT * (MT + 1) / 2, nfull = MT * MT;
  double de[BCR_LPW][4], re = 0.0;
  const bool m24 = m == 24;  // products on the vector unit, a wave per product (bcr_gemm24_xyt)
  double de9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double* P2s = bcr_align16(lds, T + (4 * m + 1) * ld) + BCR_BCAST;  // [m][ld] (m = 24 only)
  const int rrow = tid - 3 * 64;  // r' is formed by the last wave, which gets the fewest tiles: row rrow by lane rrow
  {
    double v1[BCR_EPT], v2[BCR_EPT], v3[BCR_EPT], v4[BCR_EPT];
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      v1[u] = v2[u] = v3[u] = v4[u] = 0.0;
      if (w < mm) {
        if (ql >= 0) { v1[u] = B[lv.Fb + (long)ql * mm + w]; if (elim) v4[u] = B[lv.Fa + (long)ql * mm + w]; }
        if (qr >= 0) { v2[u] = B[lv.Fa + (long)qr * mm + w]; v3[u] = B[lv.Fb + (long)qr * mm + w]; }
      }
    }
    double y1 = 0.0, y2 = 0.0;
    if (tid < m) {
      if (ql >= 0) y1 = B[lv.y + (long)ql * m + tid];
      if (qr >= 0) y2 = B[lv.y + (long)qr * m + tid];
    }
    if (rrow >= 0 && rrow < m) re = B[lv.r + (long)e * m + rrow];
    // the entries of D_e that this thread's tiles update: in flight with the factor blocks, not a second round trip
    // after the products
    if (m24) {  // the 3 x 3 tile of D_e that lane (ti, tj) of wave 0 updates (bcr_gemm24_xyt's layout)
      if (wv == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) de9[3 * a + b2] = De[(3 * (lane >> 3) + a) * 24 + 3 * (lane & 7) + b2];
      }
    } else {
#pragma unroll
      for (int p = 0; p < BCR_LPW; ++p) {
        int ti, tj;
        bcr_lower_tile(wv + 4 * p, ti, tj);
        const int c = 16 * tj + l15;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 16 * ti + l4 + 4 * q;
          de[p][q] = (wv + 4 * p < nlow && r < m && c < m) ? De[r * m + c] : 0.0;
        }
      }
    }
    if (done) return;
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      if (w < mm) {
        const int r = w / m, c = w - r * m;
        FbL[r * ld + c] = v1[u]; FaR[r * ld + c] = v2[u]; FbR[r * ld + c] = v3[u]; FaL[r * ld + c] = v4[u];
      }
    }
    if (tid < m) { yL[tid] = y1; yR[tid] = y2; }
  }
  __syncthreads();
  BCR_CUT(1);  // loads
  double* __restrict__ Dn = B + nx.D + (long)k * mm;
  double* __restrict__ En = B + nx.E + (long)k * mm;
  const bool copy_E = qr < 0;                            // n odd, e = n - 1: blocks n - 1 and 0 stay coupled directly
  const bool need_E = !elim && k == nx.n - 1;             // an even block's E' is only read where it is the last block of an
                                                         // odd count (its right neighbour is block 0: the coupling is copied)
  {
    if (m24) {
      // wave 0: D' = D_e - FbL FbL^T - FaR FaR^T;  wave 1: Ea' = -FaL FbL^T (elim) or E' = -FaR FbR^T (need_E);
      // wave 2: Eb'^T = -FbR FaR^T (elim);  wave 3: r' below.  A block without a neighbour has zeros in its place.
      const int r0 = 3 * (lane >> 3), c0 = 3 * (lane & 7);
      double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (wv == 0) {  // (the second product of D' is wave 3's: added in after the barrier below)
        bcr_gemm24_xyt(FbL, FbL, lane, acc);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) T[(r0 + a) * ld + c0 + b2] = de9[3 * a + b2] - acc[3 * a + b2];
      } else if (wv == 3) {  // FaR FaR^T, left beside T: whoever reads D' after the barrier subtracts it
        bcr_gemm24_xyt(FaR, FaR, lane, acc);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) P2s[(r0 + a) * ld + c0 + b2] = acc[3 * a + b2];
      } else if (wv == 1 && elim) {
        bcr_gemm24_xyt(FaL, FbL, lane, acc);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) T[(m + r0 + a) * ld + c0 + b2] = -acc[3 * a + b2];
      } else if (wv == 1 && need_E && !copy_E) {
        bcr_gemm24_xyt(FaR, FbR, lane, acc);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) En[(r0 + a) * m + c0 + b2] = -acc[3 * a + b2];
      } else if (wv == 2 && elim && !copy_E) {
        bcr_gemm24_xyt(FbR, FaR, lane, acc);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) T[(2 * m + r0 + a) * ld + c0 + b2] = -acc[3 * a + b2];
      }
    } else {
    // the jobs of this wave, in order: the two products of each of its lower tiles of D' = D_e - FbL FbL^T - FaR FaR^T
    // (tiles wv, wv + 4, ...), then, continuing the round robin over the waves, its tiles of
    //   elim:    Ea' = block (k - 1, k) = -FaL FbL^T  (tag 2)  and  Eb'^T = (block (k, k + 1))^T = -FbR FaR^T  (tag 3;
    //            E_e^T where block e + 1 does not exist)
    //   need_E:  E' = block (e, e + 2) = -FaR FbR^T  (tag 4; E_e where block e + 1 does not exist)
    const int ntask = elim ? (copy_E ? nfull : 2 * nfull) : (need_E && !copy_E ? nfull : 0);
    const int np = wv < nlow ? (nlow - wv + 3) / 4 : 0, q0 = (wv + 4 - nlow % 4) % 4;
    const int nq = q0 >= ntask ? 0 : (ntask - q0 + 3) / 4, njobs = 2 * np + nq;
    auto make_job = [&](int n) {
      BcrTile j{FbL, FbL, 0, 0, 0};
      if (n < 2 * np) {
        int ti, tj;
        bcr_lower_tile(wv + 4 * (n >> 1), ti, tj);
        j = (n & 1) ? BcrTile{FaR, FaR, 16 * ti, 16 * tj, 1} : BcrTile{FbL, FbL, 16 * ti, 16 * tj, 1};
      } else if (n < njobs) {
        const int q = q0 + 4 * (n - 2 * np);
        const bool second = q >= nfull;
        const int qq = second ? q - nfull : q, ti = qq / MT, tj = qq - ti * MT;
        if (!elim) j = BcrTile{FaR, FbR, 16 * ti, 16 * tj, 4};
        else if (second) j = BcrTile{FbR, FaR, 16 * ti, 16 * tj, 3};
        else j = BcrTile{FaL, FbL, 16 * ti, 16 * tj, 2};
      }
      return j;
    };
#pragma unroll
    for (int rd = 0; rd < BCR_ROUNDS2; ++rd) {
      if (BCR_CHAINS * rd >= njobs) break;
      BcrTile job[BCR_CHAINS];
      dvec4 acc[BCR_CHAINS];
#pragma unroll
      for (int s = 0; s < BCR_CHAINS; ++s) job[s] = make_job(BCR_CHAINS * rd + s);
      bcr_tiles_xyt(job, m, ld, lane, acc);
#pragma unroll
      for (int s = 0; s < BCR_CHAINS; ++s) {
        const int n = BCR_CHAINS * rd + s, c = job[s].j0 + l15;
        if (job[s].kind == 1 && (s & 1) == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = job[s].i0 + l4 + 4 * q;
            if (r < m && c < m) T[r * ld + c] = de[(n >> 1) < BCR_LPW ? (n >> 1) : 0][q] - (acc[s][q] + acc[s | 1][q]);
          }
        } else if (job[s].kind == 4) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = job[s].i0 + l4 + 4 * q;
            if (r < m && c < m) En[r * m + c] = -acc[s][q];
          }
        } else if (job[s].kind >= 2) {
          double* out = T + (job[s].kind == 3 ? 2 * m : m) * ld;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = job[s].i0 + l4 + 4 * q;
            if (r < m && c < m) out[r * ld + c] = -acc[s][q];
          }
        }
      }
    }
    }
    if (elim) {
      for (int w = tid; w < mm; w += BCR_THREADS) {
        const int r = w / m, c = w - r * m;
        T[(3 * m + r) * ld + c] = r == c ? 1.0 : 0.0;
        if (copy_E) T[(2 * m + r) * ld + c] = Ee[c * m + r];
      }
    } else if (need_E && copy_E) {
      for (int w = tid; w < mm; w += BCR_THREADS) En[w] = Ee[w];
    }
  }
  if (rrow >= 0 && rrow < m) {
    const int r = rrow;
    double s = 0.0;
    for (int t0 = 0; t0 < m; t0 += 6) {  // m is a multiple of 6
#pragma unroll
      for (int u = 0; u < 6; ++u) { const int t = t0 + u; s += FbL[r * ld + t] * yL[t] + FaR[r * ld + t] * yR[t]; }
    }
    const double v = re - s;
    if (elim) T[4 * m * ld + r] = v; else B[nx.r + (long)k * m + r] = v;
  }
  __syncthreads();
  BCR_CUT(2);  // products
  if (!elim) {
    for (int w = tid; w < mm; w += BCR_THREADS) {  // lower triangle mirrored
      const int r = w / m, c = w - r * m;
      const int at = c <= r ? r * ld + c : c * ld + r;
      Dn[w] = m24 ? T[at] - P2s[at] : T[at];
    }
    return;
  }
  const bool bad = bcr_factor_T(T, m, ld, tid, bcr_align16(lds, T + (4 * m + 1) * ld), m24 ? P2s : nullptr);
  BCR_CUT(3);  // factor
  bcr_store_factors(B, nx, k >> 1, T, m, ld, tid);
  if (bad && tid == 0) atomic_add_f64(badv, 1.0);
  BCR_CUT(4);  // stores issued
  BCR_CUTS_PRINT("k_bcr_level", level, 1);
}

// The dense tail system out of registers (see bcr_factor_T_regs): lane r of wave 0 holds row r of the lower triangle
// (row ND: the right-hand side), ND columns; after the ND elimination steps the rows go back to LDS, every lane picks
// up a COLUMN of L and the backward sweep x = L^-T y runs as ND steps of (one v_readlane, one multiply-add per lane).
// Returns after a barrier with x in xs[0 .. ND).
template <int ND>
__device__ __forceinline__ bool bcr_tail_regs(double* T, int ld, double* xs, int tid, double* bcast) {
  static_assert(ND + 1 <= 64, "one row per lane");
  bool bad = false;
  if (tid < 64) {
    const int lane = tid, rho = lane <= ND ? lane : 0;
    double row[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) row[c] = T[rho * ld + c];
    double tp = 0.0;
    double2 bc[ND / 2];
    bcr_reg_steps<ND>(row, bc, tp, bad, bcast, lane);
    // rows back to LDS (ld is odd: conflict-free), then lane j takes column j of L and y_j
    if (lane <= ND) {
#pragma unroll
      for (int c = 0; c < ND; ++c) T[rho * ld + c] = row[c];
    }
    __builtin_amdgcn_wave_barrier();  // one wave: the LDS serves its accesses in order
    const int col = lane < ND ? lane : 0;
#pragma unroll
    for (int i = 0; i < ND; ++i) row[i] = T[i * ld + col];  // L[i][col] (i >= col; the rest is never used)
    double y = T[ND * ld + col];
    const double dinv = fast_rcp(T[col * ld + col]);
    double x = 0.0;
#pragma unroll
    for (int i = ND - 1; i >= 0; --i) {
      const double xi = readlane_f64(y * dinv, i);  // x_i = y_i / L_ii
      x = lane == i ? xi : x;
      y = fma(-row[i], xi, y);                       // y_j -= L[i][j] x_i for the columns j < i (later ones: consumed)
    }
    if (lane < ND) xs[lane] = x;
  }
  __syncthreads();
  return bad;
}

// ------------------------------------------------------------------------------------------------
// k_bcr_tail: the last n <= max(2, BCR_TAIL_ROWS / m) blocks (level `plan.n_levels`), one workgroup: the cyclic block
// tridiagonal system is laid out as ONE dense symmetric matrix of Nd = n m rows in LDS, with the right-hand side as row
// Nd (lower triangle: D_i on the diagonal, E_i^T below it, the loop closure E_{n-1} in the corner; n == 2: both
// couplings of blocks 0 and 1 land on the same block), factored by the block Cholesky loop of k_bcr_factor (the
// right-hand-side row comes out as y = L^-1 r), then solved backwards column block by column block
// (x_j = L_jj^-T y_j;  y_rho -= L[j][rho] x_j for the rows above).  Writes x of the tail blocks to X and to dc.
__global__ __launch_bounds__(BCR_THREADS) void k_bcr_tail(double* __restrict__ B, BcrLevel lv, int m, long Xoff, int level, const int* __restrict__ cam0,
                                                          const int* __restrict__ cnt, const LmCtrl* ctrl,
                                                          double* __restrict__ dc, double* __restrict__ badv, BcrTop top) {
  extern __shared__ __align__(16) double lds[];
  BCR_CUTS_DECL;
  BCR_CUT(0);
  const int done = ctrl ? ctrl->done : 0;  // looked at after the round of loads (see k_bcr_factor)
  const int mm = m * m, n = lv.n, Nd = n * m, ld = Nd + 1, nb = Nd / 6, R = Nd + 1;
  const int tid = threadIdx.x;
  // Two blocks of 24 rows (the tail of a trajectory with tracks of five): eliminate block 0 with the four-wave register
  // factorisation of the levels (bcr_factor_T: [D_0; C; .; I; r_0] -> [L_0; F = C L_0^-T; .; L_0^-T; y_0]), form
  // D_1 - F F^T on one wave and r_1 - F y_0 on another, factor that the same way, and substitute back with four
  // mat-vecs of 24: 8 us where the 48-row factorisation on ONE wave took 14 (wall_clock64).
  const bool two_blocked = n == 2 && m == 24;
  const int ldb = m + 1, Rb = 4 * m + 1;
  double* T = lds;                      // [R][ld]   (two_blocked: T1 [Rb][ldb] | T2 [Rb][ldb] | P2 [m][ldb])
  double* sLi = T + (long)R * ld;       // [nb][36] inverses of the diagonal blocks of L
  double* xs = two_blocked ? lds + 2L * Rb * ldb + (long)m * ldb : sLi + nb * 36;  // [Nd]
  if (two_blocked) {
    double* T1 = lds;
    double* T2 = T1 + (long)Rb * ldb;
    double* P2 = T2 + (long)Rb * ldb;
    double* bcast = bcr_align16(lds, xs + Nd);
    double d0[BCR_EPT], d1[BCR_EPT], e0[BCR_EPT], e1[BCR_EPT];
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      d0[u] = d1[u] = e0[u] = e1[u] = 0.0;
      if (w < mm) {
        const int r = w / m, c = w - r * m;
        d0[u] = B[lv.D + w]; d1[u] = B[lv.D + mm + w];
        e0[u] = B[lv.E + c * m + r];  // E_0^T: block (1, 0)
        e1[u] = B[lv.E + mm + w];     // E_1 = block (1, 0) as it is (the loop closure)
      }
    }
    const double rv = tid < Nd ? B[lv.r + tid] : 0.0;
    if (done) return;
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      if (w < mm) {
        const int r = w / m, c = w - r * m;
        const double id = r == c ? 1.0 : 0.0;
        T1[r * ldb + c] = d0[u]; T1[(m + r) * ldb + c] = e0[u] + e1[u]; T1[(2 * m + r) * ldb + c] = 0.0; T1[(3 * m + r) * ldb + c] = id;
        T2[r * ldb + c] = d1[u]; T2[(m + r) * ldb + c] = 0.0; T2[(2 * m + r) * ldb + c] = 0.0; T2[(3 * m + r) * ldb + c] = id;
      }
    }
    if (tid < m) T1[4 * m * ldb + tid] = rv;
    else if (tid < Nd) T2[4 * m * ldb + tid - m] = rv;
    __syncthreads();
    BCR_CUT(1);  // assembly in LDS
    if (top.n > 0 && tid >= 192) {
      // (the factors of the folded back-substitution levels were written on other XCDs: one word of each 128-byte line
      // requested now, by a wave whose share of the factorisation is the smallest, so that they are close when needed)
      double sink = 0.0;
      for (int t = 0; t < top.n; ++t) {
        const long words = (long)top.lv[t].n_elim * mm;
        for (long e = (long)(tid - 192) * 16; e < words; e += 64 * 16)
          sink += B[top.lv[t].Fa + e] + B[top.lv[t].Fb + e] + B[top.lv[t].LiT + e];
      }
      if (sink == 1.2345678e-300) xs[0] = sink;  // (never true; keeps the loads)
    }
    bool bad = bcr_factor_T(T1, m, ldb, tid, bcast);
    const double* F = T1 + (long)m * ldb;
    if (tid < 64) {
      double acc[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) acc[q] = 0.0;
      bcr_gemm24_xyt(F, F, tid, acc);
      const int r0 = 3 * (tid >> 3), c0 = 3 * (tid & 7);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2) P2[(r0 + a) * ldb + c0 + b2] = acc[3 * a + b2];
    } else if (tid < 64 + m) {
      const int r = tid - 64;
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int t = 0; t < 24; t += 2) {
        s0 = fma(F[r * ldb + t], T1[4 * m * ldb + t], s0);
        s1 = fma(F[r * ldb + t + 1], T1[4 * m * ldb + t + 1], s1);
      }
      T2[4 * m * ldb + r] -= s0 + s1;
    }
    __syncthreads();
    if (bcr_factor_T(T2, m, ldb, tid, bcast, P2)) bad = true;
    double* tb = P2;  // (free again)
    if (tid < m) {  // x_1 = L_1^-T y_1
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < 24; k += 2) {
        s0 = fma(T2[(3 * m + tid) * ldb + k], T2[4 * m * ldb + k], s0);
        s1 = fma(T2[(3 * m + tid) * ldb + k + 1], T2[4 * m * ldb + k + 1], s1);
      }
      xs[m + tid] = s0 + s1;
    }
    __syncthreads();
    if (tid < m) {  // t = y_0 - F^T x_1
      double s0 = T1[4 * m * ldb + tid], s1 = 0.0;
#pragma unroll
      for (int r = 0; r < 24; r += 2) {
        s0 = fma(-F[r * ldb + tid], xs[m + r], s0);
        s1 = fma(-F[(r + 1) * ldb + tid], xs[m + r + 1], s1);
      }
      tb[tid] = s0 + s1;
    }
    __syncthreads();
    if (tid < m) {  // x_0 = L_0^-T t
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < 24; k += 2) {
        s0 = fma(T1[(3 * m + tid) * ldb + k], tb[k], s0);
        s1 = fma(T1[(3 * m + tid) * ldb + k + 1], tb[k + 1], s1);
      }
      xs[tid] = s0 + s1;
    }
    __syncthreads();
    if (bad && tid == 0) atomic_add_f64(badv, 1.0);
  } else {
  const bool two_in_regs = n == 2 && (Nd == 48 || Nd == 60);  // the usual tail: two blocks, solved out of registers
  if (two_in_regs) {
    // [D_0 . ; E_0^T + E_1  D_1 | r]: every entry of the lower triangle has one writer, everything is requested in ONE
    // round of loads, and nothing is cleared (bcr_tail_regs never lets an entry above the diagonal reach a result)
    double d0[BCR_EPT], d1[BCR_EPT], e0[BCR_EPT], e1[BCR_EPT];
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      d0[u] = d1[u] = e0[u] = e1[u] = 0.0;
      if (w < mm) {
        const int r = w / m, c = w - r * m;
        d0[u] = B[lv.D + w]; d1[u] = B[lv.D + mm + w];
        e0[u] = B[lv.E + c * m + r];  // E_0^T: block (1, 0)
        e1[u] = B[lv.E + mm + w];     // E_1 = block (1, 0) as it is (the loop closure)
      }
    }
    const double rv = tid < Nd ? B[lv.r + tid] : 0.0;
    if (done) return;
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      if (w < mm) {
        const int r = w / m, c = w - r * m;
        if (c <= r) { T[r * ld + c] = d0[u]; T[(m + r) * ld + m + c] = d1[u]; }
        T[(m + r) * ld + c] = e0[u] + e1[u];
      }
    }
    if (tid < Nd) T[Nd * ld + tid] = rv;
  } else {
  if (done) return;
  for (int e = tid; e < R * ld; e += BCR_THREADS) T[e] = 0.0;
  __syncthreads();
  // D, E, r of the level: eight loads in flight per thread and round (a plain strided loop would wait for every
  // load before issuing the next one: ~1 us a round)
  constexpr int UN = 8;
  for (int base = 0; base < n * mm; base += BCR_THREADS * UN) {
    double v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int e = base + u * BCR_THREADS + tid;
      v[u] = e < n * mm ? B[lv.D + e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int e = base + u * BCR_THREADS + tid;
      if (e < n * mm) {
        const int i = e / mm, w = e - i * mm, r = w / m, c = w - r * m;
        if (c <= r) T[(i * m + r) * ld + i * m + c] = v[u];
      }
    }
  }
  for (int e = tid; e < Nd; e += BCR_THREADS) T[Nd * ld + e] = B[lv.r + e];
  if (n >= 2) {
    // E_i = block (i, i + 1): its transpose is the lower block (i + 1, i); the last one, block (n - 1, 0), is lower as
    // it is (n == 2: it lands on the block that E_0^T fills, hence the barrier and the +=)
    for (int base = 0; base < (n - 1) * mm; base += BCR_THREADS * UN) {
      double v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int e = base + u * BCR_THREADS + tid;
        v[u] = e < (n - 1) * mm ? B[lv.E + e] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int e = base + u * BCR_THREADS + tid;
        if (e < (n - 1) * mm) {
          const int i = e / mm, w = e - i * mm, r = w / m, c = w - r * m;
          T[((i + 1) * m + c) * ld + i * m + r] = v[u];
        }
      }
    }
    double vl[BCR_EPT];
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      vl[u] = w < mm ? B[lv.E + (long)(n - 1) * mm + w] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      if (w < mm) { const int r = w / m, c = w - r * m; T[((n - 1) * m + r) * ld + c] += vl[u]; }
    }
  }
  }
  __syncthreads();
  BCR_CUT(1);  // assembly in LDS
  bool bad = false;
  bool in_regs = true;
  switch (Nd) {  // a row of the dense system per lane of ONE wave (Nd + 1 <= 64 rows): bcr_tail_regs
    case 48: bad = bcr_tail_regs<48>(T, ld, xs, tid, bcr_align16(lds, xs + Nd)); break;
    case 54: bad = bcr_tail_regs<54>(T, ld, xs, tid, bcr_align16(lds, xs + Nd)); break;
    case 60: bad = bcr_tail_regs<60>(T, ld, xs, tid, bcr_align16(lds, xs + Nd)); break;
    default: in_regs = false; break;
  }
  if (!in_regs) {
    for (int j = 0; j < nb; ++j) {
      const int c0 = 6 * j;
      double Li[36];
      if (!chol6_inverse_ld(T + c0 * ld + c0, ld, Li)) bad = true;
      double li_mine = 0.0;  // Li[tid] by a static select: a dynamic index would put Li in scratch memory
  #pragma unroll
      for (int e = 0; e < 36; ++e) li_mine = (e == tid) ? Li[e] : li_mine;
      if (tid < 36) sLi[j * 36 + tid] = li_mine;
      for (int rho = c0 + 6 + tid; rho < R; rho += BCR_THREADS) {
        double* row = T + rho * ld + c0;
        double x[6], o[6];
  #pragma unroll
        for (int t = 0; t < 6; ++t) x[t] = row[t];
  #pragma unroll
        for (int c = 0; c < 6; ++c) {
          double s = 0.0;
  #pragma unroll
          for (int t = 0; t <= c; ++t) s += x[t] * Li[6 * c + t];
          o[c] = s;
        }
  #pragma unroll
        for (int c = 0; c < 6; ++c) row[c] = o[c];
      }
      __syncthreads();
      const int nbc = nb - 1 - j, rows = R - (c0 + 6);
      for (int w = tid; w < rows * nbc; w += BCR_THREADS) {
        const int rr = w / nbc, rho = c0 + 6 + rr, kb = j + 1 + (w - rr * nbc), cb = 6 * kb;
        if (rho < cb) continue;  // lower triangle (the right-hand-side row is below everything)
        bcr_update_item(T + rho * ld + c0, T + cb * ld + c0, T + rho * ld + cb, ld);
      }
      __syncthreads();
    }
    // backward sweep on y (row Nd of T)
    double* y = T + (long)Nd * ld;
    for (int j = nb - 1; j >= 0; --j) {
      const int c0 = 6 * j;
      if (tid < 6) {  // x_j = L_jj^-T y_j:  x[c] = sum_{t >= c} Li[t][c] y[t]
        double x = 0.0;
  #pragma unroll
        for (int t = 0; t < 6; ++t) x += (t >= tid) ? sLi[j * 36 + 6 * t + tid] * y[c0 + t] : 0.0;
        xs[c0 + tid] = x;
      }
      __syncthreads();
      for (int rho = tid; rho < c0; rho += BCR_THREADS) {  // y_rho -= sum_c L[c0 + c][rho] x_j[c]
        double s = 0.0;
  #pragma unroll
        for (int c = 0; c < 6; ++c) s += T[(c0 + c) * ld + rho] * xs[c0 + c];
        y[rho] -= s;
      }
      __syncthreads();
    }
  }
  if (bad && tid == 0) atomic_add_f64(badv, 1.0);
  }  // !two_blocked
  BCR_CUT(2);  // factor + sweep
  double* __restrict__ X = B + Xoff;
  double* xtop = bcr_align16(lds, xs + Nd) + BCR_BCAST;  // [top.nx][m]: x of the blocks of the folded levels
  for (int e = tid; e < Nd; e += BCR_THREADS) {
    const int i = e / m, c = e - i * m;
    const long o = (long)i << level;
    X[o * m + c] = xs[e];
    if (top.n > 0) xtop[(o >> top.xshift) * m + c] = xs[e];
    if (c < 6 * cnt[o]) dc[6L * cam0[o] + c] = xs[e];
  }
  for (int t = 0; t < top.n; ++t) {  // the top back-substitution levels, highest first: a wave per eliminated block
    __syncthreads();                  // x of the level above is in xtop
    for (int q = tid >> 6; q < top.lv[t].n_elim; q += BCR_THREADS / 64)
      (void)bcr_backsolve_wave_m<true>(m, B, top.lv[t], Xoff, top.level[t], q, tid & 63, cam0, cnt, dc, xtop, top.xshift);
  }
  BCR_CUT(3);
  BCR_CUTS_PRINT("k_bcr_tail", level, 0);
}
