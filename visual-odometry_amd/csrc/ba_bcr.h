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

constexpr int BCR_THREADS = 256;
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

struct BcrPlan {
  int m, n_levels, N;      // n_levels elimination levels lv[0 .. n_levels); lv[n_levels] holds the tail system (D, E, r only)
  long X;                  // offset of x [N][m], indexed by the level-0 block
  BcrLevel lv[BCR_MAX_LEVELS];
};

// ------------------------------------------------------------------------------------------------
// k_bcr_assemble: the damped S (upper 6x6 blocks) and -rhs into the level-0 arrays, which must be zero.
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

// The block Cholesky of the tall matrix T = [D ; Ea ; Eb^T ; I ; r^T] ((4 m + 1) x m in LDS, row stride ld): see
// k_bcr_factor.  Called by every thread of the workgroup after a barrier that made T visible; returns with a barrier.
__device__ __forceinline__ bool bcr_factor_T(double* T, int m, int ld, int tid) {
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
      const double* a = T + rho * ld + c0;
      double x[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) x[t] = a[t];
      double* out = T + rho * ld + cb;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double* b = T + (cb + c) * ld + c0;
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < 6; ++t) s += x[t] * b[t];
        out[c] -= s;
      }
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
    LiT[e] = c >= r ? T[(3 * m + r) * ld + c] : 0.0;
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
__global__ __launch_bounds__(BCR_THREADS) void k_bcr_factor(double* __restrict__ B, BcrPlan plan, int level, const LmCtrl* ctrl,
                                                            double* __restrict__ badv) {
  extern __shared__ double lds[];
  if (ctrl && ctrl->done) return;
  const BcrLevel lv = plan.lv[level];
  const int m = plan.m, ld = m + 1, mm = m * m;
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
  const bool bad = bcr_factor_T(T, m, ld, tid);
  bcr_store_factors(B, lv, q, T, m, ld, tid);
  if (bad && tid == 0) atomic_add_f64(badv, 1.0);
}

// ------------------------------------------------------------------------------------------------
// k_bcr_update: one workgroup per even block e of level l = block e / 2 of level l + 1 (formulas in the header).
// A thread owns a 2 x 2 tile of the block (10 LDS reads per 12 multiply-adds); D' is symmetric, tiles above its
// diagonal only compute the E' part.
__global__ __launch_bounds__(BCR_THREADS) void k_bcr_update(double* __restrict__ B, BcrPlan plan, int level, const LmCtrl* ctrl) {
  extern __shared__ double lds[];
  if (ctrl && ctrl->done) return;
  const BcrLevel lv = plan.lv[level], nx = plan.lv[level + 1];
  const int m = plan.m, ld = m + 1, mm = m * m;
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
  const int h2 = m / 2;        // m is a multiple of 6: 2 x 2 tiles cover the block
  for (int w = tid; w < h2 * h2; w += BCR_THREADS) {
    const int tr = w / h2, tc = w - tr * h2, r = 2 * tr, c = 2 * tc;
    const bool lower = tc <= tr;
    const double *a1 = FbL + r * ld, *a1b = a1 + ld, *a2 = FaR + r * ld, *a2b = a2 + ld;
    const double *b1 = FbL + c * ld, *b1b = b1 + ld, *b2 = FaR + c * ld, *b2b = b2 + ld, *b3 = FbR + c * ld, *b3b = b3 + ld;
    double d00 = 0, d01 = 0, d10 = 0, d11 = 0, e00 = 0, e01 = 0, e10 = 0, e11 = 0;
    if (lower) {
      for (int t = 0; t < m; ++t) {
        const double x1 = a1[t], x1b = a1b[t], x2 = a2[t], x2b = a2b[t];
        const double p1 = b1[t], p1b = b1b[t], p2 = b2[t], p2b = b2b[t], p3 = b3[t], p3b = b3b[t];
        d00 += x1 * p1 + x2 * p2; d01 += x1 * p1b + x2 * p2b; d10 += x1b * p1 + x2b * p2; d11 += x1b * p1b + x2b * p2b;
        e00 += x2 * p3; e01 += x2 * p3b; e10 += x2b * p3; e11 += x2b * p3b;
      }
    } else {
      for (int t = 0; t < m; ++t) {
        const double x2 = a2[t], x2b = a2b[t], p3 = b3[t], p3b = b3b[t];
        e00 += x2 * p3; e01 += x2 * p3b; e10 += x2b * p3; e11 += x2b * p3b;
      }
    }
    const int o00 = r * m + c;
    if (lower) {
      const double n00 = De[o00] - d00, n01 = De[o00 + 1] - d01, n10 = De[o00 + m] - d10, n11 = De[o00 + m + 1] - d11;
      Dn_s[r * ld + c] = n00; Dn_s[r * ld + c + 1] = n01; Dn_s[(r + 1) * ld + c] = n10; Dn_s[(r + 1) * ld + c + 1] = n11;
    }
    if (copy_E) { e00 = -Ee[o00]; e01 = -Ee[o00 + 1]; e10 = -Ee[o00 + m]; e11 = -Ee[o00 + m + 1]; }
    En[o00] = -e00; En[o00 + 1] = -e01; En[o00 + m] = -e10; En[o00 + m + 1] = -e11;
  }
  for (int r = tid; r < m; r += BCR_THREADS) {
    double s = 0.0;
    for (int t = 0; t < m; ++t) s += FbL[r * ld + t] * yL[t] + FaR[r * ld + t] * yR[t];
    B[nx.r + (long)k * m + r] = B[lv.r + (long)e * m + r] - s;
  }
  __syncthreads();
  for (int w = tid; w < mm; w += BCR_THREADS) {  // lower triangle (and the 2 x 2 tiles on the diagonal) mirrored
    const int r = w / m, c = w - r * m;
    Dn[w] = (c <= r || (c == r + 1 && (r & 1) == 0)) ? Dn_s[r * ld + c] : Dn_s[c * ld + r];
  }
}

// ------------------------------------------------------------------------------------------------
// k_bcr_backsolve: one workgroup per eliminated block i of a level: x_i = L^-T (y - Fa^T x_{i-1} - Fb^T x_{i+1}); x of
// the even neighbours is already in X (indexed by the level-0 block: block i of level l is block i << l of level 0).
// Also scatters x_i to the camera step dc ([Cf][6]; cam0[blk] = first free camera of a level-0 block, cnt[blk] its
// camera count).  The three factor blocks come in with one round of loads.
__global__ __launch_bounds__(BCR_THREADS) void k_bcr_backsolve(double* __restrict__ B, BcrPlan plan, int level,
                                                               const int* __restrict__ cam0, const int* __restrict__ cnt,
                                                               const LmCtrl* ctrl, double* __restrict__ dc) {
  __shared__ double sF[3 * BCR_MAX_M * (BCR_MAX_M + 1)];
  __shared__ double xa[BCR_MAX_M], xb[BCR_MAX_M], tt[BCR_MAX_M], part[4][BCR_MAX_M];
  if (ctrl && ctrl->done) return;
  const BcrLevel lv = plan.lv[level];
  const int m = plan.m, ld = m + 1, mm = m * m, tid = threadIdx.x;
  const int q = blockIdx.x, n = lv.n;
  const int i = 2 * q + 1;
  const long o = (long)i << level;
  double* __restrict__ X = B + plan.X;
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
#pragma unroll
    for (int u = 0; u < BCR_EPT; ++u) {
      const int w = tid + u * BCR_THREADS;
      if (w < mm) {
        const int r = w / m, c = w - r * m;
        sFa[r * ld + c] = v1[u]; sFb[r * ld + c] = v2[u]; sLi[r * ld + c] = v3[u];
      }
    }
    if (tid < m) { xa[tid] = va; xb[tid] = vb; tt[tid] = vy; }
  }
  __syncthreads();
  // t_c = y_c - sum_r Fa[r][c] xa[r] + Fb[r][c] xb[r]: four row groups per column, folded in a fixed order
  const int c = tid & 63, g = tid >> 6;
  if (c < m) {
    double s = 0.0;
    for (int r = g; r < m; r += 4) s += sFa[r * ld + c] * xa[r] + sFb[r * ld + c] * xb[r];
    part[g][c] = s;
  }
  __syncthreads();
  if (tid < m) tt[tid] -= (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
  __syncthreads();
  if (tid < m) {
    double x = 0.0;
    for (int cc = tid; cc < m; ++cc) x += sLi[tid * ld + cc] * tt[cc];
    X[o * m + tid] = x;
    if (tid < 6 * cnt[o]) dc[6L * cam0[o] + tid] = x;
  }
}

// ------------------------------------------------------------------------------------------------
// k_bcr_level: k_bcr_update of level l and k_bcr_factor of level l + 1 in ONE launch (one workgroup per block k of level
// l + 1 = even block e = 2 k of level l).  A block that is even again at level l + 1 only needs its updated D', r' (and
// E' where it is the last block of an odd count) in memory.  A block that is ODD at level l + 1 is eliminated right
// here: besides D', r' it forms the two couplings it needs from the factors of ITS OWN odd neighbours at level l,
//     block (k - 1, k) = -Fa_{e-1} Fb_{e-1}^T,     block (k, k + 1) = -Fa_{e+1} Fb_{e+1}^T   (or E_e where e + 1 does not exist),
// lays them out as the tall matrix of k_bcr_factor in LDS and factors it: no round trip through memory and one launch
// per level instead of two.
__global__ __launch_bounds__(BCR_THREADS) void k_bcr_level(double* __restrict__ B, BcrPlan plan, int level, const LmCtrl* ctrl,
                                                           double* __restrict__ badv) {
  extern __shared__ double lds[];
  if (ctrl && ctrl->done) return;
  const BcrLevel lv = plan.lv[level], nx = plan.lv[level + 1];
  const int m = plan.m, ld = m + 1, mm = m * m;
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
  double* __restrict__ Dn = B + nx.D + (long)k * mm;
  double* __restrict__ En = B + nx.E + (long)k * mm;
  const bool copy_E = qr < 0;                            // n odd, e = n - 1: blocks n - 1 and 0 stay coupled directly
  const bool need_E = !elim && k == nx.n - 1;             // an even block's E' is only read where it is the last block of an
                                                         // odd count (its right neighbour is block 0: the coupling is copied)
  const int h2 = m / 2;
  for (int w = tid; w < h2 * h2; w += BCR_THREADS) {
    const int tr = w / h2, tc = w - tr * h2, r = 2 * tr, c = 2 * tc;
    const bool lower = tc <= tr;
    const double *a1 = FbL + r * ld, *a1b = a1 + ld, *a2 = FaR + r * ld, *a2b = a2 + ld;
    const double *b1 = FbL + c * ld, *b1b = b1 + ld, *b2 = FaR + c * ld, *b2b = b2 + ld;
    if (lower) {
      double d00 = 0, d01 = 0, d10 = 0, d11 = 0;
      for (int t = 0; t < m; ++t) {
        const double x1 = a1[t], x1b = a1b[t], x2 = a2[t], x2b = a2b[t];
        const double p1 = b1[t], p1b = b1b[t], p2 = b2[t], p2b = b2b[t];
        d00 += x1 * p1 + x2 * p2; d01 += x1 * p1b + x2 * p2b; d10 += x1b * p1 + x2b * p2; d11 += x1b * p1b + x2b * p2b;
      }
      const int o00 = r * m + c;
      T[r * ld + c] = De[o00] - d00; T[r * ld + c + 1] = De[o00 + 1] - d01;
      T[(r + 1) * ld + c] = De[o00 + m] - d10; T[(r + 1) * ld + c + 1] = De[o00 + m + 1] - d11;
    }
    if (elim) {
      // Ea' = block (k - 1, k):  [r][c] = -sum_t FaL[r][t] FbL[c][t];   Eb'^T: [r][c] = block (k, k + 1)[c][r] = -sum_t FaR[c][t] FbR[r][t]
      const double *l0 = FaL + r * ld, *l1 = l0 + ld, *g0 = FbR + r * ld, *g1 = g0 + ld;
      double a00 = 0, a01 = 0, a10 = 0, a11 = 0, t00 = 0, t01 = 0, t10 = 0, t11 = 0;
      for (int t = 0; t < m; ++t) {
        const double x0 = l0[t], x1 = l1[t], p0 = b1[t], p1 = b1b[t];
        a00 += x0 * p0; a01 += x0 * p1; a10 += x1 * p0; a11 += x1 * p1;
        const double y0 = g0[t], y1 = g1[t], s0 = b2[t], s1 = b2b[t];
        t00 += y0 * s0; t01 += y0 * s1; t10 += y1 * s0; t11 += y1 * s1;
      }
      double* A = T + (m + r) * ld + c;
      A[0] = -a00; A[1] = -a01; A[ld] = -a10; A[ld + 1] = -a11;
      double* Tb = T + (2 * m + r) * ld + c;
      if (copy_E) { t00 = -Ee[c * m + r]; t01 = -Ee[(c + 1) * m + r]; t10 = -Ee[c * m + r + 1]; t11 = -Ee[(c + 1) * m + r + 1]; }
      Tb[0] = -t00; Tb[1] = -t01; Tb[ld] = -t10; Tb[ld + 1] = -t11;
      double* I = T + (3 * m + r) * ld + c;
      I[0] = r == c ? 1.0 : 0.0; I[1] = 0.0; I[ld] = 0.0; I[ld + 1] = r == c ? 1.0 : 0.0;
    } else if (need_E) {
      const int o00 = r * m + c;
      if (copy_E) {
        En[o00] = Ee[o00]; En[o00 + 1] = Ee[o00 + 1]; En[o00 + m] = Ee[o00 + m]; En[o00 + m + 1] = Ee[o00 + m + 1];
      } else {  // block (e, e + 2) = -FaR FbR^T
        const double *b3 = FbR + c * ld, *b3b = b3 + ld;
        double e00 = 0, e01 = 0, e10 = 0, e11 = 0;
        for (int t = 0; t < m; ++t) {
          const double x2 = a2[t], x2b = a2b[t], p3 = b3[t], p3b = b3b[t];
          e00 += x2 * p3; e01 += x2 * p3b; e10 += x2b * p3; e11 += x2b * p3b;
        }
        En[o00] = -e00; En[o00 + 1] = -e01; En[o00 + m] = -e10; En[o00 + m + 1] = -e11;
      }
    }
  }
  for (int r = tid; r < m; r += BCR_THREADS) {
    double s = 0.0;
    for (int t = 0; t < m; ++t) s += FbL[r * ld + t] * yL[t] + FaR[r * ld + t] * yR[t];
    const double v = B[lv.r + (long)e * m + r] - s;
    if (elim) T[4 * m * ld + r] = v; else B[nx.r + (long)k * m + r] = v;
  }
  __syncthreads();
  if (!elim) {
    for (int w = tid; w < mm; w += BCR_THREADS) {  // lower triangle (and the 2 x 2 tiles on the diagonal) mirrored
      const int r = w / m, c = w - r * m;
      Dn[w] = (c <= r || (c == r + 1 && (r & 1) == 0)) ? T[r * ld + c] : T[c * ld + r];
    }
    return;
  }
  const bool bad = bcr_factor_T(T, m, ld, tid);
  bcr_store_factors(B, nx, k >> 1, T, m, ld, tid);
  if (bad && tid == 0) atomic_add_f64(badv, 1.0);
}

// ------------------------------------------------------------------------------------------------
// k_bcr_tail: the last n <= max(2, BCR_TAIL_ROWS / m) blocks (level `plan.n_levels`), one workgroup: the cyclic block
// tridiagonal system is laid out as ONE dense symmetric matrix of Nd = n m rows in LDS, with the right-hand side as row
// Nd (lower triangle: D_i on the diagonal, E_i^T below it, the loop closure E_{n-1} in the corner; n == 2: both
// couplings of blocks 0 and 1 land on the same block), factored by the block Cholesky loop of k_bcr_factor (the
// right-hand-side row comes out as y = L^-1 r), then solved backwards column block by column block
// (x_j = L_jj^-T y_j;  y_rho -= L[j][rho] x_j for the rows above).  Writes x of the tail blocks to X and to dc.
__global__ __launch_bounds__(BCR_THREADS) void k_bcr_tail(double* __restrict__ B, BcrPlan plan, const int* __restrict__ cam0,
                                                          const int* __restrict__ cnt, const LmCtrl* ctrl,
                                                          double* __restrict__ dc, double* __restrict__ badv) {
  extern __shared__ double lds[];
  if (ctrl && ctrl->done) return;
  const int level = plan.n_levels;
  const BcrLevel lv = plan.lv[level];
  const int m = plan.m, mm = m * m, n = lv.n, Nd = n * m, ld = Nd + 1, nb = Nd / 6, R = Nd + 1;
  const int tid = threadIdx.x;
  double* T = lds;                      // [R][ld]
  double* sLi = T + (long)R * ld;       // [nb][36] inverses of the diagonal blocks of L
  double* xs = sLi + nb * 36;           // [Nd]
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
  __syncthreads();
  bool bad = false;
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
      const double* a = T + rho * ld + c0;
      double x[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) x[t] = a[t];
      double* out = T + rho * ld + cb;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const double* b = T + (cb + c) * ld + c0;
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < 6; ++t) s += x[t] * b[t];
        out[c] -= s;
      }
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
  double* __restrict__ X = B + plan.X;
  for (int e = tid; e < Nd; e += BCR_THREADS) {
    const int i = e / m, c = e - i * m;
    const long o = (long)i << level;
    X[o * m + c] = xs[e];
    if (c < 6 * cnt[o]) dc[6L * cam0[o] + c] = xs[e];
  }
  if (bad && tid == 0) atomic_add_f64(badv, 1.0);
}
