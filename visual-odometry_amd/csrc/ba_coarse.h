// Aggregation coarse space for the one-workgroup PCG (k_solve_small), gfx950, fp64.
//
// Block-Jacobi leaves the GLOBAL modes of a trajectory's reduced camera system to CG itself: at BASELINE config 2 the
// PCG iteration counts of a solve's LM iterations are 3, 12, 31, 30, 34, and larger Jacobi blocks or a block-tridiagonal
// preconditioner do not change that (numpy on the same systems: 30 -> 30 / 30 / 25 with blocks of 2 / 3 / 5 cameras),
// because the slow modes are smooth along the trajectory, not couplings of neighbours.  A piecewise-constant coarse
// space does: aggregates of `ag` consecutive free cameras, one 6-vector per aggregate (P = the 0/1 prolongation),
//     M^-1 = blockdiag(S_ff)^-1  +  P (P^T S P)^-1 P^T                    (additive two-level; 30 -> 12 iterations)
// Any SPD matrix serves as (P^T S P)^-1, so the one of TWO LM iterations ago is used and nothing of this is on the
// critical path: k_gather_S of iteration i adds its S block (with its damping) to the coarse sums of buffer i & 1
// (fp64 atomics: 36 or 72 per block); when the host has seen iteration i finish it launches k_coarse_build on a second
// stream, with no event on the main stream, and the solve of iteration i + 2 takes the inverse if it is ready by then
// (a query on the host; a stream wait only if it is not).  The first two LM iterations of a solve run on block-Jacobi
// alone (they need 3 and 12 iterations at config 2).
//
// k_coarse_build, one workgroup: the sums Sc = P^T (S + lam D) P (nc = 6 ceil(Cf / ag) <= COARSE_MAX rows) out of
// `csum` (which it clears), the tall matrix [Sc; I] through the right-looking block Cholesky of ba_bcr.h (which leaves
// L^-T in the identity rows), Sc^-1 = L^-T L^-1 written to `out` [nc][nc], out[nc * nc] = 1 if every pivot was positive
// (k_solve_small ignores the coarse term otherwise).
#pragma once
#include "ba_bcr.h"

constexpr int COARSE_MAX = 64;       // coarse unknowns (k_solve_small spreads a row over 8 of its 512 threads): 10 aggregates
constexpr int COARSE_THREADS = 512;

__global__ __launch_bounds__(COARSE_THREADS) void k_coarse_build(DeviceProblem d, double* __restrict__ csum, double* __restrict__ out) {
  extern __shared__ double lds[];
  const int nc = d.c_nc, ld = nc + 1, R = 2 * nc, tid = threadIdx.x;
  double* T = lds;  // [2 nc][ld]: Sc (lower triangle used) over the identity
  for (int e = tid; e < nc * nc; e += COARSE_THREADS) {
    const int row = e / nc, col = e - row * nc;
    T[row * ld + col] = csum[e];
    csum[e] = 0.0;  // consumed: clean for the gather that takes this buffer next
    T[(nc + row) * ld + col] = row == col ? 1.0 : 0.0;
  }
  __syncthreads();
  bool bad = false;
  const int nb = nc / 6;
  for (int j = 0; j < nb; ++j) {
    const int c0 = 6 * j;
    double Li[36];
    if (!chol6_inverse_ld(T + c0 * ld + c0, ld, Li)) bad = true;
    // panel: the rows of Sc below the diagonal block and the identity rows that are non-zero in these columns
    for (int rho = c0 + 6 + tid; rho < R; rho += COARSE_THREADS) {
      if (rho >= nc && rho - nc >= c0 + 6) continue;
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
    for (int w = tid; w < rows * nbc; w += COARSE_THREADS) {
      const int rr = w / nbc, rho = c0 + 6 + rr, cb = 6 * (j + 1 + (w - rr * nbc));
      if (rho < nc ? rho < cb : rho - nc >= c0 + 6) continue;  // Sc: on or below the diagonal; identity rows still zero
      bcr_update_item(T + rho * ld + c0, T + cb * ld + c0, T + rho * ld + cb, ld);
    }
    __syncthreads();
  }
  // Sc^-1 = L^-T L^-1: entry (r, c) = sum_k LiT[r][k] LiT[c][k], LiT upper triangular (k >= max(r, c)); whole 6-blocks of
  // k so that six products are in flight at a time (the entries below the diagonal of LiT are exact zeros)
  const double* LiT = T + nc * ld;
  for (int e = tid; e < nc * nc; e += COARSE_THREADS) {
    const int r = e / nc, c = e - r * nc;
    if (c > r) continue;
    double s = 0.0;
    for (int k0 = 6 * (r / 6); k0 < nc; k0 += 6) {
#pragma unroll
      for (int u = 0; u < 6; ++u) s += LiT[r * ld + k0 + u] * LiT[c * ld + k0 + u];
    }
    out[r * nc + c] = s;
    out[c * nc + r] = s;
  }
  if (tid == 0) out[nc * nc] = bad ? 0.0 : 1.0;
}
