// Dense fp64 Cholesky solve of the reduced camera system  S dc = -rhs  on the whole GPU, for camera
// systems too large for one workgroup and too ill-conditioned for block-Jacobi PCG (a 1000-camera
// ring needs > 1000 PCG iterations per LM step once the damping is small; SURVEY.md section 7
// "dense fp64 Cholesky (MFMA) selectable by camera count").
//
// Layout: A is row-major, ld = nt * 128 columns, (nt + 1) * 128 rows.  Rows [0, nt*128) hold the
// lower triangle of the damped S (identity on the padding diagonal), tile row nt holds b = -rhs in its
// first row: the factorisation then leaves y = L^-1 b there (forward substitution for free), and a
// right-looking backward sweep turns it into x.
//
// Right-looking blocked algorithm, tile 128, three launches per block column k:
//   k_dense_potrf   one workgroup: Cholesky of the 128 x 128 diagonal tile in LDS (packed lower)
//   k_dense_trsm    one workgroup per 64 rows below: X = T L_kk^-T (4 lanes per row, L_kk in LDS)
//   k_dense_update  one workgroup per trailing 128 x 128 tile: C -= P_i P_j^T with
//                   v_mfma_f64_16x16x4_f64 (each wave a 64 x 64 quadrant = 4 x 4 MFMA tiles, panel
//                   chunks of 32 columns staged through LDS)                      <- the flops: (6C)^3 / 3
// then nt launches of k_dense_backsolve.  Bound: fp64 MFMA (78.6 TFLOP/s) for k_dense_update; the
// potrf / trsm pair is a latency chain of nt steps.
#pragma once
#include "ba_kernels.h"

constexpr int DT = 128;               // tile
constexpr int DPACK = DT * (DT + 1) / 2;
constexpr int DCHUNK = 32, DSTRIDE = 36;  // LDS panel chunk: 128 rows x 32 columns, row stride 36 doubles

typedef double dvec4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool dense_skip(const LmCtrl* ctrl) { return ctrl != nullptr && ctrl->done != 0; }

// A <- damped S (lower), padding diagonal, rhs row.  The buffer was zeroed by a memset.
__global__ __launch_bounds__(256) void k_dense_assemble(DeviceProblem d, double* __restrict__ A, long ld, int nt,
                                                        double lam, double min_diag, int solver_mode) {
  if (solver_mode) {
    if (d.ctrl->done) return;
    lam = d.ctrl->lam;
  }
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nS = (long)d.nu * 36, n = 6L * d.Cf, npad = (long)nt * DT;
  if (t < nS) {
    const int s = (int)(t / 36), e = (int)(t - 36L * s), r = e / 6, c = e - 6 * r;
    const int i = d.urow[s], j = d.ucol[s];
    double v = d.S[t];
    if (i == j) {
      if (c > r) return;
      if (r == c) v += lam * fmax(u_diag(d, i, r), min_diag);
      A[(6L * i + r) * ld + 6L * i + c] = v;
    } else {
      A[(6L * j + c) * ld + 6L * i + r] = v;  // S_ij, i < j, transposed into the lower triangle
    }
  } else if (t < nS + npad) {
    const long p = t - nS;
    if (p >= n) A[p * ld + p] = 1.0;
    if (p < n) A[npad * ld + p] = -d.rhs[p];
  }
}

// Cholesky of diagonal tile k, in place (lower).  One workgroup, L packed in LDS.
__global__ __launch_bounds__(256) void k_dense_potrf(double* __restrict__ A, long ld, int k, const LmCtrl* ctrl,
                                                     double* __restrict__ bad_flag) {
  extern __shared__ double Lp[];  // [DPACK]
  if (dense_skip(ctrl)) return;
  const int t = threadIdx.x;
  double* At = A + ((long)k * DT) * ld + (long)k * DT;
  for (int i = 0; i < DT; ++i)
    for (int j = t; j <= i; j += 256) Lp[i * (i + 1) / 2 + j] = At[(long)i * ld + j];
  __syncthreads();
  const int ti = t >> 4, tj = t & 15;
  bool bad = false;
  for (int c = 0; c < DT; ++c) {
    double piv = Lp[c * (c + 1) / 2 + c];
    if (!(piv > 0.0)) { bad = true; piv = 1.0; }
    const double inv = 1.0 / sqrt(piv);
    __syncthreads();
    if (t < DT - c) {
      const int i = c + t;
      Lp[i * (i + 1) / 2 + c] *= inv;  // i == c: sqrt(piv)
    }
    __syncthreads();
    for (int i = c + 1 + ti; i < DT; i += 16) {
      const double lic = Lp[i * (i + 1) / 2 + c];
      for (int j = c + 1 + tj; j <= i; j += 16) Lp[i * (i + 1) / 2 + j] -= lic * Lp[j * (j + 1) / 2 + c];
    }
    __syncthreads();
  }
  for (int i = 0; i < DT; ++i)
    for (int j = t; j <= i; j += 256) At[(long)i * ld + j] = Lp[i * (i + 1) / 2 + j];
  if (bad && t == 0) atomic_add_f64(bad_flag, 1.0);
}

// Rows below the diagonal tile: X = T L_kk^-T.  One workgroup per 64 rows, 4 lanes per row.
__global__ __launch_bounds__(256) void k_dense_trsm(double* __restrict__ A, long ld, int k, const LmCtrl* ctrl) {
  extern __shared__ double sm[];
  double* Lp = sm;             // [DPACK]
  double* invd = Lp + DPACK;   // [DT]
  double* X = invd + DT;       // [64][DT + 1]
  if (dense_skip(ctrl)) return;
  const int t = threadIdx.x;
  const double* At = A + ((long)k * DT) * ld + (long)k * DT;
  for (int i = 0; i < DT; ++i)
    for (int j = t; j <= i; j += 256) Lp[i * (i + 1) / 2 + j] = At[(long)i * ld + j];
  const long row0 = (long)(k + 1) * DT + (long)blockIdx.x * 64;
  double* T = A + row0 * ld + (long)k * DT;
  for (int e = t; e < 64 * DT; e += 256) {
    const int r = e >> 7, c = e & (DT - 1);
    X[r * (DT + 1) + c] = T[(long)r * ld + c];
  }
  __syncthreads();
  if (t < DT) invd[t] = 1.0 / Lp[t * (t + 1) / 2 + t];
  __syncthreads();
  const int row = t >> 2, part = t & 3;
  double* xr = X + row * (DT + 1);
  for (int j = 0; j < DT; ++j) {
    const double* lj = Lp + j * (j + 1) / 2;
    double s = 0.0;
    for (int u = part; u < j; u += 4) s += xr[u] * lj[u];
    s += dpp_move_f64<0xB1>(s);  // the 4 lanes of a row are one quad
    s += dpp_move_f64<0x4E>(s);
    if (part == 0) xr[j] = (xr[j] - s) * invd[j];
    __syncthreads();
  }
  for (int e = t; e < 64 * DT; e += 256) {
    const int r = e >> 7, c = e & (DT - 1);
    T[(long)r * ld + c] = X[r * (DT + 1) + c];
  }
}

// Trailing update after block column k: C(ti, tj) -= P_ti P_tj^T for k < tj <= ti (tile row nt, the
// rhs row, takes part with tj < nt).  256 threads, wave (wr, wc) owns a 64 x 64 quadrant.
__global__ __launch_bounds__(256) void k_dense_update(double* __restrict__ A, long ld, int k, int nt, const LmCtrl* ctrl) {
  __shared__ __align__(16) double sA[DT * DSTRIDE];
  __shared__ __align__(16) double sB[DT * DSTRIDE];
  if (dense_skip(ctrl)) return;
  // block -> (r, c), c <= r, of the trailing lower triangle
  const int bid = blockIdx.x;
  int r = (int)((sqrt(8.0 * bid + 1.0) - 1.0) * 0.5);
  while ((r + 1) * (r + 2) / 2 <= bid) ++r;
  while (r * (r + 1) / 2 > bid) --r;
  const int c = bid - r * (r + 1) / 2;
  const int ti = k + 1 + r, tj = k + 1 + c;
  if (tj >= nt) return;  // the corner below the rhs row is never used
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 1, wc = w & 1;
  const double* Pi = A + ((long)ti * DT) * ld + (long)k * DT;
  const double* Pj = A + ((long)tj * DT) * ld + (long)k * DT;
  dvec4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = dvec4{0.0, 0.0, 0.0, 0.0};
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int kc = 0; kc < DT; kc += DCHUNK) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = t + 256 * it, row = idx >> 4, c2 = (idx & 15) * 2;
      const double2 va = *reinterpret_cast<const double2*>(Pi + (long)row * ld + kc + c2);
      const double2 vb = *reinterpret_cast<const double2*>(Pj + (long)row * ld + kc + c2);
      *reinterpret_cast<double2*>(&sA[row * DSTRIDE + c2]) = va;
      *reinterpret_cast<double2*>(&sB[row * DSTRIDE + c2]) = vb;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < DCHUNK; ks += 8) {
      double2 fa[4], fb[4];
      // lane supplies k = ks + 2 * (lane >> 4) + {0, 1}: two MFMA k-groups (the order of the k sum is free
      // as long as A and B use the same assignment)
#pragma unroll
      for (int a = 0; a < 4; ++a)
        fa[a] = *reinterpret_cast<const double2*>(&sA[(wr * 64 + 16 * a + l15) * DSTRIDE + ks + 2 * l4]);
#pragma unroll
      for (int b = 0; b < 4; ++b)
        fb[b] = *reinterpret_cast<const double2*>(&sB[(wc * 64 + 16 * b + l15) * DSTRIDE + ks + 2 * l4]);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
        }
    }
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  double* Ct = A + ((long)ti * DT + wr * 64) * ld + (long)tj * DT + wc * 64;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double* p = Ct + (long)(16 * a + l4 + 4 * q) * ld + 16 * b + l15;
        *p -= acc[a][b][q];
      }
}

// Backward sweep, step k (descending): x_k = L_kk^-T y_k (every workgroup, redundantly), then
// workgroup j < k updates y_j -= L_kj^T x_k; workgroup j == k stores x_k.  y / x live in the rhs row.
__global__ __launch_bounds__(256) void k_dense_backsolve(double* __restrict__ A, long ld, int k, int nt, const LmCtrl* ctrl,
                                                         double* __restrict__ dc, int n) {
  extern __shared__ double sm[];
  double* Lp = sm;           // [DPACK]
  double* y = Lp + DPACK;    // [DT]
  double* part = y + DT;     // [DT]
  if (dense_skip(ctrl)) return;
  const int t = threadIdx.x, j = blockIdx.x;
  const double* At = A + ((long)k * DT) * ld + (long)k * DT;
  double* yrow = A + ((long)nt * DT) * ld;
  for (int i = 0; i < DT; ++i)
    for (int c = t; c <= i; c += 256) Lp[i * (i + 1) / 2 + c] = At[(long)i * ld + c];
  if (t < DT) y[t] = yrow[(long)k * DT + t];
  __syncthreads();
  for (int m = DT - 1; m >= 0; --m) {
    const double xm = y[m] / Lp[m * (m + 1) / 2 + m];
    __syncthreads();
    if (t == 0) y[m] = xm;
    if (t < m) y[t] -= Lp[m * (m + 1) / 2 + t] * xm;
    __syncthreads();
  }
  if (j == k) {
    if (t < DT) {
      const long g = (long)k * DT + t;
      yrow[g] = y[t];
      if (g < n) dc[g] = y[t];
    }
    return;
  }
  // y_j[c] -= sum_r L[k*128 + r][j*128 + c] x_k[r]; threads (c, half of r)
  const int c = t & (DT - 1), half = t >> 7;
  const double* Lkj = A + ((long)k * DT + 64 * half) * ld + (long)j * DT + c;
  double s = 0.0;
#pragma unroll 8
  for (int r2 = 0; r2 < 64; ++r2) s += Lkj[(long)r2 * ld] * y[64 * half + r2];
  if (half) part[c] = s;
  __syncthreads();
  if (!half) yrow[(long)j * DT + c] -= s + part[c];
}
