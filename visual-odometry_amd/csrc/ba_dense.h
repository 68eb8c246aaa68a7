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
//   k_dense_potrf   one workgroup: Cholesky of the 128 x 128 diagonal tile, rows in registers
//   k_dense_trsm    one workgroup per 128 rows below: X = T L_kk^-T (2 lanes per row, rows in registers)
//   k_dense_update  one workgroup per trailing 128 x 128 tile: C -= P_i P_j^T with
//                   v_mfma_f64_16x16x4_f64 (8 waves of 64 x 32 outputs = 4 x 2 MFMA tiles each, panel
//                   chunks of 32 columns software-pipelined through LDS)          <- the flops: (6C)^3 / 3
// then nt launches of k_dense_backsolve.  Bound: fp64 MFMA issue for k_dense_update (measured on MI355X:
// one v_mfma_f64_16x16x4_f64 per 58.6 ns per wave, 33-44 TFLOP/s chip-wide, tools/mfma_f64_bench.hip);
// the potrf / trsm pair is a latency chain of nt steps.
#pragma once
#include "ba_kernels.h"

constexpr int DT = 128;               // tile
constexpr int DPACK = DT * (DT + 1) / 2;
constexpr int DCHUNK = 32, DSTRIDE = 36;  // LDS panel chunk: 128 rows x 32 columns, row stride 36 doubles

typedef double dvec4 __attribute__((ext_vector_type(4)));

// Lower triangle of a 128 x 128 tile <-> packed LDS copy, as flat batched loops (a row-by-row loop
// waits for one global round trip per row).
__device__ __forceinline__ void tile_to_packed(const double* __restrict__ At, long ld, double* Lp) {
  const int t = threadIdx.x;
#pragma unroll 1
  for (int e0 = 0; e0 < DT * DT; e0 += 256 * 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u + t, i = e >> 7, j = e & (DT - 1);
      v[u] = (j <= i) ? At[(long)i * ld + j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u + t, i = e >> 7, j = e & (DT - 1);
      if (j <= i) Lp[i * (i + 1) / 2 + j] = v[u];
    }
  }
}
__device__ __forceinline__ void packed_to_tile(const double* Lp, double* __restrict__ At, long ld) {
  const int t = threadIdx.x;
  for (int e = t; e < DT * DT; e += 256) {
    const int i = e >> 7, j = e & (DT - 1);
    if (j <= i) At[(long)i * ld + j] = Lp[i * (i + 1) / 2 + j];
  }
}

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

// Cholesky of diagonal tile k, in place (lower).  One workgroup; thread (i, h) keeps columns
// [64 h, 64 h + 64) of row i in REGISTERS (the column loop is fully unrolled, so every index is
// static): per column one broadcast of the scaled column through LDS and 2 barriers, no LDS
// read-modify-write chains (those cost ~200 cycles per element and made this kernel 10x slower).
template <int H>
__device__ __forceinline__ void potrf_half(double* __restrict__ Arow, int i, double* rc, double* pivinv, bool& bad) {
  double row[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) row[j] = Arow[j];
#pragma unroll
  for (int c = 0; c < DT; ++c) {
    const int hc = c >> 6, cl = c & 63;
    if (H == hc && i == c) {
      double piv = row[cl];
      if (!(piv > 0.0)) { bad = true; piv = 1.0; }
      const double dd = sqrt(piv);
      row[cl] = dd;
      *pivinv = 1.0 / dd;
    }
    __syncthreads();
    if (H == hc && i > c) {
      row[cl] *= *pivinv;
      rc[i] = row[cl];
    }
    __syncthreads();
    if ((H == 1 || c < 63) && i > c) {
      const double lic = rc[i];
#pragma unroll
      for (int jl = 0; jl < 64; ++jl)
        if (64 * H + jl > c) row[jl] -= lic * rc[64 * H + jl];  // compile-time test: H and c are constants here
    }
  }
#pragma unroll
  for (int j = 0; j < 64; ++j)
    if (64 * H + j <= i) Arow[j] = row[j];
}

__global__ __launch_bounds__(256) void k_dense_potrf(double* __restrict__ A, long ld, int k, const LmCtrl* ctrl,
                                                     double* __restrict__ bad_flag) {
  __shared__ double rc[DT];  // column c of L (scaled), rows > c
  __shared__ double pivinv;
  if (dense_skip(ctrl)) return;
  const int t = threadIdx.x, i = t & (DT - 1), h = t >> 7;
  double* Arow = A + ((long)k * DT + i) * ld + (long)k * DT + 64 * h;
  bool bad = false;
  // the two halves run different straight-line code with the same number of barriers (waves 0-1 / 2-3)
  if (h == 0) potrf_half<0>(Arow, i, rc, &pivinv, bad);
  else potrf_half<1>(Arow, i, rc, &pivinv, bad);
  if (bad) atomic_add_f64(bad_flag, 1.0);
}

// Rows below the diagonal tile: X = T L_kk^-T.  One workgroup per 128 rows; the two lanes (row, p) of a
// pair keep the entries of parity p of the row in registers, L_kk sits in LDS de-interleaved by column
// parity; x_j = (t_j - sum_{u<j} x_u L[j][u]) / L[j][j] with the sum split over the pair and combined by
// one DPP move.  The j loop is fully unrolled: static register indices, no barrier inside.
__global__ __launch_bounds__(256) void k_dense_trsm(double* __restrict__ A, long ld, int k, const LmCtrl* ctrl) {
  extern __shared__ double sm[];
  double* Lde = sm;                 // [DT][2][64]: L[j][2 u2 + p] at (j * 2 + p) * 64 + u2
  double* invd = Lde + DT * DT;     // [DT]
  if (dense_skip(ctrl)) return;
  const int t = threadIdx.x;
  const double* At = A + ((long)k * DT) * ld + (long)k * DT;
#pragma unroll 1
  for (int e0 = 0; e0 < DT * DT; e0 += 256 * 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u + t, j = e >> 7, c = e & (DT - 1);
      v[u] = (c <= j) ? At[(long)j * ld + c] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u + t, j = e >> 7, c = e & (DT - 1);
      Lde[(j * 2 + (c & 1)) * 64 + (c >> 1)] = v[u];
      if (c == j) invd[j] = 1.0 / v[u];
    }
  }
  const int row = t >> 1, p = t & 1;
  double* Trow = A + ((long)(k + 1) * DT + (long)blockIdx.x * DT + row) * ld + (long)k * DT;
  double x[64];
#pragma unroll
  for (int u2 = 0; u2 < 64; ++u2) x[u2] = Trow[2 * u2 + p];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < DT; ++j) {
    const double* lj = Lde + (j * 2 + p) * 64;
    double s0 = 0.0, s1 = 0.0;
    // own-parity u = 2 u2 + p < j
#pragma unroll
    for (int u2 = 0; u2 < 64; ++u2) {
      if (2 * u2 + 1 < j) {  // both parities are below j: no lane test needed
        if (u2 & 1) s1 += x[u2] * lj[u2]; else s0 += x[u2] * lj[u2];
      } else if (2 * u2 < j) {  // only the even entry is below j
        if (p == 0) s0 += x[u2] * lj[u2];
      }
    }
    double s = s0 + s1;
    s += dpp_move_f64<0xB1>(s);  // partner lane of the pair
    if (p == (j & 1)) x[j >> 1] = (x[j >> 1] - s) * invd[j];
  }
#pragma unroll
  for (int u2 = 0; u2 < 64; ++u2) Trow[2 * u2 + p] = x[u2];
}

// Trailing update after block column k: C(ti, tj) -= P_ti P_tj^T for k < tj <= ti (tile row nt, the
// rhs row, takes part with tj < nt).  512 threads = 8 waves of 64 x 32 outputs each.
__global__ __launch_bounds__(512, 2) void k_dense_update(double* __restrict__ A, long ld, int k, int nt, const LmCtrl* ctrl) {
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
  // 8 waves: wave (wr, wc) owns rows [64 wr, +64) x columns [32 wc, +32) = 4 x 2 MFMA tiles (64 accumulator
  // registers: everything stays in VGPRs, two waves per SIMD)
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wr = w >> 2, wc = w & 3;
  const double* Pi = A + ((long)ti * DT) * ld + (long)k * DT;
  const double* Pj = A + ((long)tj * DT) * ld + (long)k * DT;
  dvec4 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = dvec4{0.0, 0.0, 0.0, 0.0};
  const int l15 = lane & 15, l4 = lane >> 4;
  // software pipeline: the loads of chunk kc + 1 are in flight while chunk kc is multiplied
  double2 va[4], vb[4];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = t + 512 * it, row = idx >> 4, c2 = (idx & 15) * 2;
      va[it] = *reinterpret_cast<const double2*>(Pi + (long)row * ld + kc + c2);
      vb[it] = *reinterpret_cast<const double2*>(Pj + (long)row * ld + kc + c2);
    }
  };
  fetch(0);
  for (int kc = 0; kc < DT; kc += DCHUNK) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = t + 512 * it, row = idx >> 4, c2 = (idx & 15) * 2;
      *reinterpret_cast<double2*>(&sA[row * DSTRIDE + c2]) = va[it];
      *reinterpret_cast<double2*>(&sB[row * DSTRIDE + c2]) = vb[it];
    }
    __syncthreads();
    if (kc + DCHUNK < DT) fetch(kc + DCHUNK);
#pragma unroll
    for (int ks = 0; ks < DCHUNK; ks += 8) {
      double2 fa[4], fb[2];
      // lane supplies k = ks + 2 * (lane >> 4) + {0, 1}: two MFMA k-groups (the order of the k sum is free
      // as long as A and B use the same assignment)
#pragma unroll
      for (int a = 0; a < 4; ++a)
        fa[a] = *reinterpret_cast<const double2*>(&sA[(wr * 64 + 16 * a + l15) * DSTRIDE + ks + 2 * l4]);
#pragma unroll
      for (int b = 0; b < 2; ++b)
        fb[b] = *reinterpret_cast<const double2*>(&sB[(wc * 32 + 16 * b + l15) * DSTRIDE + ks + 2 * l4]);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
    }
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  double* Ct = A + ((long)ti * DT + wr * 64) * ld + (long)tj * DT + wc * 32;
#pragma unroll
  for (int a = 0; a < 4; ++a) {  // 8 loads in flight, then 8 stores
    double cv[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) cv[b][q] = Ct[(long)(16 * a + l4 + 4 * q) * ld + 16 * b + l15];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) Ct[(long)(16 * a + l4 + 4 * q) * ld + 16 * b + l15] = cv[b][q] - acc[a][b][q];
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
  tile_to_packed(At, ld, Lp);
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
