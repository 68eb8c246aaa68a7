// k_solve_components: dense Cholesky solve of the reduced camera system, one wave per connected component of the
// camera graph (cameras that share no landmark, directly or through a chain of cameras, belong to different
// components: S is block diagonal over them).  This is the linear solver of batched local bundle adjustment (BASELINE
// config 5: thousands of independent 10-keyframe windows solved as ONE problem) and of any problem whose components
// have <= COMP_MAX_CAMS free cameras.  Exact step: no PCG iterations, no preconditioner.
//
// The component's system lives in LDS as the tall matrix T = [A ; b^T] ((n + 1) x n, n = 6 x cameras <= 72, lower
// triangle of A) and goes through the right-looking BLOCK Cholesky that ba_bcr.h uses (6 columns per step: every lane
// factors the 6 x 6 diagonal block itself, a lane scales a row of the panel, the trailing update is spread over the
// (row, column block) items of the lower triangle), which leaves y^T = (L^-1 b)^T in the last row; a block backward
// sweep then gives x = L^-T y.  9 block steps for a 10-keyframe window where the first version took 54 column steps of
// three barriers each plus 108 substitution steps; the assembly reads the component's block list in one round (a lane
// per block for the slot / camera numbers, then the 36 entries of eight blocks in flight at a time) where it walked the
// list block by block, three dependent loads each.  Config 5 (1024 windows): 140 -> 43 us per launch.
// Bound: LDS latency of ~3 nb dependent phases; reads S once (288 B per block).
#pragma once
#include "ba_bcr.h"

constexpr int COMP_MAX_BLOCKS = COMP_MAX_CAMS * (COMP_MAX_CAMS + 1) / 2;

__host__ __device__ constexpr size_t comp_lds_bytes(int n) {
  return ((size_t)(n + 1) * (n + 1) + (size_t)(n / 6) * 36 + n) * sizeof(double) + (size_t)4 * COMP_MAX_BLOCKS * sizeof(int);
}

__global__ __launch_bounds__(64) void k_solve_components(DeviceProblem d, double lam, double min_diag) {
  extern __shared__ double sm[];
  const bool per_comp = lam < 0.0 && d.clm.lam != nullptr;
  if (lam < 0.0) {
    if (d.ctrl->done) return;
    lam = d.ctrl->lam;
  }
  const int comp = blockIdx.x, lane = threadIdx.x;
  const int cam0 = d.comp_ptr[comp], nc = d.comp_ptr[comp + 1] - cam0, n = 6 * nc, ld = n + 1, nb = nc, R = n + 1;
  if (per_comp) {  // a damping per component; a frozen component takes no step
    lam = d.clm.lam[comp];
    if (d.clm.done[comp]) {
      for (int t = lane; t < n; t += 64) d.dc[6L * d.comp_cams[cam0 + t / 6] + t % 6] = 0.0;
      if (lane == 0 && comp == 0) { d.scal[SC_PCGIT] = 0.0; d.scal[SC_DONE] = 1.0; }
      return;
    }
  }
  double* T = sm;                   // [n + 1][ld]: A (lower triangle used), then b^T
  double* sLi = T + (size_t)R * ld; // [nb][36] inverses of the diagonal blocks of L
  double* xs = sLi + nb * 36;       // [n]
  int* meta = reinterpret_cast<int*>(xs + n);  // [4][COMP_MAX_BLOCKS]: slot, local row, local column, row camera
  const int e0 = d.comp_slot_ptr[comp], ne = d.comp_slot_ptr[comp + 1] - e0;
  for (int q = lane; q < ne; q += 64) {
    const int s = d.comp_slots[e0 + q], i = d.urow[s], j = d.ucol[s];
    meta[q] = s;
    meta[COMP_MAX_BLOCKS + q] = d.comp_local[i];
    meta[2 * COMP_MAX_BLOCKS + q] = d.comp_local[j];
    meta[3 * COMP_MAX_BLOCKS + q] = d.free2cam[i];
  }
  for (int t = lane; t < R * ld; t += 64) T[t] = 0.0;
  __syncthreads();
  // blocks of the component: slot (i, j), i <= j (global free-camera numbers) -> lower entry T[6 lj + c][6 li + r]
  constexpr int UN = 8;
  for (int base = 0; base < ne * 36; base += 64 * UN) {
    double v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int t = base + u * 64 + lane;
      v[u] = 0.0;
      if (t < ne * 36) {
        const int q = t / 36, e = t - 36 * q, r = e / 6, c = e - 6 * r;
        v[u] = d.S[(long)meta[q] * 36 + e];
        if (meta[COMP_MAX_BLOCKS + q] == meta[2 * COMP_MAX_BLOCKS + q] && r == c)
          v[u] += lam * fmax(d.Ugc[(long)meta[3 * COMP_MAX_BLOCKS + q] * 27 + sym6_index(r, r)], min_diag);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int t = base + u * 64 + lane;
      if (t < ne * 36) {
        const int q = t / 36, e = t - 36 * q, r = e / 6, c = e - 6 * r;
        const int li = meta[COMP_MAX_BLOCKS + q], lj = meta[2 * COMP_MAX_BLOCKS + q];
        if (li == lj) {
          if (c <= r) T[(6 * li + r) * ld + 6 * li + c] = v[u];
        } else {
          T[(6 * lj + c) * ld + 6 * li + r] = v[u];  // lj > li: the transposed block is the lower one
        }
      }
    }
  }
  for (int t = lane; t < n; t += 64) {
    const int f = d.comp_cams[cam0 + t / 6];
    T[n * ld + t] = -d.rhs[6L * f + t % 6];
  }
  __syncthreads();
  // block Cholesky of T: columns c0 .. c0 + 5 per step
  bool bad = false;
  for (int j = 0; j < nb; ++j) {
    const int c0 = 6 * j;
    double Li[36];
    if (!chol6_inverse_ld(T + c0 * ld + c0, ld, Li)) bad = true;
    double li_mine = 0.0;  // Li[lane] by a static select: a dynamic index would put Li in scratch memory
#pragma unroll
    for (int e = 0; e < 36; ++e) li_mine = (e == lane) ? Li[e] : li_mine;
    if (lane < 36) sLi[j * 36 + lane] = li_mine;
    for (int rho = c0 + 6 + lane; rho < R; rho += 64) {  // panel: row <- row L_jj^-T
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
    // trailing update over the lower triangle: block pairs (rb >= kb) of the m = nb - 1 - j remaining block rows, six
    // rows each, then the right-hand-side row against every column block
    const int m = nb - 1 - j, npair = m * (m + 1) / 2, nitem = 6 * npair + m;
    for (int w = lane; w < nitem; w += 64) {
      int rho, kb;
      if (w < 6 * npair) {
        const int q = w / 6;
        int rb = 0;
        while ((rb + 1) * (rb + 2) / 2 <= q) ++rb;
        kb = q - rb * (rb + 1) / 2;
        rho = c0 + 6 + 6 * rb + (w - 6 * q);
      } else {
        kb = w - 6 * npair;
        rho = n;
      }
      const int cb = c0 + 6 + 6 * kb;
      bcr_update_item(T + rho * ld + c0, T + cb * ld + c0, T + rho * ld + cb, ld);
    }
    __syncthreads();
  }
  // backward sweep on y (row n of T): x_j = L_jj^-T y_j, then y_rho -= sum_c L[c0 + c][rho] x_j[c] for the rows above
  double* y = T + (size_t)n * ld;
  for (int j = nb - 1; j >= 0; --j) {
    const int c0 = 6 * j;
    if (lane < 6) {
      double x = 0.0;
#pragma unroll
      for (int t = 0; t < 6; ++t) x += (t >= lane) ? sLi[j * 36 + 6 * t + lane] * y[c0 + t] : 0.0;
      xs[c0 + lane] = x;
    }
    __syncthreads();
    for (int rho = lane; rho < c0; rho += 64) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 6; ++c) s += T[(c0 + c) * ld + rho] * xs[c0 + c];
      y[rho] -= s;
    }
    __syncthreads();
  }
  for (int t = lane; t < n; t += 64) {
    const int f = d.comp_cams[cam0 + t / 6];
    d.dc[6L * f + t % 6] = xs[t];
  }
  if (lane == 0) {
    if (bad) {
      if (per_comp) d.clm.stat[CL_STRIDE * comp + 4] = 1.0;  // only this component's step is refused
      else atomic_add_f64(&SCAL_MINE(d)[SC_BADV], 1.0);
    }
    if (comp == 0) { d.scal[SC_PCGIT] = 0.0; d.scal[SC_DONE] = 1.0; }
  }
}
