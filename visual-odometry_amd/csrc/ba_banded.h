// Banded + arrow block Cholesky of the reduced camera system (gfx950, fp64).
//
// Cameras along a trajectory see a landmark for a few consecutive frames, so S = U - sum W Vd^-1 W^T is
// block-banded in the camera order (half-bandwidth bw blocks); a closed loop (the ring scenes of BASELINE
// configs[2..3]) adds coupling between the first and the last cameras, which an "arrow" of the last na block
// rows / columns absorbs.  The factorisation then costs O(C (bw + na)^2) block products instead of the dense
// O(C^3), but it is one sequential chain of C block columns: ONE workgroup walks it, with the active window
// (the bw + 1 band columns that column j still updates, the na corner columns, all with their arrow rows and
// the right-hand side as one more row) resident in LDS, three barriers per column.
//
// Storage (global, B): block column k of the lower triangle is nrow = bw + 1 + na + 1 slots of 36 doubles:
//   band columns   k <  Cb = Cf - na:  slot i - k for band rows k <= i <= k + bw (i < Cb);
//   all columns:                       slot bw + 1 + (i - Cb) for arrow rows i >= Cb (corner columns use only
//                                      these, their diagonal block included);
//   last slot:                         the right-hand side as row 0 of a block (the augmented matrix
//                                      [S b; b^T .] factorises to [L 0; y^T .] with L y = b: forward
//                                      substitution for free).
// k_banded_factor leaves L in B, with inv(L_jj) in the diagonal slots; k_banded_backsolve (one wave) sweeps
// L^T x = y backwards.  Both are launched once per linear solve.
#pragma once
#include "ba_kernels.h"

struct BandedShape {
  int Cf, Cb, bw, na, nrow;
};
constexpr int BND_THREADS = 256;
constexpr long BND_MAX_LDS_BLOCKS = 400;  // (bw + 1 + na) * nrow blocks of 288 B in LDS (115 KB)
constexpr int BND_MAX_NROW = 28;          // slots per column (4 * BND_THREADS doubles hold one column)

__host__ __device__ inline int bnd_diag_slot(const BandedShape& s, int j) { return j < s.Cb ? 0 : s.bw + 1 + (j - s.Cb); }
// slot of block (i, k), i >= k, in column k; i == Cf is the right-hand-side row
__host__ __device__ inline int bnd_slot(const BandedShape& s, int i, int k) {
  if (i >= s.Cf) return s.nrow - 1;
  return i < s.Cb ? i - k : s.bw + 1 + (i - s.Cb);
}
// camera (row index) of slot sl of column j, or -1 if the slot holds nothing below the diagonal of column j
__host__ __device__ inline int bnd_row_of(const BandedShape& s, int j, int sl) {
  if (sl == s.nrow - 1) return s.Cf;
  if (sl > s.bw) {
    const int i = s.Cb + (sl - s.bw - 1);
    return i > j ? i : -1;
  }
  if (j >= s.Cb || sl == 0) return -1;
  const int i = j + sl;
  return i < s.Cb ? i : -1;
}

// B must be zero.  Scatters the damped S (upper blocks, transposed into the lower triangle) and -rhs.
__global__ __launch_bounds__(256) void k_banded_assemble(DeviceProblem d, double* __restrict__ B, BandedShape sh, double lam,
                                                         double min_diag, int solver_mode) {
  if (solver_mode) {
    if (d.ctrl->done) return;
    lam = d.ctrl->lam;
  }
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nS = (long)d.nu * 36, cs = (long)sh.nrow * 36;
  if (t < nS) {
    const int s = (int)(t / 36), e = (int)(t - 36L * s), r = e / 6, c = e - 6 * r;
    const int i = d.urow[s], j = d.ucol[s];  // i <= j
    double v = d.S[t];
    if (i == j) {
      if (c > r) return;
      if (r == c) v += lam * fmax(u_diag(d, i, r), min_diag);
      double* blk = B + i * cs + bnd_diag_slot(sh, i) * 36;
      blk[6 * r + c] = v;
      blk[6 * c + r] = v;
    } else {
      B[i * cs + bnd_slot(sh, j, i) * 36 + 6 * c + r] = v;  // block (j, i) = S_ij^T
    }
  } else if (t < nS + 6L * d.Cf) {
    const long p = t - nS;
    const int k = (int)(p / 6), e = (int)(p - 6L * k);
    B[k * cs + (long)(sh.nrow - 1) * 36 + e] = -d.rhs[p];
  }
}

// L^-1 (lower, row-major) of the Cholesky factor of the SPD 6x6 block A (lower part used).
__device__ __forceinline__ bool chol6_inverse(const double* A, double Li[36]) {
  double Lm[36];
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 36; ++i) { Lm[i] = 0.0; Li[i] = 0.0; }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s = A[7 * j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= Lm[6 * j + k] * Lm[6 * j + k];
    if (!(s > 0.0)) { ok = false; s = 1.0; }
    // 1 / sqrt(s): hardware estimate + two Newton steps (this chain is on the critical path of every column)
    double id = __builtin_amdgcn_rsq(s);
    id = id * fma(-0.5 * s * id, id, 1.5);
    id = id * fma(-0.5 * s * id, id, 1.5);
    const double dd = s * id;
    Lm[7 * j] = dd;
    Li[7 * j] = id;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double t = A[6 * i + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= Lm[6 * i + k] * Lm[6 * j + k];
      Lm[6 * i + j] = t * id;
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double t = 0.0;
#pragma unroll
      for (int k = j; k < i; ++k) t -= Lm[6 * i + k] * Li[6 * k + j];
      Li[6 * i + j] = t * Li[7 * i];
    }
  return ok;
}

__global__ __launch_bounds__(BND_THREADS) void k_banded_factor(double* __restrict__ B, BandedShape sh, const LmCtrl* ctrl,
                                                               double* __restrict__ badv) {
  extern __shared__ double lds[];
  if (ctrl && ctrl->done) return;
  const int tid = threadIdx.x;
  const int cs = sh.nrow * 36, nwin = sh.bw + 1;
  double* win = lds;                                // [nwin + na][cs]: band ring | corner columns
  auto col_lds = [&](int k) { return win + (long)(k < sh.Cb ? k % nwin : nwin + (k - sh.Cb)) * cs; };
  // initial window: band columns 0 .. bw, all corner columns
  for (int k = 0; k < min(nwin, sh.Cb); ++k)
    for (int e = tid; e < cs; e += BND_THREADS) col_lds(k)[e] = B[(long)k * cs + e];
  for (int k = sh.Cb; k < sh.Cf; ++k)
    for (int e = tid; e < cs; e += BND_THREADS) col_lds(k)[e] = B[(long)k * cs + e];
  // (row slot, column slot) pairs of a band column's update, sa >= sb, sb a camera slot
  unsigned char* pa = (unsigned char*)(win + (long)(nwin + sh.na) * cs);
  unsigned char* pb = pa + 512;
  __shared__ int npairs_s;
  if (tid == 0) {
    int np = 0;
    for (int sa = 1; sa < sh.nrow; ++sa)
      for (int sb = 1; sb <= min(sa, sh.nrow - 2); ++sb) { pa[np] = (unsigned char)sa; pb[np] = (unsigned char)sb; ++np; }
    npairs_s = np;
  }
  __syncthreads();
  const int npairs = npairs_s;
  const int pgrp = tid / 36, e0 = tid - 36 * pgrp, rr0 = e0 / 6, cc0 = e0 - 6 * rr0;
  int jm = 0;  // j mod nwin
  bool bad = false;
  for (int j = 0; j < sh.Cf; ++j) {
    double* cj = col_lds(j);
    const int ds = bnd_diag_slot(sh, j);
    // the band column that enters the window when column j leaves it: requested now, parked after the update
    const int knew = j + nwin;
    const bool have_new = j < sh.Cb && knew < sh.Cb;
    double fresh[4];  // a column is at most BND_MAX_NROW * 36 = 1008 doubles: 4 per thread
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + q * BND_THREADS;
      fresh[q] = (have_new && e < cs) ? B[(long)knew * cs + e] : 0.0;
    }
    // --- L_jj and the rows below it: every thread factors the diagonal block itself (no exchange), then
    //     takes one row of one block: L_ij = S_ij L_jj^-T
    double Li[36];
    if (!chol6_inverse(cj + ds * 36, Li)) bad = true;
    const int sl = tid / 6, r = tid - 6 * sl;
    double out[6];
    bool mine = false;
    if (sl < sh.nrow && sl != ds) {
      const int i = bnd_row_of(sh, j, sl);
      if (i >= 0 && (i < sh.Cf || r == 0)) {
        mine = true;
        const double* src = cj + sl * 36 + 6 * r;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double t = 0.0;
#pragma unroll
          for (int m = 0; m <= c; ++m) t += src[m] * Li[6 * c + m];
          out[c] = t;
        }
      }
    }
    // (a thread reads and writes its own row only, and the diagonal block is not overwritten: no barrier here)
    if (mine) {
#pragma unroll
      for (int c = 0; c < 6; ++c) cj[sl * 36 + 6 * r + c] = out[c];
    }
    double li_mine = 0.0;  // Li[tid] by a static select: a dynamic index would put Li in scratch memory
#pragma unroll
    for (int e = 0; e < 36; ++e) li_mine = (e == tid) ? Li[e] : li_mine;
    __syncthreads();
    // --- trailing update: block (i, k) -= L_ij L_kj^T for rows i >= k below the diagonal of column j
    //     (k a camera, i a camera or the right-hand-side row).  A thread owns one entry (rr, cc) of a block and
    //     walks the (row slot, column slot) pairs, 7 at a time across the workgroup; for band columns the pair
    //     list does not depend on j (slots are offsets from j, arrow and rhs slots are absolute).
    if (j < sh.Cb) {
      for (int p = pgrp < BND_THREADS / 36 ? pgrp : npairs; p < npairs; p += BND_THREADS / 36) {  // 252 of 256 threads
        const int sa = pa[p], sb = pb[p];
        if ((sa <= sh.bw && j + sa >= sh.Cb) || (sb <= sh.bw && j + sb >= sh.Cb)) continue;  // beyond the band's end
        if (sa == sh.nrow - 1 && rr0 != 0) continue;                                          // rhs: one row
        double* tcol;
        if (sb <= sh.bw) {
          int idx = jm + sb;
          if (idx >= nwin) idx -= nwin;
          tcol = win + (long)idx * cs;
        } else {
          tcol = win + (long)(nwin + sb - sh.bw - 1) * cs;
        }
        const int tslot = sa <= sh.bw ? sa - sb : sa;
        const double* a = cj + sa * 36 + 6 * rr0;
        const double* b = cj + sb * 36 + 6 * cc0;
        double t = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m) t += a[m] * b[m];
        tcol[tslot * 36 + 6 * rr0 + cc0] -= t;
      }
    } else {  // the few corner columns: generic enumeration
      const int nsl = sh.nrow;
      for (int w = tid; w < nsl * nsl * 36; w += BND_THREADS) {
        const int pair = w / 36, e = w - 36 * pair, sa = pair / nsl, sb = pair - nsl * sa;
        if (sa < sb) continue;  // slots ascend with the row index inside a column
        const int i = bnd_row_of(sh, j, sa), k = bnd_row_of(sh, j, sb);
        if (i < 0 || k < 0 || k >= sh.Cf) continue;
        const int rr = e / 6, cc = e - 6 * rr;
        if (i >= sh.Cf && rr != 0) continue;
        const double* a = cj + sa * 36 + 6 * rr;
        const double* b = cj + sb * 36 + 6 * cc;
        double t = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m) t += a[m] * b[m];
        col_lds(k)[bnd_slot(sh, i, k) * 36 + 6 * rr + cc] -= t;
      }
    }
    if (++jm == nwin) jm = 0;
    __syncthreads();
    if (have_new) {  // visible to the next column's update after that column's barrier
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = tid + q * BND_THREADS;
        if (e < cs) cj[e] = fresh[q];
      }
    }
    // L goes out to global memory only now: loads and stores retire in order through one counter, so stores
    // issued before the wait for `fresh` would put their acknowledgement latency on every column's path
    if (mine) {
#pragma unroll
      for (int c = 0; c < 6; ++c) B[(long)j * cs + sl * 36 + 6 * r + c] = out[c];
    }
    if (tid < 36) B[(long)j * cs + ds * 36 + tid] = li_mine;
  }
  if (bad && tid == 0) atomic_add_f64(badv, 1.0);
}

// L^T x = y, one wave: x_j = inv(L_jj)^T (y_j - sum_{i in N(j)} L_ij^T x_i), j = Cf - 1 .. 0.  x of the
// window's cameras and of the arrow cameras stays in LDS; x goes to `x_out` [Cf][6].  (Staging column j - 1
// through registers into an LDS double buffer while column j is processed was slower: 1.44 vs 1.10 us a column.)
__global__ __launch_bounds__(64) void k_banded_backsolve(const double* __restrict__ B, BandedShape sh, const LmCtrl* ctrl,
                                                         double* __restrict__ x_out) {
  extern __shared__ double lds[];
  if (ctrl && ctrl->done) return;
  const int lane = threadIdx.x;
  const int cs = sh.nrow * 36, nwin = sh.bw + 1;
  double* xr = lds;                       // [nwin][6] ring of band solutions
  double* xa = xr + nwin * 6;             // [na][6]
  double* part = xa + max(sh.na, 1) * 6;  // [10][6] partial sums | then t[6]
  auto xptr = [&](int i) { return i < sh.Cb ? xr + (i % nwin) * 6 : xa + (i - sh.Cb) * 6; };
  const int g = lane / 6, c = lane - 6 * g;  // lane = (slot group g < 10, column c)
  for (int j = sh.Cf - 1; j >= 0; --j) {
    const double* cj = B + (long)j * cs;
    const int ds = bnd_diag_slot(sh, j);
    double acc = 0.0;
    if (g < 10) {
      for (int sl = g; sl < sh.nrow - 1; sl += 10) {
        const int i = bnd_row_of(sh, j, sl);
        if (i < 0) continue;
        const double* L = cj + sl * 36;
        const double* xi = xptr(i);
#pragma unroll
        for (int r = 0; r < 6; ++r) acc += L[6 * r + c] * xi[r];
      }
      part[g * 6 + c] = acc;
    }
    __syncthreads();
    if (lane < 6) {
      double t = cj[(sh.nrow - 1) * 36 + lane];  // y_j
      for (int q = 0; q < 10; ++q) t -= part[q * 6 + lane];
      part[60 + lane] = t;
    }
    __syncthreads();
    if (lane < 6) {
      const double* Li = cj + ds * 36;
      double x = 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) x += (m >= lane) ? Li[6 * m + lane] * part[60 + m] : 0.0;
      xptr(j)[lane] = x;
      x_out[6L * j + lane] = x;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Two-level variant for long trajectories (closed or open): the chain of C block columns is cut into P
// segments by separators of bw consecutive cameras each (ring order: sep 0 | interior 0 | sep 1 | interior 1 |
// ... | interior P-1, whose right-hand neighbour is sep 0 again -- the loop closure is just another band
// coupling).  The interior of a segment only couples with itself and with its two separators, so every
// segment is a banded problem with an arrow of 2 bw cameras (left separator, right separator) and is
// eliminated by ITS OWN workgroup, all at the same time (k_bseg_factor: the loop of k_banded_factor over the
// band columns only).  What the segments leave on their arrow blocks is added to the small system of the
// P bw separator cameras (k_bseg_corner_add), itself banded + arrow, which the one-workgroup kernels above
// factor and solve; the segments then back-substitute their interiors, again in parallel (k_bseg_backsolve).
// Sequential chain: C / P + P bw columns instead of C.
// Storage: one buffer [segment 0 | segment 1 | ... | level-2 system]; a segment is its interior columns followed
// by its 2 bw corner columns, all with nrow = bw + 1 + 2 bw + 1 slots.
struct BandedSeg {
  long base;       // offset (doubles) of the segment's storage
  int n_int;       // interior cameras (band columns)
  int first;       // free-camera index of the first interior camera; its left separator is [first - bw, first)
};

// The damped S and -rhs to precomputed places: dst[s] = offset of the 6x6 block of upper slot s, tr[s] = store
// transposed, rdst[f] = offset of the rhs row of free camera f.  B must be zero.
__global__ __launch_bounds__(256) void k_bseg_assemble(DeviceProblem d, double* __restrict__ B, const long* __restrict__ dst,
                                                       const unsigned char* __restrict__ tr, const long* __restrict__ rdst,
                                                       double lam, double min_diag, int solver_mode,
                                                       const int* __restrict__ cam_owner = nullptr, int rank = 0) {
  if (solver_mode) {
    if (d.ctrl->done) return;
    lam = d.ctrl->lam;
  }
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nS = (long)d.nu * 36;
  if (t < nS) {
    const int s = (int)(t / 36), e = (int)(t - 36L * s), r = e / 6, c = e - 6 * r;
    const int i = d.urow[s], j = d.ucol[s];
    double v = d.S[t];
    double* blk = B + dst[s];
    if (i == j) {
      if (c > r) return;
      // (sharded solve: d.S is this rank's partial sum; the damping comes from the camera's owner alone)
      if (r == c && (cam_owner == nullptr || cam_owner[i] == rank)) v += lam * fmax(u_diag(d, i, r), min_diag);
      blk[6 * r + c] = v;
      blk[6 * c + r] = v;
    } else if (tr[s]) {
      blk[6 * c + r] = v;
    } else {
      blk[6 * r + c] = v;
    }
  } else if (t < nS + 6L * d.Cf) {
    const long p = t - nS;
    const int k = (int)(p / 6), e = (int)(p - 6L * k);
    if (cam_owner == nullptr || cam_owner[k] == rank) B[rdst[k] + e] = -d.rhs[p];  // (sharded: rhs is all-reduced, one writer)
  }
}

// One workgroup per segment: eliminate the band columns (as k_banded_factor does), then write the updated
// corner columns back.
__global__ __launch_bounds__(BND_THREADS) void k_bseg_factor(double* __restrict__ Ball, const BandedSeg* __restrict__ segs, int bw,
                                                             const LmCtrl* ctrl, double* __restrict__ badv) {
  extern __shared__ double lds[];
  if (ctrl && ctrl->done) return;
  const BandedSeg sg = segs[blockIdx.x];
  const BandedShape sh{sg.n_int + 2 * bw, sg.n_int, bw, 2 * bw, 3 * bw + 2};
  double* __restrict__ B = Ball + sg.base;
  const int tid = threadIdx.x;
  const int cs = sh.nrow * 36, nwin = sh.bw + 1;
  double* win = lds;
  auto col_lds = [&](int k) { return win + (long)(k < sh.Cb ? k % nwin : nwin + (k - sh.Cb)) * cs; };
  for (int k = 0; k < min(nwin, sh.Cb); ++k)
    for (int e = tid; e < cs; e += BND_THREADS) col_lds(k)[e] = B[(long)k * cs + e];
  for (int k = sh.Cb; k < sh.Cf; ++k)
    for (int e = tid; e < cs; e += BND_THREADS) col_lds(k)[e] = B[(long)k * cs + e];
  unsigned char* pa = (unsigned char*)(win + (long)(nwin + sh.na) * cs);
  unsigned char* pb = pa + 512;
  __shared__ int npairs_s;
  if (tid == 0) {
    int np = 0;
    for (int sa = 1; sa < sh.nrow; ++sa)
      for (int sb = 1; sb <= min(sa, sh.nrow - 2); ++sb) { pa[np] = (unsigned char)sa; pb[np] = (unsigned char)sb; ++np; }
    npairs_s = np;
  }
  __syncthreads();
  const int npairs = npairs_s;
  const int pgrp = tid / 36, e0 = tid - 36 * pgrp, rr0 = e0 / 6, cc0 = e0 - 6 * rr0;
  int jm = 0;
  bool bad = false;
  for (int j = 0; j < sh.Cb; ++j) {
    double* cj = col_lds(j);
    const int knew = j + nwin;
    const bool have_new = knew < sh.Cb;
    double fresh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + q * BND_THREADS;
      fresh[q] = (have_new && e < cs) ? B[(long)knew * cs + e] : 0.0;
    }
    double Li[36];
    if (!chol6_inverse(cj, Li)) bad = true;
    const int sl = tid / 6, r = tid - 6 * sl;
    double out[6];
    bool mine = false;
    if (sl < sh.nrow && sl != 0) {
      const int i = bnd_row_of(sh, j, sl);
      if (i >= 0 && (i < sh.Cf || r == 0)) {
        mine = true;
        const double* src = cj + sl * 36 + 6 * r;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double t = 0.0;
#pragma unroll
          for (int m = 0; m <= c; ++m) t += src[m] * Li[6 * c + m];
          out[c] = t;
        }
      }
    }
    if (mine) {
#pragma unroll
      for (int c = 0; c < 6; ++c) cj[sl * 36 + 6 * r + c] = out[c];
    }
    double li_mine = 0.0;
#pragma unroll
    for (int e = 0; e < 36; ++e) li_mine = (e == tid) ? Li[e] : li_mine;
    __syncthreads();
    for (int p = pgrp < BND_THREADS / 36 ? pgrp : npairs; p < npairs; p += BND_THREADS / 36) {
      const int sa = pa[p], sb = pb[p];
      if ((sa <= sh.bw && j + sa >= sh.Cb) || (sb <= sh.bw && j + sb >= sh.Cb)) continue;
      if (sa == sh.nrow - 1 && rr0 != 0) continue;
      double* tcol;
      if (sb <= sh.bw) {
        int idx = jm + sb;
        if (idx >= nwin) idx -= nwin;
        tcol = win + (long)idx * cs;
      } else {
        tcol = win + (long)(nwin + sb - sh.bw - 1) * cs;
      }
      const int tslot = sa <= sh.bw ? sa - sb : sa;
      const double* a = cj + sa * 36 + 6 * rr0;
      const double* b = cj + sb * 36 + 6 * cc0;
      double t = 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) t += a[m] * b[m];
      tcol[tslot * 36 + 6 * rr0 + cc0] -= t;
    }
    if (++jm == nwin) jm = 0;
    __syncthreads();
    if (have_new) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = tid + q * BND_THREADS;
        if (e < cs) cj[e] = fresh[q];
      }
    }
    if (mine) {
#pragma unroll
      for (int c = 0; c < 6; ++c) B[(long)j * cs + sl * 36 + 6 * r + c] = out[c];
    }
    if (tid < 36) B[(long)j * cs + tid] = li_mine;
  }
  __syncthreads();
  for (int k = sh.Cb; k < sh.Cf; ++k)
    for (int e = tid; e < cs; e += BND_THREADS) B[(long)k * cs + e] = col_lds(k)[e];
  if (bad && tid == 0) atomic_add_f64(badv, 1.0);
}

// What segment b left on its arrow blocks, added to the level-2 system: pair q = (a, a'), a >= a', of its 2 bw
// arrow cameras goes to cdst[b][q] (transposed if ctr[b][q]); the rhs row of arrow camera a to crdst[b][a].
// Segments of one parity never share a level-2 block (P is even), so each launch takes one parity and adds
// with plain read-modify-writes: the sums are bitwise reproducible.
__global__ __launch_bounds__(256) void k_bseg_corner_add(double* __restrict__ Ball, const BandedSeg* __restrict__ segs, int bw,
                                                         int parity, const long* __restrict__ cdst,
                                                         const unsigned char* __restrict__ ctr, const long* __restrict__ crdst,
                                                         const LmCtrl* ctrl, int b0, int b1) {
  if (ctrl && ctrl->done) return;
  // segments [b0, b1) (all of them on one GPU; a rank's own in the sharded solve), those of this parity
  const int b = b0 + ((parity - b0) & 1) + 2 * blockIdx.x;
  if (b >= b1) return;
  const BandedSeg sg = segs[b];
  const int na = 2 * bw, nrow = 3 * bw + 2, cs = nrow * 36, npair = na * (na + 1) / 2;
  const double* __restrict__ C0 = Ball + sg.base + (long)sg.n_int * cs;  // corner columns
  for (int w = threadIdx.x; w < npair * 36 + na * 6; w += blockDim.x) {
    if (w < npair * 36) {
      const int q = w / 36, e = w - 36 * q, rr = e / 6, cc = e - 6 * rr;
      int a = 0;
      while ((a + 1) * (a + 2) / 2 <= q) ++a;  // q = a (a + 1) / 2 + a'
      const int a2 = q - a * (a + 1) / 2;
      const double v = C0[(long)a2 * cs + (bw + 1 + a) * 36 + 6 * rr + cc];  // block (row a, col a')
      double* dstp = Ball + cdst[(long)b * npair + q];
      // the level-2 system already holds the assembled (damped) blocks; a segment's diagonal corner blocks
      // were kept full, so both triangles are added there
      if (ctr[(long)b * npair + q]) dstp[6 * cc + rr] += v; else dstp[6 * rr + cc] += v;
    } else {
      const int w2 = w - npair * 36, a = w2 / 6, e = w2 - 6 * a;
      Ball[crdst[(long)b * na + a] + e] += C0[(long)a * cs + (long)(nrow - 1) * 36 + e];
    }
  }
}

// Interiors of all segments, one wave each: x of the segment's separators comes from the level-2 solution x2
// (level-2 index of arrow camera a of segment b: l2[b][a]); writes x of the interior and of the left separator
// to x_out [Cf][6].
__global__ __launch_bounds__(64) void k_bseg_backsolve(const double* __restrict__ Ball, const BandedSeg* __restrict__ segs, int bw,
                                                       const int* __restrict__ l2, const double* __restrict__ x2,
                                                       const LmCtrl* ctrl, double* __restrict__ x_out, int b0) {
  extern __shared__ double lds[];
  if (ctrl && ctrl->done) return;
  const int bseg = b0 + blockIdx.x;  // (b0: the first segment of this rank in the sharded solve, else 0)
  const BandedSeg sg = segs[bseg];
  const BandedShape sh{sg.n_int + 2 * bw, sg.n_int, bw, 2 * bw, 3 * bw + 2};
  const double* __restrict__ B = Ball + sg.base;
  const int lane = threadIdx.x;
  const int cs = sh.nrow * 36, nwin = sh.bw + 1;
  double* xr = lds;
  double* xa = xr + nwin * 6;
  double* part = xa + sh.na * 6;
  for (int w = lane; w < sh.na * 6; w += 64) {
    const int a = w / 6, e = w - 6 * a;
    const double v = x2[6L * l2[(long)bseg * sh.na + a] + e];
    xa[w] = v;
    if (a < bw) x_out[6L * (sg.first - bw + a) + e] = v;  // the left separator's cameras
  }
  __syncthreads();
  auto xptr = [&](int i) { return i < sh.Cb ? xr + (i % nwin) * 6 : xa + (i - sh.Cb) * 6; };
  const int g = lane / 6, c = lane - 6 * g;
  for (int j = sh.Cb - 1; j >= 0; --j) {
    const double* cj = B + (long)j * cs;
    double acc = 0.0;
    if (g < 10) {
      for (int sl = g; sl < sh.nrow - 1; sl += 10) {
        const int i = bnd_row_of(sh, j, sl);
        if (i < 0) continue;
        const double* L = cj + sl * 36;
        const double* xi = xptr(i);
#pragma unroll
        for (int r = 0; r < 6; ++r) acc += L[6 * r + c] * xi[r];
      }
      part[g * 6 + c] = acc;
    }
    __syncthreads();
    if (lane < 6) {
      double t = cj[(sh.nrow - 1) * 36 + lane];
      for (int q = 0; q < 10; ++q) t -= part[q * 6 + lane];
      part[60 + lane] = t;
    }
    __syncthreads();
    if (lane < 6) {
      double x = 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) x += (m >= lane) ? cj[6 * m + lane] * part[60 + m] : 0.0;
      xptr(j)[lane] = x;
      x_out[6L * (sg.first + j) + lane] = x;
    }
    __syncthreads();
  }
}
