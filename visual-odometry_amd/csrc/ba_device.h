// Device-side maths shared by the bundle-adjustment kernels (gfx950, fp64).
//
// Conventions are the reference's:
//   pose = [t(3), q(4) = w,x,y,z], R(q) rotates camera->world (reference src/geometry.py:192-213),
//   Xc = R^T (P - t)                       (reference src/camera_pose.py:85-86)
//   pixel = (fx x/z + cx, fy y/z + cy) [* radial factor]   (reference src/camera.py:134-165)
//   r = observed - projected               (reference src/bundle_adjustment.py:80-83)
// Local pose perturbation (6-DoF): t <- t + dt, q <- normalise(q (x) [1, dtheta/2]);
//   d r/d dt = +Jpi R^T, d r/d dtheta = -Jpi [Xc]x, d r/d P = -Jpi R^T.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Intrinsics {
  double fx, fy, cx, cy, k1, k2, k3;
  int has_dist;
};

// Robust loss on every scalar residual, with the semantics of scipy.optimize.least_squares' `loss=` /
// `f_scale=` arguments (scipy/optimize/_lsq/least_squares.py:186-236, the hook the reference's call at
// bundle_adjustment.py:201-212 leaves at its default 'linear'): cost = 0.5 C^2 sum rho((f / C)^2).
enum RobustKind { LOSS_LINEAR = 0, LOSS_SOFT_L1 = 1, LOSS_HUBER = 2, LOSS_CAUCHY = 3, LOSS_ARCTAN = 4 };
struct RobustLoss {
  int kind;
  double c2, inv_c2;  // f_scale^2 and its reciprocal
};

// 0.5 C^2 rho(z), z = f^2 / C^2
__device__ __forceinline__ double robust_cost(const RobustLoss& l, double f) {
  const double z = f * f * l.inv_c2;
  double rho;
  switch (l.kind) {
    case LOSS_SOFT_L1: rho = 2.0 * (sqrt(1.0 + z) - 1.0); break;
    case LOSS_HUBER: rho = z <= 1.0 ? z : 2.0 * sqrt(z) - 1.0; break;
    case LOSS_CAUCHY: rho = log1p(z); break;
    case LOSS_ARCTAN: rho = atan(z); break;
    default: rho = z;
  }
  return 0.5 * l.c2 * rho;
}

// Re-weighting of one scalar residual for the linearisation: f and its Jacobian row are both multiplied
// by s = sqrt(rho'(z)), so that J^T f is the gradient of the robust cost and J^T J = J^T rho' J majorises its
// Hessian (every rho here is concave in z).  scipy adds Triggs' second-order term, s^2 = max(rho' + 2 rho'' z,
// eps) (scipy/optimize/_lsq/common.py:scale_for_robust_loss_function); with that clamp the rows of outliers
// vanish from J^T J and from the Marquardt diagonal and LM rejects every step from a poor guess, so only the
// first-order weight is kept: same objective and minimisers, different Gauss-Newton model.
// Returns the cost term 0.5 C^2 rho(z).
__device__ __forceinline__ double robust_scale(const RobustLoss& l, double& f, double& s) {
  const double z = f * f * l.inv_c2;
  double rho, r1;
  switch (l.kind) {
    case LOSS_SOFT_L1: {
      const double t = 1.0 + z, it = 1.0 / sqrt(t);
      rho = 2.0 * (t * it - 1.0); r1 = it;
    } break;
    case LOSS_HUBER:
      if (z <= 1.0) { rho = z; r1 = 1.0; }
      else { const double iz = 1.0 / sqrt(z); rho = 2.0 * z * iz - 1.0; r1 = iz; }
      break;
    case LOSS_CAUCHY: rho = log1p(z); r1 = 1.0 / (1.0 + z); break;
    case LOSS_ARCTAN: rho = atan(z); r1 = 1.0 / (1.0 + z * z); break;
    default: rho = z; r1 = 1.0;
  }
  s = sqrt(r1);
  f *= s;
  return 0.5 * l.c2 * rho;
}

__device__ __forceinline__ void quat_to_R(const double* __restrict__ q, double R[9]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * y * y - 2 * z * z; R[1] = 2 * x * y - 2 * w * z;     R[2] = 2 * x * z + 2 * w * y;
  R[3] = 2 * x * y + 2 * w * z;     R[4] = 1 - 2 * x * x - 2 * z * z; R[5] = 2 * y * z - 2 * w * x;
  R[6] = 2 * x * z - 2 * w * y;     R[7] = 2 * y * z + 2 * w * x;     R[8] = 1 - 2 * x * x - 2 * y * y;
}

// Camera-frame point Xc = R^T (P - t).
__device__ __forceinline__ void to_camera(const double R[9], const double* __restrict__ t, const double P[3],
                                          double X[3]) {
  const double d0 = P[0] - t[0], d1 = P[1] - t[1], d2 = P[2] - t[2];
  X[0] = d0 * R[0] + d1 * R[3] + d2 * R[6];
  X[1] = d0 * R[1] + d1 * R[4] + d2 * R[7];
  X[2] = d0 * R[2] + d1 * R[5] + d2 * R[8];
}

// Residual only.
__device__ __forceinline__ void reproject(const Intrinsics& K, const double X[3], double u, double v, double& r0,
                                          double& r1) {
  const double iz = 1.0 / X[2];
  double xn = X[0] * iz, yn = X[1] * iz;
  if (K.has_dist) {
    const double r2 = xn * xn + yn * yn;
    const double f = 1.0 + K.k1 * r2 + K.k2 * r2 * r2 + K.k3 * r2 * r2 * r2;
    xn *= f;
    yn *= f;
  }
  r0 = u - (K.fx * xn + K.cx);
  r1 = v - (K.fy * yn + K.cy);
}

// Residual + Jpi (2x3) = d pixel / d Xc.
__device__ __forceinline__ void reproject_jac(const Intrinsics& K, const double X[3], double u, double v, double& r0,
                                              double& r1, double Jpi[6]) {
  const double iz = 1.0 / X[2];
  const double xn = X[0] * iz, yn = X[1] * iz;
  // d(xn,yn)/dX = [iz 0 -xn iz; 0 iz -yn iz]
  double a00 = K.fx, a01 = 0.0, a10 = 0.0, a11 = K.fy;  // d pixel / d (xn, yn)
  double xd = xn, yd = yn;
  if (K.has_dist) {
    const double r2 = xn * xn + yn * yn;
    const double f = 1.0 + K.k1 * r2 + K.k2 * r2 * r2 + K.k3 * r2 * r2 * r2;
    const double fp = K.k1 + 2.0 * K.k2 * r2 + 3.0 * K.k3 * r2 * r2;  // df / d r2
    xd = xn * f;
    yd = yn * f;
    a00 = K.fx * (f + 2.0 * xn * xn * fp);
    a01 = K.fx * (2.0 * xn * yn * fp);
    a10 = K.fy * (2.0 * xn * yn * fp);
    a11 = K.fy * (f + 2.0 * yn * yn * fp);
  }
  r0 = u - (K.fx * xd + K.cx);
  r1 = v - (K.fy * yd + K.cy);
  Jpi[0] = a00 * iz; Jpi[1] = a01 * iz; Jpi[2] = -(a00 * xn + a01 * yn) * iz;
  Jpi[3] = a10 * iz; Jpi[4] = a11 * iz; Jpi[5] = -(a10 * xn + a11 * yn) * iz;
}

// Full linearisation of one observation: residual, Jc (2x6 = [d/dt | d/dtheta]), Jp (2x3).
__device__ __forceinline__ void linearize_obs(const Intrinsics& K, const double* __restrict__ pose, const double P[3],
                                              double u, double v, double r[2], double Jc[12], double Jp[6],
                                              double& zc) {
  double R[9], X[3], Jpi[6];
  quat_to_R(pose + 3, R);
  to_camera(R, pose, P, X);
  zc = X[2];
  reproject_jac(K, X, u, v, r[0], r[1], Jpi);
  // A = Jpi R^T (2x3):  A[i][j] = sum_k Jpi[i][k] R[j][k]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double a = Jpi[3 * i + 0] * R[3 * j + 0] + Jpi[3 * i + 1] * R[3 * j + 1] + Jpi[3 * i + 2] * R[3 * j + 2];
      Jc[6 * i + j] = a;   // d r / d dt
      Jp[3 * i + j] = -a;  // d r / d P
    }
    // -Jpi [X]x ; [X]x = [0 -z y; z 0 -x; -y x 0]
    const double j0 = Jpi[3 * i + 0], j1 = Jpi[3 * i + 1], j2 = Jpi[3 * i + 2];
    Jc[6 * i + 3] = -(j1 * X[2] - j2 * X[1]);
    Jc[6 * i + 4] = -(-j0 * X[2] + j2 * X[0]);
    Jc[6 * i + 5] = -(j0 * X[1] - j1 * X[0]);
  }
}

// Inverse of the damped symmetric 3x3 block  Vd = V + lam * max(diag V, min_diag).
// v = (xx, xy, xz, yy, yz, zz); out in the same packing.  Returns the determinant.
__device__ __forceinline__ double damped_inv3(const double v[6], double lam, double min_diag, double o[6]) {
  const double a = v[0] + lam * fmax(v[0], min_diag), b = v[1], c = v[2];
  const double d = v[3] + lam * fmax(v[3], min_diag), e = v[4];
  const double f = v[5] + lam * fmax(v[5], min_diag);
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a * f - c * c) * id; o[4] = (b * c - a * e) * id;
  o[5] = (a * d - b * b) * id;
  return det;
}

// y = sym3(o) * x
__device__ __forceinline__ void sym3_mul(const double o[6], const double x[3], double y[3]) {
  y[0] = o[0] * x[0] + o[1] * x[1] + o[2] * x[2];
  y[1] = o[1] * x[0] + o[3] * x[1] + o[4] * x[2];
  y[2] = o[2] * x[0] + o[4] * x[1] + o[5] * x[2];
}

// Index of (i, j), i <= j, in the packed upper triangle of a 6x6 symmetric matrix (21 entries).
__host__ __device__ constexpr int sym6_index(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// In-place inverse of a 6x6 SPD matrix (row-major, full storage) by Cholesky; returns false if a
// pivot is not positive (the block is then replaced by the inverse of its diagonal).
__device__ inline bool spd_inverse6(double A[36]) {
  double Lm[36];
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 36; ++i) Lm[i] = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s = A[7 * j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= Lm[6 * j + k] * Lm[6 * j + k];
    if (!(s > 0.0)) { ok = false; s = 1.0; }
    const double d = sqrt(s);
    Lm[7 * j] = d;
    const double id = 1.0 / d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double t = A[6 * i + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= Lm[6 * i + k] * Lm[6 * j + k];
      Lm[6 * i + j] = t * id;
    }
  }
  if (!ok) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) A[6 * i + j] = (i == j) ? 1.0 / fmax(A[7 * i], 1e-300) : 0.0;
    return false;
  }
  // Li = L^-1 (lower), then A^-1 = Li^T Li
  double Li[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) Li[i] = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    Li[7 * j] = 1.0 / Lm[7 * j];
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double t = 0.0;
#pragma unroll
      for (int k = j; k < i; ++k) t -= Lm[6 * i + k] * Li[6 * k + j];
      Li[6 * i + j] = t / Lm[7 * i];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double t = 0.0;
#pragma unroll
      for (int k = i; k < 6; ++k) t += Li[6 * k + i] * Li[6 * k + j];
      A[6 * i + j] = t;
      A[6 * j + i] = t;
    }
  return true;
}

// fp64 atomic add that compiles to the hardware instruction (global_atomic_add_f64 / ds_add_f64)
// on gfx950; the memory is always hipMalloc'd (coarse-grained) or LDS.
__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

// max of non-negative doubles through their bit pattern (monotone for x >= 0)
__device__ __forceinline__ void atomic_max_nonneg(double* p, double v) {
  atomicMax(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(__double_as_longlong(v)));
}

// Sum over the 64 lanes of a wave, returned to every lane (wave-uniform: it comes back through
// v_readlane).  Four DPP steps build the 16-lane row sums (xor 1, xor 2, half-row mirror, row mirror: a
// few cycles each, no LDS crossbar round trips as ds_bpermute-based shuffles need), then the four row
// sums are read out of lanes 0 / 16 / 32 / 48.  Fixed summation order.
// 1 / x for a positive, finite, normal x: hardware estimate + two Newton steps (full fp64 precision,
// about a third of the dependent instructions of the IEEE division sequence)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}

template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_move_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move_f64<0x141>(v);  // row_half_mirror
  v += dpp_move_f64<0x140>(v);  // row_mirror
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Sum / max over aligned groups of WIDTH lanes, returned to every lane of the group (the same bits in all
// of them: every step adds the two partial sums of a pair).  Within a 16-lane row the steps are DPP moves
// (a few cycles each); ds_bpermute shuffles (~100 cycles each, measured) only cross rows.
template <int WIDTH>
__device__ __forceinline__ double group_sum(double v) {
  static_assert(WIDTH == 2 || WIDTH == 4 || WIDTH == 8 || WIDTH == 16 || WIDTH == 32 || WIDTH == 64, "group width");
  v += dpp_move_f64<0xB1>(v);                             // quad_perm [1,0,3,2]
  if constexpr (WIDTH >= 4) v += dpp_move_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  if constexpr (WIDTH >= 8) v += dpp_move_f64<0x141>(v);  // row_half_mirror
  if constexpr (WIDTH >= 16) v += dpp_move_f64<0x140>(v); // row_mirror
  if constexpr (WIDTH >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (WIDTH >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
template <int WIDTH>
__device__ __forceinline__ double group_max(double v) {
  v = fmax(v, dpp_move_f64<0xB1>(v));
  if constexpr (WIDTH >= 4) v = fmax(v, dpp_move_f64<0x4E>(v));
  if constexpr (WIDTH >= 8) v = fmax(v, dpp_move_f64<0x141>(v));
  if constexpr (WIDTH >= 16) v = fmax(v, dpp_move_f64<0x140>(v));
  if constexpr (WIDTH >= 32) v = fmax(v, __shfl_xor(v, 16, 64));
  if constexpr (WIDTH >= 64) v = fmax(v, __shfl_xor(v, 32, 64));
  return v;
}

// Sum over a 256-thread block; result valid in thread 0.  `red` = 4 doubles of LDS.
__device__ __forceinline__ double block_sum_256(double v, double* red) {
  v = group_sum<64>(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) s += red[i];
  }
  return s;
}
__device__ __forceinline__ double block_max_256(double v, double* red) {
  v = group_max<64>(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) s = fmax(s, red[i]);
  }
  return s;
}

// Wave-cooperative scatter-add.  Every lane with `valid` has staged NV doubles in LDS row
// stage[lane * STRIDE .. + NV) that must be added to dst[key .. key + NV).  Lanes whose keys are
// equal are summed first; then ONE atomic wave-instruction adds the NV contiguous doubles of a
// target (contiguous fp64 atomics run at the full memory-side rate, one-lane-per-row scatter an order
// of magnitude slower -- MI355X_MICROARCH.md "Global float atomics").  Call from ALL lanes of a
// 64-thread block (one wave), after a barrier that makes the staged rows visible.
template <int NV, int STRIDE>
__device__ __forceinline__ void wave_scatter_add(const double* stage, bool valid, long key, double* dst) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int b = __ffsll((long long)todo) - 1;
    const long kb = __shfl(key, b);
    const unsigned long long same = __ballot(valid && key == kb);
    if (lane < NV) {
      double s = 0.0;
      unsigned long long m = same;
      while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        s += stage[l * STRIDE + lane];
      }
      atomic_add_f64(dst + kb + lane, s);
    }
    todo &= ~same;
  }
}
