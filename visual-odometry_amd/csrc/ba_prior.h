// Quadratic pose prior (ba_set_pose_prior): E = c0 + g^T d + 0.5 d^T H d over the chart coordinates d of a few cameras.
// Two one-workgroup kernels: k_prior_add puts the prior's Gauss-Newton terms into the assembled reduced system (U / gc,
// S, rhs) and its energy into the cost of the linearisation point; k_prior_trial adds the energy of the trial poses
// to the trial cost and the prior's camera-to-camera cross terms to the model decrease.  A few hundred flops each:
// latency only.  No counterpart in the reference.
#pragma once
#include "ba_kernels.h"

// d (6) of one pose in the chart around ref, and J = d d_rot / d dtheta (row-major 3x3) of the retraction
// q <- normalise(q (x) [1, dtheta / 2]):  q = ref (x) [1, a] / |.|  =>  a' = (a + e + a x e) / (1 - a.e),  e = dtheta / 2.
__device__ __forceinline__ void prior_chart(const double* __restrict__ ps, const double* __restrict__ ref, double del[6],
                                            double J[9]) {
  del[0] = ps[0] - ref[0]; del[1] = ps[1] - ref[1]; del[2] = ps[2] - ref[2];
  const double rw = ref[3], rx = ref[4], ry = ref[5], rz = ref[6];
  const double qw = ps[3], qx = ps[4], qy = ps[5], qz = ps[6];
  const double w = rw * qw + rx * qx + ry * qy + rz * qz;  // conj(ref) (x) q
  const double x = rw * qx - rx * qw - ry * qz + rz * qy;
  const double y = rw * qy + rx * qz - ry * qw - rz * qx;
  const double z = rw * qz - rx * qy + ry * qx - rz * qw;
  const double iw = 1.0 / w;
  const double a0 = x * iw, a1 = y * iw, a2 = z * iw;
  del[3] = 2.0 * a0; del[4] = 2.0 * a1; del[5] = 2.0 * a2;
  J[0] = 1.0 + a0 * a0; J[1] = -a2 + a0 * a1;     J[2] = a1 + a0 * a2;
  J[3] = a2 + a1 * a0;  J[4] = 1.0 + a1 * a1;     J[5] = -a0 + a1 * a2;
  J[6] = -a1 + a2 * a0; J[7] = a0 + a2 * a1;      J[8] = 1.0 + a2 * a2;
}

// Entry (k, r) of the 6x6 chart Jacobian of prior camera a: identity on the translation, J on the rotation.
__device__ __forceinline__ double prior_J(const double (*sJ)[9], int a, int k, int r) {
  if (k < 3 || r < 3) return k == r ? 1.0 : 0.0;
  return sJ[a][3 * (k - 3) + (r - 3)];
}

__global__ __launch_bounds__(256) void k_prior_add(DeviceProblem d, const double* __restrict__ poses) {
  __shared__ double sdel[6 * PRIOR_MAX_CAMS], sv[6 * PRIOR_MAX_CAMS], sJ[PRIOR_MAX_CAMS][9];
  const PriorDev& p = d.prior;
  if (!poses) {
    if (d.ctrl->done) return;
    poses = d.poses_buf[d.ctrl->sel];
  }
  const int np = p.n, n6 = 6 * np, t = threadIdx.x;
  if (t < np) prior_chart(poses + 7 * p.cam[t], p.ref + 7 * t, &sdel[6 * t], sJ[t]);
  __syncthreads();
  if (t < n6) {
    double a = p.g[t];
    for (int j = 0; j < n6; ++j) a += p.H[(long)t * n6 + j] * sdel[j];
    sv[t] = a;  // H d + g
  }
  __syncthreads();
  if (t < n6) {  // gradient of the energy in the step coordinates: J^T (H d + g)
    const int a = t / 6, i = t - 6 * a;
    double gj = 0.0;
    for (int k = 0; k < 6; ++k) gj += prior_J(sJ, a, k, i) * sv[6 * a + k];
    d.rhs[6 * p.fr[a] + i] += gj;
    d.Ugc[(long)p.cam[a] * 27 + 21 + i] += gj;
  }
  if (t == 0) {
    double e = p.c0;
    for (int i = 0; i < n6; ++i) e += sdel[i] * 0.5 * (sv[i] + p.g[i]);
    atomic_add_f64(&d.scal[SC_COST], e);
  }
  // J_a^T H_ab J_b into S (upper blocks) and, on the diagonal, into U
  for (int item = t; item < np * np * 36; item += 256) {
    const int a = item / (np * 36), rem = item - a * np * 36, b = rem / 36, e = rem - 36 * b, r = e / 6, c = e - 6 * r;
    const int slot = p.slot[a * np + b];
    if (slot < 0) continue;  // the lower triangle: the transpose of a pair that is handled
    double val = 0.0;
    for (int k = (r < 3 ? r : 3); k < (r < 3 ? r + 1 : 6); ++k) {
      const double jk = prior_J(sJ, a, k, r);
      for (int l = (c < 3 ? c : 3); l < (c < 3 ? c + 1 : 6); ++l)
        val += jk * p.H[(long)(6 * a + k) * n6 + 6 * b + l] * prior_J(sJ, b, l, c);
    }
    d.S[(long)slot * 36 + e] += val;
    if (a == b && r <= c) d.Ugc[(long)p.cam[a] * 27 + sym6_index(r, c)] += val;
  }
}

// Energy at the trial poses -> trial cost; sum over a != b of dc_a^T (J_a^T H_ab J_b) dc_b -> model curvature (the
// diagonal blocks are part of U and are counted by the pose blocks of k_backsub).
__global__ __launch_bounds__(128) void k_prior_trial(DeviceProblem d, const double* __restrict__ poses,
                                                     const double* __restrict__ poses_t) {
  __shared__ double sdel[6 * PRIOR_MAX_CAMS], su[6 * PRIOR_MAX_CAMS], sJ[PRIOR_MAX_CAMS][9], se[6 * PRIOR_MAX_CAMS],
      sx[6 * PRIOR_MAX_CAMS];
  const PriorDev& p = d.prior;
  if (!poses) {
    if (d.ctrl->done) return;
    poses = d.poses_buf[d.ctrl->sel];
    poses_t = d.poses_buf[d.ctrl->sel ^ 1];
  }
  const int np = p.n, n6 = 6 * np, t = threadIdx.x;
  if (t < np) {
    double Jt[9];
    prior_chart(poses_t + 7 * p.cam[t], p.ref + 7 * t, &sdel[6 * t], Jt);
    double dl[6];
    prior_chart(poses + 7 * p.cam[t], p.ref + 7 * t, dl, sJ[t]);
  }
  __syncthreads();
  if (t < n6) {  // u = J dc (the change of the chart coordinates that the model sees)
    const int a = t / 6, k = t - 6 * a;
    double u = 0.0;
    for (int r = 0; r < 6; ++r) u += prior_J(sJ, a, k, r) * d.dc[6 * p.fr[a] + r];
    su[t] = u;
  }
  __syncthreads();
  if (t < n6) {
    const int a = t / 6;
    double hd = 0.0, hu = 0.0;
    for (int j = 0; j < n6; ++j) {
      const double hij = p.H[(long)t * n6 + j];
      hd += hij * sdel[j];
      if (j / 6 != a) hu += hij * su[j];
    }
    se[t] = sdel[t] * (p.g[t] + 0.5 * hd);
    sx[t] = su[t] * hu;
  }
  __syncthreads();
  if (t == 0) {
    double e = p.c0, x = 0.0;
    for (int i = 0; i < n6; ++i) { e += se[i]; x += sx[i]; }
    atomic_add_f64(&d.scal[SC_COST_T], e);
    atomic_add_f64(&d.scal[SC_DHD_C], x);
  }
}
