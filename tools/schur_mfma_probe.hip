// Probe behind DESIGN.md section 3 ("signature runs on the matrix unit"): the pair phase of k_schur_staged for a run of
// landmarks with ONE camera signature, two ways, on the same synthetic input (track length 5, `nl` landmarks per run):
//   valu: the formulation of k_schur_staged: a lane per (landmark of a group of four, camera pair), Z rows read from LDS,
//         126 FMAs per lane and landmark;
//   mfma: the run's stacked Z (30 rows + the row of L^T gp = 31 of 32, K = 3 per landmark) multiplied by its own
//         transpose with v_mfma_f64_16x16x4_f64: the three upper 16 x 16 tiles of the 32 x 32 product, K in steps of 4.
// Both form Z = W L per observation from the stored factors (12 doubles) the way the kernel does and write the run's
// 30 x 30 upper blocks + 30 rhs entries; the outputs are compared.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/schur_mfma_probe tools/schur_mfma_probe.hip
//   tools/bin/schur_mfma_probe [runs] [landmarks per run]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef double dvec4 __attribute__((ext_vector_type(4)));
constexpr int D = 5;             // track length
constexpr int LPC = 12;          // landmarks per chunk (60 observations <= 64 lanes)
constexpr int SZ_STRIDE = 22;    // valu: doubles per observation row of Z in LDS
constexpr int KC = 3 * LPC;      // mfma: K columns per chunk
constexpr int LDK = KC + 1;      // mfma: row stride of the stacked Z in LDS (37: 16 rows x 4 columns spread over the banks)

// Z = W L for one observation (jw: the 12 stored factors; l: L of the landmark), as k_schur_staged forms it
__device__ __forceinline__ void form_z(const double jw[12], const double* l, double z[18]) {
  const double l00 = l[0], l10 = l[1], l20 = l[2], l11 = l[3], l21 = l[4], l22 = l[5];
  const double q0[3] = {jw[0] * l00 + jw[1] * l10 + jw[2] * l20, jw[1] * l11 + jw[2] * l21, jw[2] * l22};
  const double q1[3] = {jw[3] * l00 + jw[4] * l10 + jw[5] * l20, jw[4] * l11 + jw[5] * l21, jw[5] * l22};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      z[3 * i + c] = -(jw[i] * q0[c] + jw[3 + i] * q1[c]);
      z[9 + 3 * i + c] = jw[6 + i] * q0[c] + jw[9 + i] * q1[c];
    }
}

// ---------------------------------------------------------------- valu: one wave per run
__global__ __launch_bounds__(64, 2) void k_valu(const double* __restrict__ W, const double* __restrict__ Lg, long n, int nl,
                                                double* __restrict__ out) {
  __shared__ __align__(16) double sZ[64 * SZ_STRIDE];
  __shared__ double sL[LPC * 9];
  const int lane = threadIdx.x, run = blockIdx.x;
  const int gl = lane & 15, grp = lane >> 4;
  int b0 = 0;
  while ((b0 + 1) * (b0 + 2) / 2 <= gl) ++b0;
  const int a0 = gl - b0 * (b0 + 1) / 2;
  double acc[36], accr[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) accr[i] = 0.0;
  double pw[12], pl[9];
  auto issue = [&](int c) {
    const long k0 = ((long)run * nl + (long)c * LPC) * D;
    const int nc = min(LPC, nl - c * LPC);
    if (lane < nc * D)
#pragma unroll
      for (int r = 0; r < 12; ++r) pw[r] = W[(long)r * n + k0 + lane];
    if (lane < nc)
#pragma unroll
      for (int r = 0; r < 9; ++r) pl[r] = Lg[((long)run * nl + (long)c * LPC + lane) * 9 + r];
  };
  const int n_chunks = (nl + LPC - 1) / LPC;
  issue(0);
  for (int c = 0; c < n_chunks; ++c) {
    const int nc = min(LPC, nl - c * LPC);
    __syncthreads();
    if (lane < nc)
#pragma unroll
      for (int r = 0; r < 9; ++r) sL[lane * 9 + r] = pl[r];
    __syncthreads();
    if (lane < nc * D) {
      double z[18];
      form_z(pw, &sL[(lane / D) * 9], z);
#pragma unroll
      for (int i = 0; i < 18; ++i) sZ[lane * SZ_STRIDE + i] = z[i];
    }
    __syncthreads();
    if (c + 1 < n_chunks) issue(c + 1);
    for (int l0 = 0; l0 < nc; l0 += 4) {
      const int li = l0 + grp;
      if (li < nc && gl < 15) {
        const int kb = li * D;
        const double* __restrict__ Ya = &sZ[(kb + a0) * SZ_STRIDE];
        const double* __restrict__ Wb = &sZ[(kb + b0) * SZ_STRIDE];
        double Y[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) Y[i] = Ya[i];
        if (a0 == b0) {
          const double g0 = sL[li * 9 + 6], g1 = sL[li * 9 + 7], g2 = sL[li * 9 + 8];
#pragma unroll
          for (int i = 0; i < 6; ++i) accr[i] = fma(Y[3 * i + 2], g2, fma(Y[3 * i + 1], g1, fma(Y[3 * i], g0, accr[i])));
        }
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
          const double w0 = Wb[3 * cc], w1 = Wb[3 * cc + 1], w2 = Wb[3 * cc + 2];
#pragma unroll
          for (int i = 0; i < 6; ++i) acc[6 * i + cc] = fma(Y[3 * i + 2], w2, fma(Y[3 * i + 1], w1, fma(Y[3 * i], w0, acc[6 * i + cc])));
        }
      }
    }
  }
  // the four groups hold partial sums of the same 15 pairs: add them (the real kernel writes all four rows)
#pragma unroll
  for (int i = 0; i < 36; ++i) { acc[i] += __shfl_xor(acc[i], 16); acc[i] += __shfl_xor(acc[i], 32); }
#pragma unroll
  for (int i = 0; i < 6; ++i) { accr[i] += __shfl_xor(accr[i], 16); accr[i] += __shfl_xor(accr[i], 32); }
  if (lane < 15) {
    double* o = out + (long)run * 960;  // [30][32]: the 30 x 30 product | column 30 = rhs
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) o[(6 * a0 + i) * 32 + 6 * b0 + cc] = acc[6 * i + cc];
    if (a0 == b0)
#pragma unroll
      for (int i = 0; i < 6; ++i) o[(6 * a0 + i) * 32 + 30] = accr[i];
  }
}

// ---------------------------------------------------------------- valu with the next step's Z_a rows fetched from LDS while the current pairs are multiplied
__global__ __launch_bounds__(64, 2) void k_valu_pf(const double* __restrict__ W, const double* __restrict__ Lg, long n, int nl,
                                                double* __restrict__ out) {
  __shared__ __align__(16) double sZ[64 * SZ_STRIDE];
  __shared__ double sL[LPC * 9];
  const int lane = threadIdx.x, run = blockIdx.x;
  const int gl = lane & 15, grp = lane >> 4;
  int b0 = 0;
  while ((b0 + 1) * (b0 + 2) / 2 <= gl) ++b0;
  const int a0 = gl - b0 * (b0 + 1) / 2;
  double acc[36], accr[6];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) accr[i] = 0.0;
  double pw[12], pl[9];
  auto issue = [&](int c) {
    const long k0 = ((long)run * nl + (long)c * LPC) * D;
    const int nc = min(LPC, nl - c * LPC);
    if (lane < nc * D)
#pragma unroll
      for (int r = 0; r < 12; ++r) pw[r] = W[(long)r * n + k0 + lane];
    if (lane < nc)
#pragma unroll
      for (int r = 0; r < 9; ++r) pl[r] = Lg[((long)run * nl + (long)c * LPC + lane) * 9 + r];
  };
  const int n_chunks = (nl + LPC - 1) / LPC;
  issue(0);
  for (int c = 0; c < n_chunks; ++c) {
    const int nc = min(LPC, nl - c * LPC);
    __syncthreads();
    if (lane < nc)
#pragma unroll
      for (int r = 0; r < 9; ++r) sL[lane * 9 + r] = pl[r];
    __syncthreads();
    if (lane < nc * D) {
      double z[18];
      form_z(pw, &sL[(lane / D) * 9], z);
#pragma unroll
      for (int i = 0; i < 18; ++i) sZ[lane * SZ_STRIDE + i] = z[i];
    }
    __syncthreads();
    if (c + 1 < n_chunks) issue(c + 1);
    double Y[18];
    {
      const int li = grp;
      if (li < nc && gl < 15) {
        const double* __restrict__ Ya = &sZ[(li * D + a0) * SZ_STRIDE];
#pragma unroll
        for (int i = 0; i < 18; ++i) Y[i] = Ya[i];
      }
    }
#pragma unroll 2
    for (int l0 = 0; l0 < nc; l0 += 4) {
      const int li = l0 + grp;
      double Yn[18];
      if (li + 4 < nc && gl < 15) {  // the rows of the next step, in flight during this step's products
        const double* __restrict__ Yb = &sZ[((li + 4) * D + a0) * SZ_STRIDE];
#pragma unroll
        for (int i = 0; i < 18; ++i) Yn[i] = Yb[i];
      }
      if (li < nc && gl < 15) {
        const int kb = li * D;
        const double* __restrict__ Wb = &sZ[(kb + b0) * SZ_STRIDE];
        if (a0 == b0) {
          const double g0 = sL[li * 9 + 6], g1 = sL[li * 9 + 7], g2 = sL[li * 9 + 8];
#pragma unroll
          for (int i = 0; i < 6; ++i) accr[i] = fma(Y[3 * i + 2], g2, fma(Y[3 * i + 1], g1, fma(Y[3 * i], g0, accr[i])));
        }
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
          const double w0 = Wb[3 * cc], w1 = Wb[3 * cc + 1], w2 = Wb[3 * cc + 2];
#pragma unroll
          for (int i = 0; i < 6; ++i) acc[6 * i + cc] = fma(Y[3 * i + 2], w2, fma(Y[3 * i + 1], w1, fma(Y[3 * i], w0, acc[6 * i + cc])));
        }
      }
#pragma unroll
      for (int i = 0; i < 18; ++i) Y[i] = Yn[i];
    }
  }
  // the four groups hold partial sums of the same 15 pairs: add them (the real kernel writes all four rows)
#pragma unroll
  for (int i = 0; i < 36; ++i) { acc[i] += __shfl_xor(acc[i], 16); acc[i] += __shfl_xor(acc[i], 32); }
#pragma unroll
  for (int i = 0; i < 6; ++i) { accr[i] += __shfl_xor(accr[i], 16); accr[i] += __shfl_xor(accr[i], 32); }
  if (lane < 15) {
    double* o = out + (long)run * 960;  // [30][32]: the 30 x 30 product | column 30 = rhs
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) o[(6 * a0 + i) * 32 + 6 * b0 + cc] = acc[6 * i + cc];
    if (a0 == b0)
#pragma unroll
      for (int i = 0; i < 6; ++i) o[(6 * a0 + i) * 32 + 30] = accr[i];
  }
}

// ---------------------------------------------------------------- valu, half a block per lane: two lanes per camera pair (rows 0-2 and 3-5
// of its 6 x 6 block), two landmarks per step: 18 + 3 accumulators per lane instead of 36 + 6
template <int MINW>
__global__ __launch_bounds__(64, MINW) void k_valu_half(const double* __restrict__ W, const double* __restrict__ Lg, long n, int nl,
                                                        double* __restrict__ out) {
  __shared__ __align__(16) double sZ[64 * SZ_STRIDE];
  __shared__ double sL[LPC * 9];
  const int lane = threadIdx.x, run = blockIdx.x;
  const int gl = lane & 31, grp = lane >> 5, pr = gl >> 1, hf = gl & 1;
  int b0 = 0;
  while ((b0 + 1) * (b0 + 2) / 2 <= pr) ++b0;
  const int a0 = pr - b0 * (b0 + 1) / 2;
  double acc[18], accr[3];
#pragma unroll
  for (int i = 0; i < 18; ++i) acc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) accr[i] = 0.0;
  double pw[12], pl[9];
  auto issue = [&](int c) {
    const long k0 = ((long)run * nl + (long)c * LPC) * D;
    const int nc = min(LPC, nl - c * LPC);
    if (lane < nc * D)
#pragma unroll
      for (int r = 0; r < 12; ++r) pw[r] = W[(long)r * n + k0 + lane];
    if (lane < nc)
#pragma unroll
      for (int r = 0; r < 9; ++r) pl[r] = Lg[((long)run * nl + (long)c * LPC + lane) * 9 + r];
  };
  const int n_chunks = (nl + LPC - 1) / LPC;
  issue(0);
  for (int c = 0; c < n_chunks; ++c) {
    const int nc = min(LPC, nl - c * LPC);
    __syncthreads();
    if (lane < nc)
#pragma unroll
      for (int r = 0; r < 9; ++r) sL[lane * 9 + r] = pl[r];
    __syncthreads();
    if (lane < nc * D) {
      double z[18];
      form_z(pw, &sL[(lane / D) * 9], z);
#pragma unroll
      for (int i = 0; i < 18; ++i) sZ[lane * SZ_STRIDE + i] = z[i];
    }
    __syncthreads();
    if (c + 1 < n_chunks) issue(c + 1);
    for (int l0 = 0; l0 < nc; l0 += 2) {
      const int li = l0 + grp;
      if (li < nc && pr < 15) {
        const int kb = li * D;
        const double* __restrict__ Ya = &sZ[(kb + a0) * SZ_STRIDE + 9 * hf];
        const double* __restrict__ Wb = &sZ[(kb + b0) * SZ_STRIDE];
        double Y[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Y[i] = Ya[i];
        if (a0 == b0) {
          const double g0 = sL[li * 9 + 6], g1 = sL[li * 9 + 7], g2 = sL[li * 9 + 8];
#pragma unroll
          for (int i = 0; i < 3; ++i) accr[i] = fma(Y[3 * i + 2], g2, fma(Y[3 * i + 1], g1, fma(Y[3 * i], g0, accr[i])));
        }
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
          const double w0 = Wb[3 * cc], w1 = Wb[3 * cc + 1], w2 = Wb[3 * cc + 2];
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[6 * i + cc] = fma(Y[3 * i + 2], w2, fma(Y[3 * i + 1], w1, fma(Y[3 * i], w0, acc[6 * i + cc])));
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 18; ++i) acc[i] += __shfl_xor(acc[i], 32);
#pragma unroll
  for (int i = 0; i < 3; ++i) accr[i] += __shfl_xor(accr[i], 32);
  if (lane < 30) {
    double* o = out + (long)run * 960;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) o[(6 * a0 + 3 * hf + i) * 32 + 6 * b0 + cc] = acc[6 * i + cc];
    if (a0 == b0)
#pragma unroll
      for (int i = 0; i < 3; ++i) o[(6 * a0 + 3 * hf + i) * 32 + 30] = accr[i];
  }
}

// ---------------------------------------------------------------- mfma: one wave per run, WAVES waves per block
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_mfma(const double* __restrict__ W, const double* __restrict__ Lg, long n, int nl,
                                                     int runs, double* __restrict__ out) {
  __shared__ double sZall[WAVES][2][32 * LDK];  // per wave, double-buffered: rows 0..29 = Z stacked by camera, 30 = L^T gp, 31 = 0
  __shared__ double sLall[WAVES][LPC * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int run = blockIdx.x * WAVES + wv;
  if (run >= runs) return;
  double* sL = sLall[wv];
  dvec4 acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  double pw[12], pl[9];
  auto issue = [&](int c) {
    const long k0 = ((long)run * nl + (long)c * LPC) * D;
    const int nc = min(LPC, nl - c * LPC);
    if (lane < nc * D)
#pragma unroll
      for (int r = 0; r < 12; ++r) pw[r] = W[(long)r * n + k0 + lane];
    if (lane < nc)
#pragma unroll
      for (int r = 0; r < 9; ++r) pl[r] = Lg[((long)run * nl + (long)c * LPC + lane) * 9 + r];
  };
  const int n_chunks = (nl + LPC - 1) / LPC;
  for (int i = lane; i < 2 * 32 * LDK; i += 64) sZall[wv][0][i] = 0.0;  // row 31 and the padding stay zero
  issue(0);
  const int row = lane & 15, kq = lane >> 4;
  for (int c = 0; c < n_chunks; ++c) {
    const int nc = min(LPC, nl - c * LPC);
    double* sZ = sZall[wv][c & 1];
    // a wave's LDS operations complete in order: no workgroup barrier, the waves of a block share nothing
    if (lane < nc)
#pragma unroll
      for (int r = 0; r < 9; ++r) sL[lane * 9 + r] = pl[r];
    __builtin_amdgcn_wave_barrier();
    if (lane < nc * D) {
      const int l = lane / D, a = lane - D * l;
      double z[18];
      form_z(pw, &sL[l * 9], z);
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) sZ[(6 * a + i) * LDK + 3 * l + cc] = z[3 * i + cc];
    } else if (lane >= 60 && lane - 60 < 3) {
      const int cc = lane - 60;
      for (int l = 0; l < LPC; ++l) sZ[30 * LDK + 3 * l + cc] = l < nc ? sL[l * 9 + 6 + cc] : 0.0;
    }
    if (nc < LPC && lane < 30)  // a short last chunk: its unused columns must not keep the other buffer's values
      for (int k = 3 * nc; k < KC; ++k) sZ[lane * LDK + k] = 0.0;
    __builtin_amdgcn_wave_barrier();
    if (c + 1 < n_chunks) issue(c + 1);
#pragma unroll
    for (int k0 = 0; k0 < KC; k0 += 4) {
      const double z0 = sZ[row * LDK + k0 + kq], z1 = sZ[(16 + row) * LDK + k0 + kq];
      acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0, z0, acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0, z1, acc01, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(z1, z1, acc11, 0, 0, 0);
    }
  }
  // C / D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  double* o = out + (long)run * 960;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = kq + 4 * q;
    o[r * 32 + row] = acc00[q];
    o[r * 32 + 16 + row] = acc01[q];
    if (16 + r < 30) o[(16 + r) * 32 + 16 + row] = acc11[q];
  }
}

int main(int argc, char** argv) {
  const int runs = argc > 1 ? std::atoi(argv[1]) : 1000, nl = argc > 2 ? std::atoi(argv[2]) : 200;
  const long n = (long)runs * nl * D, L = (long)runs * nl;
  std::vector<double> hW(12 * n), hL(9 * L);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24) - 0.5; };
  for (auto& v : hW) v = rnd();
  for (long j = 0; j < L; ++j) {
    for (int r = 0; r < 9; ++r) hL[9 * j + r] = 0.3 * rnd();
    hL[9 * j] += 1.0; hL[9 * j + 3] += 1.0; hL[9 * j + 5] += 1.0;
  }
  double *dW, *dL, *o1, *o2;
  CHECK(hipMalloc(&dW, hW.size() * 8)); CHECK(hipMalloc(&dL, hL.size() * 8));
  CHECK(hipMalloc(&o1, (size_t)runs * 960 * 8)); CHECK(hipMalloc(&o2, (size_t)runs * 960 * 8));
  CHECK(hipMemcpy(dW, hW.data(), hW.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dL, hL.data(), hL.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemset(o1, 0, (size_t)runs * 960 * 8)); CHECK(hipMemset(o2, 0, (size_t)runs * 960 * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto time_it = [&](const char* name, auto&& launch) {
    launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    CHECK(hipGetLastError());
    const double macs = (double)L * (15 * 108 + 5 * 18 + D * 48);
    std::printf("%-22s %9.1f us   %6.2f TFLOP/s useful (pairs + rhs + Z)\n", name, best * 1e3, 2 * macs / (best * 1e-3) / 1e12);
  };
  std::printf("%d runs x %d landmarks x %d observations (%ld observations)\n", runs, nl, D, n);
  time_it("valu, wave per run", [&] { k_valu<<<runs, 64>>>(dW, dL, n, nl, o1); });
  time_it("valu, rows prefetched", [&] { k_valu_pf<<<runs, 64>>>(dW, dL, n, nl, o2); });
  {
    std::vector<double> h1((size_t)runs * 960), h2((size_t)runs * 960);
    CHECK(hipMemcpy(h1.data(), o1, h1.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h2.data(), o2, h2.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (size_t i = 0; i < h1.size(); ++i) worst = std::fmax(worst, std::fabs(h1[i] - h2[i]));
    std::printf("  (prefetched against plain: largest difference %.3e)\n", worst);
  }
  for (int v = 0; v < 3; ++v) {
    CHECK(hipMemset(o2, 0, (size_t)runs * 960 * 8));
    if (v == 0) time_it("valu, half blocks, 2 w", [&] { k_valu_half<2><<<runs, 64>>>(dW, dL, n, nl, o2); });
    if (v == 1) time_it("valu, half blocks, 3 w", [&] { k_valu_half<3><<<runs, 64>>>(dW, dL, n, nl, o2); });
    if (v == 2) time_it("valu, half blocks, 4 w", [&] { k_valu_half<4><<<runs, 64>>>(dW, dL, n, nl, o2); });
    std::vector<double> h1((size_t)runs * 960), h2((size_t)runs * 960);
    CHECK(hipMemcpy(h1.data(), o1, h1.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h2.data(), o2, h2.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int r = 0; r < runs; ++r)
      for (int i = 0; i < 30; ++i)
        for (int j = i; j < 31; ++j) worst = std::fmax(worst, std::fabs(h1[(size_t)r * 960 + i * 32 + j] - h2[(size_t)r * 960 + i * 32 + j]));
    std::printf("  (half blocks against plain: largest difference %.3e)\n", worst);
  }
  time_it("mfma, 1 wave / block", [&] { k_mfma<1><<<runs, 64>>>(dW, dL, n, nl, runs, o2); });
  time_it("mfma, 2 waves / block", [&] { k_mfma<2><<<(runs + 1) / 2, 128>>>(dW, dL, n, nl, runs, o2); });
  time_it("mfma, 4 waves / block", [&] { k_mfma<4><<<(runs + 3) / 4, 256>>>(dW, dL, n, nl, runs, o2); });
  std::vector<double> h1((size_t)runs * 960), h2((size_t)runs * 960);
  CHECK(hipMemcpy(h1.data(), o1, h1.size() * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(h2.data(), o2, h2.size() * 8, hipMemcpyDeviceToHost));
  double worst = 0.0, scale = 0.0;
  for (int r = 0; r < runs; ++r)
    for (int i = 0; i < 30; ++i)
      for (int j = i; j < 31; ++j) {  // upper triangle of the 30 x 30 product | column 30 = rhs
        const double a = h1[(size_t)r * 960 + i * 32 + j], b = h2[(size_t)r * 960 + i * 32 + j];
        worst = std::fmax(worst, std::fabs(a - b));
        scale = std::fmax(scale, std::fabs(a));
      }
  std::printf("largest difference between the two outputs: %.3e (largest entry %.3e)\n", worst, scale);
  return worst <= 1e-9 * scale ? 0 : 2;
}
