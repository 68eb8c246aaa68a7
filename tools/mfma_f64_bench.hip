// Microbenchmark behind DESIGN.md "fp64 MFMA vs fp64 FMA": hipcc -O3 --offload-arch=gfx950 -o mfma_bench tools/mfma_f64_bench.hip
// Measured on MI355X: v_mfma_f64_16x16x4_f64 58.6 ns per instruction per wave (one wave per SIMD), 33-44 TFLOP/s chip-wide;
// v_fma_f64 2.5 ns per instruction per wave, 46-60 TFLOP/s chip-wide.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dvec4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  dvec4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = dvec4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void kf(double* out, int iters) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fma(a, b, acc[i]);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* d; (void)hipMalloc(&d, 256 * 1024 * 8 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int blocks : {1, 256, 1024}) {
    int iters = 4096;
    k<<<blocks, 256>>>(d, 16);
    (void)hipEventRecord(e0); k<<<blocks, 256>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)iters * 8;  // mfma per wave
    printf("mfma f64 16x16x4: blocks %d: %.3f ms, %.1f ns per MFMA per wave, %.2f TFLOP/s\n", blocks, ms, ms * 1e6 / mf,
           blocks * 4.0 * mf * 2048 / (ms * 1e-3) / 1e12);
    kf<<<blocks, 256>>>(d, 16);
    (void)hipEventRecord(e0); kf<<<blocks, 256>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("v_fma_f64: blocks %d: %.3f ms, %.2f ns per FMA per wave, %.2f TFLOP/s\n", blocks, ms, ms * 1e6 / (iters * 16.0),
           blocks * 4.0 * iters * 16.0 * 128 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
