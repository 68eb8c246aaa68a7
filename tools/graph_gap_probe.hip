// Does a HIP graph shorten the boundary between dependent kernels?  A chain of N small kernels (each reads what the
// previous one wrote), enqueued (a) launch by launch into a stream that is already busy (the host runs ahead, as
// lm_solve does), (b) as one captured graph launched R times.  Prints the time per kernel of both.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/graph_gap_probe tools/graph_gap_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_step(double* buf, int n, int work) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = buf[i];
  for (int k = 0; k < work; ++k) x = x * 1.0000001 + 1e-9;
  buf[(i + 1) % n] = x;  // the next kernel's thread i + 1 depends on it
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 20, R = argc > 2 ? atoi(argv[2]) : 200;
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int blocks : {1, 64, 1024}) {
    for (int work : {0, 2000}) {
      const int n = blocks * 256;
      double* buf; CK(hipMalloc(&buf, n * sizeof(double))); CK(hipMemset(buf, 0, n * sizeof(double)));
      auto chain = [&]() { for (int k = 0; k < N; ++k) k_step<<<blocks, 256, 0, s>>>(buf, n, work); };
      for (int r = 0; r < 20; ++r) chain();
      CK(hipStreamSynchronize(s));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < R; ++r) chain();
      CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms_stream; CK(hipEventElapsedTime(&ms_stream, e0, e1));
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      chain();
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms_graph; CK(hipEventElapsedTime(&ms_graph, e0, e1));
      printf("blocks %5d work %5d: stream %.3f us per kernel, graph %.3f us per kernel (chain of %d, %d repeats)\n", blocks, work,
             1e3 * ms_stream / (R * N), 1e3 * ms_graph / (R * N), N, R);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(buf));
    }
  }
  return 0;
}
