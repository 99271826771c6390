// Per-CU L2 streaming micro-benchmark: one workgroup re-reads (and optionally re-writes) an
// L2-resident array; reports bytes per shader clock.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NT, int UNROLL, bool WRITE>
__global__ __launch_bounds__(NT) void stream(const double *in, double *out, int n, int reps, long long *cyc, double *sink) {
  const int tid = threadIdx.x;
  double acc = 0.0;
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
    for (int base = 0; base < n; base += NT * UNROLL) {
      double v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) { int i = base + u * NT + tid; v[u] = in[i < n ? i : 0]; }
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        acc += v[u];
        if (WRITE) { int i = base + u * NT + tid; if (i < n) out[i] = v[u] * 1.0000001; }
      }
    }
    __syncthreads();
  }
  long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * NT + tid] = acc;
}
template <int NT, int UNROLL, bool WRITE>
void run(const char *name, int n, int blocks) {
  double *in, *out, *sink; long long *cyc;
  hipMalloc(&in, (size_t)n * 8 * blocks); hipMalloc(&out, (size_t)n * 8 * blocks); hipMalloc(&sink, blocks * NT * 8); hipMalloc(&cyc, blocks * 8);
  hipMemset(in, 0, (size_t)n * 8 * blocks); hipMemset(out, 0, (size_t)n * 8 * blocks);
  const int reps = 200;
  stream<NT, UNROLL, WRITE><<<1, NT>>>(in, out, n, 2, cyc, sink);
  hipDeviceSynchronize();
  stream<NT, UNROLL, WRITE><<<blocks, NT>>>(in, out, n, reps, cyc, sink);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double bytes = (double)n * 8 * reps * (WRITE ? 2 : 1);
  printf("%-28s n=%d KB=%d blocks=%d threads=%d unroll=%d : %.1f B/clk per CU (cycles/rep %.0f)\n", name, n, n * 8 / 1024, blocks, NT, UNROLL,
         bytes / (double)h[0], (double)h[0] / reps);
  hipFree(in); hipFree(out); hipFree(sink); hipFree(cyc);
}
int main() {
  const int n = 70624;   // nnz(A) of the 12x50 QP
  run<512, 8, false>("read", n, 1);
  run<512, 16, false>("read", n, 1);
  run<512, 32, false>("read", n, 1);
  run<1024, 8, false>("read", n, 1);
  run<1024, 16, false>("read", n, 1);
  run<256, 32, false>("read", n, 1);
  run<512, 16, true>("read+write", n, 1);
  run<1024, 16, true>("read+write", n, 1);
  run<512, 16, false>("read small (L1?)", 2048, 1);
  run<512, 16, false>("read 4MB", 524288, 1);
  return 0;
}
