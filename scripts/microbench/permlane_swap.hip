#include <hip/hip_runtime.h>
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *out, const unsigned *in) {
  unsigned a = in[threadIdx.x], b = in[64 + threadIdx.x];
  uint2v r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  uint2v s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r.x; out[64 + threadIdx.x] = r.y; out[128 + threadIdx.x] = s.x; out[192 + threadIdx.x] = s.y;
}
int main() {
  unsigned h[128], o[256], *di, *dout;
  for (int i = 0; i < 64; i++) { h[i] = 100 + i; h[64 + i] = 200 + i; }
  hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
  hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dout, di);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  const char *nm[4] = {"p32.x", "p32.y", "p16.x", "p16.y"};
  for (int r = 0; r < 4; r++) { printf("%s:", nm[r]); for (int i = 0; i < 64; i++) printf(" %u", o[r * 64 + i]); printf("\n"); }
  return 0;
}
