// Which lane holds which element of v_mfma_f64_16x16x4_f64?  Probes with unit operands (diagnostic only).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/l16 scripts/microbench/mfma_f64_16x16_layout.hip && /tmp/l16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *a, const double *b, double *d) {
  const int l = threadIdx.x;
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
  for (int v = 0; v < 4; v++) d[4 * l + v] = c[v];
}
int main() {
  double ha[64], hb[64], hd[256], *da, *db, *dd;
  (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 2048);
  // A = unit at lane la, B = lane number + 1 in every lane: D = B[k(la)][j] in row i(la)
  for (int la : {0, 1, 15, 16, 17, 35, 63}) {
    for (int l = 0; l < 64; l++) { ha[l] = l == la ? 1.0 : 0.0; hb[l] = l + 1; }
    (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(hd, dd, 2048, hipMemcpyDeviceToHost);
    printf("A unit at lane %2d -> nonzero D at (lane, reg) = value (the B lane that supplied it + 1):", la);
    int cnt = 0;
    for (int e = 0; e < 256; e++) if (hd[e] != 0.0 && cnt++ < 6) printf(" (%d,%d)=%g", e / 4, e % 4, hd[e]);
    printf("  [%d nonzeros]\n", cnt);
  }
  return 0;
}
