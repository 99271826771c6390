// Which lane holds which element of v_mfma_f64_4x4x4 (4 blocks of 16 lanes)?  Prints the layout hypothesis that
// reproduces D = A B per block.   hipcc --offload-arch=gfx950 -O2 mfma_f64_4x4_layout.hip -o /tmp/mfma_layout && /tmp/mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *a, const double *b, double *d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main() {
  double ha[64], hb[64], hd[64], *da, *db, *dd;
  for (int l = 0; l < 64; l++) { ha[l] = 1.0 + 0.37 * l + 0.01 * l * l; hb[l] = 2.0 - 0.11 * l + 0.003 * l * l * l; }
  (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 512);
  (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
  (void)hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
  // every operand: the three 2-bit fields of the lane number (bits 1:0, 3:2, 5:4) are (block, row, column) in some order
  const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  const char *names[3] = {"l&3", "(l>>2)&3", "l>>4"};
  auto fld = [](int l, int f) { return (l >> (2 * f)) & 3; };
  for (int pa = 0; pa < 6; pa++) for (int pb = 0; pb < 6; pb++) for (int pd = 0; pd < 6; pd++) {
    double A[4][4][4], B[4][4][4], err = 0;
    for (int l = 0; l < 64; l++) {
      A[fld(l, perm[pa][0])][fld(l, perm[pa][1])][fld(l, perm[pa][2])] = ha[l];     // [block][i][k]
      B[fld(l, perm[pb][0])][fld(l, perm[pb][1])][fld(l, perm[pb][2])] = hb[l];     // [block][k][j]
    }
    for (int l = 0; l < 64; l++) {
      const int blk = fld(l, perm[pd][0]), i = fld(l, perm[pd][1]), j = fld(l, perm[pd][2]);
      double s = 0; for (int kk = 0; kk < 4; kk++) s += A[blk][i][kk] * B[blk][kk][j];
      err = fmax(err, fabs(s - hd[l]));
    }
    if (err < 1e-6)
      printf("A: block=%s i=%s k=%s | B: block=%s k=%s j=%s | D: block=%s i=%s j=%s\n", names[perm[pa][0]], names[perm[pa][1]],
             names[perm[pa][2]], names[perm[pb][0]], names[perm[pb][1]], names[perm[pb][2]], names[perm[pd][0]], names[perm[pd][1]], names[perm[pd][2]]);
  }
  return 0;
}
