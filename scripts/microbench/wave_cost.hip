// What ONE wavefront per problem can do (diagnostic only; r04 prototype for the structured on-chip ADMM tier):
// cycles per wavefront-instruction for the building blocks of a block-tridiagonal sweep and of the row phase, when a
// SIMD holds one or two single-wavefront workgroups.  Every workgroup is one wavefront; `lds_kb` of dynamic LDS per
// workgroup limits how many share a CU (40 KB -> 4 per CU = 1 per SIMD, 20 KB -> 8 per CU = 2 per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wave_cost scripts/microbench/wave_cost.hip && /tmp/wave_cost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define FMAC_DPP(acc, w, g, k) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(g))

typedef double d2 __attribute__((ext_vector_type(2)));

// mode 0: 100 independent fma (10 chains); 1: 100 dependent fma; 2: 70 dependent fmac_dpp (10 mat-vecs of 7);
// 3: 70 fmac_dpp in two alternating accumulators; 4: LDS write -> read round trip x 10 (no barrier: one wavefront);
// 5: a sweep step x 10: 4 x ds_read_b128 (the row of G) + 1 ds_read_b64 (rhs) + fma + 7 fmac_dpp + ds_write_b64;
// 6: a row slot x 10: 7 fma (dot) + 24 dependent-ish scalar fma/min/max + 7 fma (column partials)
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += 64) lds[i] = 1.0 + 1e-6 * i;
  __syncthreads();
  double a[10];
  for (int i = 0; i < 10; i++) a[i] = 1.0 + tid * 1e-3 + i;
  double x = 1.0000001, y = 1e-9;
  double J[7], part[7];
  for (int i = 0; i < 7; i++) { J[i] = 0.5 + 1e-3 * (tid + i); part[i] = 0.0; }
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 10; j++)
#pragma unroll
        for (int i = 0; i < 10; i++) a[i] = __builtin_fma(a[i], x, y);
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 100; j++) a[0] = __builtin_fma(a[0], x, y);
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 10; j++) {
        FMAC_DPP(a[0], a[1], J[0], 0); FMAC_DPP(a[0], a[1], J[1], 1); FMAC_DPP(a[0], a[1], J[2], 2); FMAC_DPP(a[0], a[1], J[3], 3);
        FMAC_DPP(a[0], a[1], J[4], 4); FMAC_DPP(a[0], a[1], J[5], 5); FMAC_DPP(a[0], a[1], J[6], 6);
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 10; j++) {
        FMAC_DPP(a[0], a[1], J[0], 0); FMAC_DPP(a[2], a[1], J[1], 1); FMAC_DPP(a[0], a[1], J[2], 2); FMAC_DPP(a[2], a[1], J[3], 3);
        FMAC_DPP(a[0], a[1], J[4], 4); FMAC_DPP(a[2], a[1], J[5], 5); FMAC_DPP(a[0], a[1], J[6], 6);
      }
    } else if (MODE == 4) {
#pragma unroll
      for (int j = 0; j < 10; j++) {
        lds[(tid + j) & 2047] = a[0];
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        a[0] += lds[(tid * 5 + 17 * j) & 2047];
      }
    } else if (MODE == 5) {
      double v = a[0];
#pragma unroll
      for (int j = 0; j < 10; j++) {
        const d2 g0 = *(const d2 *)(lds + 8 * ((tid & 15) + 16 * j)), g1 = *(const d2 *)(lds + 8 * ((tid & 15) + 16 * j) + 2);
        const d2 g2 = *(const d2 *)(lds + 8 * ((tid & 15) + 16 * j) + 4), g3 = *(const d2 *)(lds + 8 * ((tid & 15) + 16 * j) + 6);
        const double rr = lds[1024 + 8 * j + (tid & 7)];
        double w = __builtin_fma(-g3.y, v, rr), acc = 0.0;
        FMAC_DPP(acc, w, g0.x, 0); FMAC_DPP(acc, w, g0.y, 1); FMAC_DPP(acc, w, g1.x, 2); FMAC_DPP(acc, w, g1.y, 3);
        FMAC_DPP(acc, w, g2.x, 4); FMAC_DPP(acc, w, g2.y, 5); FMAC_DPP(acc, w, g3.x, 6);
        v = acc;
        lds[1200 + 8 * j + (tid & 7)] = v;
      }
      a[0] = v;
    } else if (MODE == 6) {
#pragma unroll
      for (int j = 0; j < 10; j++) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 7; i++) s = __builtin_fma(J[i], a[i], s);
        double u = s;
#pragma unroll
        for (int i = 0; i < 8; i++) { u = __builtin_fma(u, x, y); u = fmin(u, a[8]); u = __builtin_fma(u, y, a[9]); }
#pragma unroll
        for (int i = 0; i < 7; i++) part[i] = __builtin_fma(J[i], u, part[i]);
        a[7] += u;
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double acc = 0.0;
  for (int i = 0; i < 10; i++) acc += a[i];
  for (int i = 0; i < 7; i++) acc += part[i];
  out[blockIdx.x * 64 + tid] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *what, int per_rep, int grid, int lds_kb) {
  double *out; long long *cyc;
  hipMalloc(&out, (size_t)grid * 64 * sizeof(double)); hipMalloc(&cyc, grid * sizeof(long long));
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  const int reps = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), (size_t)lds_kb * 1024, 0, out, cyc, 10);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), (size_t)lds_kb * 1024, 0, out, cyc, reps);
  hipDeviceSynchronize();
  std::vector<long long> h(grid);
  hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-64s grid %5d lds %2d KB: median %8.2f cycles per rep, %6.2f per unit (min %.2f max %.2f per rep)\n", what, grid, lds_kb,
         (double)h[grid / 2] / reps, (double)h[grid / 2] / reps / per_rep, (double)h[0] / reps, (double)h[grid - 1] / reps);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int cfg = 0; cfg < 3; cfg++) {
    const int grid = cfg == 0 ? 1024 : cfg == 1 ? 2048 : 4096, lds = cfg == 0 ? 40 : cfg == 1 ? 20 : 10;
    printf("---- %d single-wavefront workgroups per CU (s_memtime ticks at 100 MHz x 24 = shader cycles if readcyclecounter is s_memtime)\n", grid / 256);
    run<0>("100 independent v_fma_f64", 100, grid, lds);
    run<1>("100 dependent v_fma_f64", 100, grid, lds);
    run<2>("70 dependent v_fmac_f64_dpp row_newbcast (10 mat-vecs 7x7)", 70, grid, lds);
    run<3>("70 v_fmac_f64_dpp, two accumulators", 70, grid, lds);
    run<4>("10 LDS write -> wait -> read round trips", 10, grid, lds);
    run<5>("10 sweep steps (5 LDS reads, fma, 7 fmac_dpp, 1 LDS write)", 10, grid, lds);
    run<6>("10 row slots (7 fma dot, 24 update ops, 7 fma partials)", 10, grid, lds);
  }
  return 0;
}
