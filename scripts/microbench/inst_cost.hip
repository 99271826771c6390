// Instruction-cost micro-benchmark for the row-local ADMM kernel's building blocks (diagnostic only): what one
// wavefront-instruction of each kind costs when 1 or 2 wavefronts share a SIMD, and what a barrier hand-over through LDS
// costs.  One workgroup on one CU; cycles from s_memtime around REPS repetitions of an unrolled body.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/inst_cost scripts/microbench/inst_cost.hip && /tmp/inst_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REPS 2000
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double fold32(double a, double b) {
  const u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double fold16(double a, double b) {
  const u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double quad1(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return v + __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xb1, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo, 0xb1, 0xf, 0xf, true));
}

// mode: 0 independent fma (10 chains x 10), 1 dependent fma chain (100), 2 fold32 x 20 (dependent pairs), 3 fold16 x 20,
//       4 quad step x 20 (dependent), 5 barrier x 10, 6 LDS write -> barrier -> read (b64) x 10, 7 ds_read_b128 x 16 + wait,
//       8 the r02 5-row reduction (fold32 x3, fold16 x2, quad x4), 9 dependent ds_read_b64 chain (pointer chase) x 20,
//       10 the same reduction by v_mfma_f64_4x4x4, 11 / 12 a whole W phase (45 fma + reduction) with either reduction,
//       13 v_readlane_b32 x 16 (reload of spilled SGPRs), 14 v_mov_b32 x 16
template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, long long *cyc, int reps) {
  __shared__ __attribute__((aligned(16))) double lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = (double)((i * 7 + 3) % 4096) * 8.0;   // also a pointer-chase table (byte offsets)
  __syncthreads();
  double a[10];
  for (int i = 0; i < 10; i++) a[i] = 1.0 + tid * 1e-3 + i;
  double x = 1.0000001, y = 1e-9;
  unsigned int p = (tid * 8) & 32767;
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
      for (int j = 0; j < 20; j++) a[0] = fold32(a[0], a[1]);
    } else if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 20; j++) a[0] = fold16(a[0], a[1]);
    } else if (MODE == 4) {
#pragma unroll
      for (int j = 0; j < 20; j++) a[0] = quad1(a[0]);
    } else if (MODE == 5) {
#pragma unroll
      for (int j = 0; j < 10; j++) __syncthreads();
    } else if (MODE == 6) {
#pragma unroll
      for (int j = 0; j < 10; j++) {
        lds[(tid + j) & 4095] = a[0];
        __syncthreads();
        a[0] += lds[(tid * 5 + 17 * j) & 4095];
        __syncthreads();
      }
    } else if (MODE == 7) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = *(const d2 *)(lds + ((2 * tid + 128 * j) & 4094));
#pragma unroll
      for (int j = 0; j < 16; j++) a[j % 10] += v[j].x + v[j].y;
    } else if (MODE == 8) {
      const double u0 = fold32(a[0], a[1]), u1 = fold32(a[2], a[3]), u2_ = fold32(a[4], 0.0);
      double t0_ = fold16(u0, u1), t1_ = fold16(u2_, 0.0);
      t0_ = quad1(t0_); t1_ = quad1(t1_);
      const int lo = __double2loint(t0_), hi = __double2hiint(t0_), lo1 = __double2loint(t1_), hi1 = __double2hiint(t1_);
      t0_ += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x4e, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo, 0x4e, 0xf, 0xf, true));
      t1_ += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi1, 0x4e, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo1, 0x4e, 0xf, 0xf, true));
      a[0] = t0_; a[1] = t1_; a[2] += t0_; a[3] += t1_; a[4] += t0_;
    } else if (MODE == 10) {
      // the same 5-row reduction on the matrix pipe: v_mfma_f64_4x4x4 with B = 1 sums over lane >> 4 and hands the result
      // to the lanes with lane >> 4 = (source lane & 3); applied twice it sums all 16 lanes of a block ((lane >> 2) & 3).
      // Second application: each lane passes on the value whose index is its own lane & 3 -> 4 rows at once.
      double d1[5];
#pragma unroll
      for (int v = 0; v < 5; v++) d1[v] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[v], 1.0, 0.0, 0, 0, 0);
      const int pq = tid & 3;
      const double sel = pq == 0 ? d1[0] : pq == 1 ? d1[1] : pq == 2 ? d1[2] : d1[3];
      const double t0_ = __builtin_amdgcn_mfma_f64_4x4x4f64(sel, 1.0, 0.0, 0, 0, 0);
      const double t1_ = __builtin_amdgcn_mfma_f64_4x4x4f64(d1[4], 1.0, 0.0, 0, 0, 0);
      a[0] = t0_; a[1] = t1_; a[2] += t0_; a[3] += t1_; a[4] += t0_;
    } else if (MODE == 11) {
      // 45 independent multiply-adds (5 chains x 9) + the swap reduction: one W phase without its LDS reads
      double acc[5] = {0, 0, 0, 0, 0};
#pragma unroll
      for (int cc = 0; cc < 9; cc++)
#pragma unroll
        for (int rr = 0; rr < 5; rr++) acc[rr] = __builtin_fma(a[(rr + cc) % 10], x + cc, acc[rr]);
      const double u0 = fold32(acc[0], acc[1]), u1 = fold32(acc[2], acc[3]), u2_ = fold32(acc[4], 0.0);
      double t0_ = fold16(u0, u1), t1_ = fold16(u2_, 0.0);
      t0_ = quad1(t0_); t1_ = quad1(t1_);
      const int lo = __double2loint(t0_), hi = __double2hiint(t0_), lo1 = __double2loint(t1_), hi1 = __double2hiint(t1_);
      t0_ += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x4e, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo, 0x4e, 0xf, 0xf, true));
      t1_ += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi1, 0x4e, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo1, 0x4e, 0xf, 0xf, true));
      a[0] = t0_ * 1e-3 + 1.0; a[1] = t1_ * 1e-3 + 1.0;
    } else if (MODE == 12) {
      // the same W phase with the reduction on the matrix pipe
      double acc[5] = {0, 0, 0, 0, 0};
#pragma unroll
      for (int cc = 0; cc < 9; cc++)
#pragma unroll
        for (int rr = 0; rr < 5; rr++) acc[rr] = __builtin_fma(a[(rr + cc) % 10], x + cc, acc[rr]);
      double d1[5];
#pragma unroll
      for (int v = 0; v < 5; v++) d1[v] = __builtin_amdgcn_mfma_f64_4x4x4f64(acc[v], 1.0, 0.0, 0, 0, 0);
      const int pq = tid & 3;
      const double sel = pq == 0 ? d1[0] : pq == 1 ? d1[1] : pq == 2 ? d1[2] : d1[3];
      const double t0_ = __builtin_amdgcn_mfma_f64_4x4x4f64(sel, 1.0, 0.0, 0, 0, 0);
      const double t1_ = __builtin_amdgcn_mfma_f64_4x4x4f64(d1[4], 1.0, 0.0, 0, 0, 0);
      a[0] = t0_ * 1e-3 + 1.0; a[1] = t1_ * 1e-3 + 1.0;
    } else if (MODE == 13) {
      // 16 v_readlane_b32 into 16 different SGPRs, then one use of each (what a reload of spilled SGPRs costs)
      int s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15;
      const int src = __double2loint(a[0]);
      asm volatile("v_readlane_b32 %0, %16, 0\n\tv_readlane_b32 %1, %16, 1\n\tv_readlane_b32 %2, %16, 2\n\tv_readlane_b32 %3, %16, 3\n\t"
                   "v_readlane_b32 %4, %16, 4\n\tv_readlane_b32 %5, %16, 5\n\tv_readlane_b32 %6, %16, 6\n\tv_readlane_b32 %7, %16, 7\n\t"
                   "v_readlane_b32 %8, %16, 8\n\tv_readlane_b32 %9, %16, 9\n\tv_readlane_b32 %10, %16, 10\n\tv_readlane_b32 %11, %16, 11\n\t"
                   "v_readlane_b32 %12, %16, 12\n\tv_readlane_b32 %13, %16, 13\n\tv_readlane_b32 %14, %16, 14\n\tv_readlane_b32 %15, %16, 15\n\ts_nop 1"
                   : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7), "=s"(s8), "=s"(s9), "=s"(s10),
                     "=s"(s11), "=s"(s12), "=s"(s13), "=s"(s14), "=s"(s15) : "v"(src));
      p += (unsigned)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + s8 + s9 + s10 + s11 + s12 + s13 + s14 + s15) & 8u;
    } else if (MODE == 14) {
      // 16 v_mov_b32 VGPR -> VGPR (the same count of plain 32-bit VALU operations, for comparison)
      int v0 = (int)p, v1, v2, v3;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4" : "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v0) : "v"(v0));
        p += (unsigned)(v1 + v2 + v3) & 8u;
      }
    } else if (MODE == 9) {
#pragma unroll
      for (int j = 0; j < 20; j++) p = (unsigned int)*(const double *)((const char *)lds + p);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 10; i++) s += a[i];
  out[blockIdx.x * blockDim.x + tid] = s + p;
  if ((tid & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int per_rep, int threads) {
  double *out; long long *cyc;
  hipMalloc(&out, 512 * 8); hipMalloc(&cyc, 8 * 8);
  k<MODE><<<1, threads>>>(out, cyc, 10); hipDeviceSynchronize();
  k<MODE><<<1, threads>>>(out, cyc, REPS); hipDeviceSynchronize();
  std::vector<long long> h(8); hipMemcpy(h.data(), cyc, 64, hipMemcpyDeviceToHost);
  long long mx = 0; for (int w = 0; w < threads / 64; w++) mx = std::max(mx, h[w]);
  printf("%-44s %4d threads: %8.1f cycles per repetition = %6.2f per item (%d items)\n", name, threads, (double)mx / REPS, (double)mx / REPS / per_rep, per_rep);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int threads : {64, 256, 512}) {
    run<0>("v_fma_f64 independent", 100, threads);
    run<1>("v_fma_f64 dependent chain", 100, threads);
    run<2>("fold32 (2 permlane32_swap + add), dependent", 20, threads);
    run<3>("fold16 (2 permlane16_swap + add), dependent", 20, threads);
    run<4>("quad step (2 mov_dpp + add), dependent", 20, threads);
    run<5>("s_barrier", 10, threads);
    run<6>("LDS write -> barrier -> read -> barrier", 10, threads);
    run<7>("16 ds_read_b128 + 16 adds", 16, threads);
    run<8>("5-row reduction of the W phase", 1, threads);
    run<9>("dependent ds_read_b64 (pointer chase)", 20, threads);
    run<10>("5-row reduction by 7 v_mfma_f64_4x4x4", 1, threads);
    run<11>("W phase: 45 fma + swap reduction", 1, threads);
    run<12>("W phase: 45 fma + mfma reduction", 1, threads);
    run<13>("16 v_readlane_b32 to 16 SGPRs", 16, threads);
    run<14>("16 v_mov_b32 (VGPR to VGPR)", 16, threads);
  }
  return 0;
}
