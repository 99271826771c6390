// N1 evidence (north_star: "MFMA used only for the dense batched Jacobian x step contraction"): the only GEMM-shaped work
// of the path is the per-block normal matrix  S_blk = J' diag(w rho) J  of the reduced KKT system (J = the R x ds Jacobian
// block of one constraint block: 100 x 12 at BASELINE configs[4], 100 x 24 with blocks on two timesteps) and the model
// merit J dx (a mat-vec).  This times S_blk on the f64 vector ALU against v_mfma_f64_16x16x4_f64, one wavefront per block,
// J resident in LDS, many blocks per launch; every block is accumulated `reps` times (S = reps x J' W J) so that the timed
// loop is arithmetic, not the load of J (diagnostic only; never part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/jtrj scripts/microbench/jtrj_mfma.hip && /tmp/jtrj
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

// VALU: lane l owns outputs (i, j) = pairs l, l + 64, ... of the ds x ds matrix; every row of J is read from LDS
template <int DS, int R>
__global__ __launch_bounds__(256) void jtrj_valu(const double *J, const double *w, double *S, int nblk, int reps) {
  __shared__ double sj[4][R * DS];
  __shared__ double sw[4][R];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int blk = blockIdx.x * 4 + wv; blk < nblk; blk += gridDim.x * 4) {
    for (int e = lane; e < R * DS; e += 64) sj[wv][e] = J[(size_t)blk * R * DS + e];
    for (int e = lane; e < R; e += 64) sw[wv][e] = w[(size_t)blk * R + e];
    constexpr int NO = (DS * DS + 63) / 64;
    double acc[NO];
#pragma unroll
    for (int o = 0; o < NO; o++) acc[o] = 0.0;
    __builtin_amdgcn_wave_barrier();
    for (int rep = 0; rep < reps; rep++) {
      for (int r = 0; r < R; r++) {
        const double wr = sw[wv][r];
#pragma unroll
        for (int o = 0; o < NO; o++) {
          const int e = lane + 64 * o, i = (e < DS * DS ? e : 0) / DS, j = (e < DS * DS ? e : 0) % DS;
          acc[o] += (wr * sj[wv][r * DS + i]) * sj[wv][r * DS + j];
        }
      }
    }
#pragma unroll
    for (int o = 0; o < NO; o++) { const int e = lane + 64 * o; if (e < DS * DS) S[(size_t)blk * DS * DS + e] = acc[o]; }
  }
}

// MFMA: D (16 x 16) += A (16 x 4) B (4 x 16) with A[i][k] = w_r J[r][i], B[k][j] = J[r][j], r = 4 step + k; lane l supplies
// A[l % 16][l / 16] and B[l / 16][l % 16] and holds D[4 v + l / 16][l % 16], v < 4 (gfx950 layout of the 16x16x4 f64 MFMA,
// probed with scripts/microbench/mfma_f64_16x16_layout.hip)
template <int DS, int R>
__global__ __launch_bounds__(256) void jtrj_mfma(const double *J, const double *w, double *S, int nblk, int reps) {
  __shared__ double sj[4][R * DS];
  __shared__ double sw[4][R];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NT = (DS + 15) / 16;
  for (int blk = blockIdx.x * 4 + wv; blk < nblk; blk += gridDim.x * 4) {
    for (int e = lane; e < R * DS; e += 64) sj[wv][e] = J[(size_t)blk * R * DS + e];
    for (int e = lane; e < R; e += 64) sw[wv][e] = w[(size_t)blk * R + e];
    d4 acc[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; a++)
#pragma unroll
      for (int b = 0; b < NT; b++) acc[a][b] = d4{0, 0, 0, 0};
    __builtin_amdgcn_wave_barrier();
    for (int rep = 0; rep < reps; rep++) {
      for (int st = 0; st < R / 4; st++) {
        const int r = 4 * st + (lane >> 4);
        const double wr = sw[wv][r];
        double av[NT], bv[NT];
#pragma unroll
        for (int a = 0; a < NT; a++) {
          const int i = 16 * a + (lane & 15);
          const double v = i < DS ? sj[wv][r * DS + i] : 0.0;
          av[a] = wr * v; bv[a] = v;
        }
#pragma unroll
        for (int a = 0; a < NT; a++)
#pragma unroll
          for (int b = 0; b < NT; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
      }
    }
#pragma unroll
    for (int a = 0; a < NT; a++)
#pragma unroll
      for (int b = 0; b < NT; b++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const int i = 16 * a + 4 * v + (lane >> 4), j = 16 * b + (lane & 15);
          if (i < DS && j < DS) S[(size_t)blk * DS * DS + i * DS + j] = acc[a][b][v];
        }
  }
}

template <int DS, int R>
static void run(int nblk, int reps) {
  std::vector<double> hJ((size_t)nblk * R * DS), hw((size_t)nblk * R);
  for (size_t i = 0; i < hJ.size(); i++) hJ[i] = std::sin(0.37 * (double)i) + 0.1;
  for (size_t i = 0; i < hw.size(); i++) hw[i] = 0.1 + 0.01 * (double)(i % 7);
  double *J, *w, *S1, *S2;
  hipMalloc(&J, hJ.size() * 8); hipMalloc(&w, hw.size() * 8); hipMalloc(&S1, (size_t)nblk * DS * DS * 8); hipMalloc(&S2, (size_t)nblk * DS * DS * 8);
  hipMemcpy(J, hJ.data(), hJ.size() * 8, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[2];
  const int grid = 256 * 2;
  for (int which = 0; which < 2; which++) {
    for (int pass = 0; pass < 2; pass++) {
      hipEventRecord(e0);
      if (which == 0) jtrj_valu<DS, R><<<grid, 256>>>(J, w, S1, nblk, reps);
      else jtrj_mfma<DS, R><<<grid, 256>>>(J, w, S2, nblk, reps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[which], e0, e1);
    }
  }
  std::vector<double> a((size_t)nblk * DS * DS), b(a.size());
  hipMemcpy(a.data(), S1, a.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), S2, b.size() * 8, hipMemcpyDeviceToHost);
  double err = 0, nrm = 0;
  for (size_t i = 0; i < a.size(); i++) { err = std::fmax(err, std::fabs(a[i] - b[i])); nrm = std::fmax(nrm, std::fabs(a[i])); }
  const double flop = 2.0 * (double)nblk * reps * R * DS * DS;
  printf("J %3d x %2d, %6d blocks x %d repetitions: VALU %8.3f ms = %6.2f TFLOP/s   MFMA 16x16x4 %8.3f ms = %6.2f TFLOP/s (useful flops; padded tile %d x %d)   max |diff| %.2e of %.2e\n",
         R, DS, nblk, reps, ms[0], flop / ms[0] * 1e-9, ms[1], flop / ms[1] * 1e-9, 16 * ((DS + 15) / 16), 16 * ((DS + 15) / 16), err, nrm);
  hipFree(J); hipFree(w); hipFree(S1); hipFree(S2);
}

int main() {
  run<12, 100>(51200, 20);      // 12-DOF x 50: 50 blocks x 1024 problems
  run<24, 100>(50176, 20);      // the same rows on blocks of two timesteps
  run<16, 100>(51200, 20);      // a full 16 x 16 tile
  run<7, 12>(20480, 200);       // 7-DOF x 20 shaped blocks (10 rows padded to 12)
  return 0;
}
