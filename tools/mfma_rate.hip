// Matrix-pipe rate probe for gfx950 (register operands only: what the matrix pipe sustains, not a convolution):
//   mode 0  fp32 MFMA 32x32x2 (8 per 16 k-values)
//   mode 1  "3 x bf16": 6 bf16 MFMAs 32x32x16 per 16 k-values (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid)
//   mode 2  "2 x f16":  3 f16 MFMAs 32x32x16 per 16 k-values (hi*hi, hi*lo, lo*hi) -- the scheme of csrc/sr3d_hconv.hip
//   mode 3  the same products as 16x16x32 MFMAs (4 accumulator registers per tile, K = 32)
//   mode 4  the SAME NUMBER of MFMA instructions as mode 3, but the CDNA3-era v_mfma_f32_16x16x16_f16 (K = 16): if it issued
//           at twice the cadence it would be the natural instruction for the odd 27th tap of sr3d_hconv.hip (27 taps = 13
//           K = 32 pairs + one K = 16 single instead of a zero-padded 14th pair); it does not -- see the printed times
// Each mode runs for tens of milliseconds with lane-dependent operands (power management reacts within milliseconds;
// a 2 ms burst with constant operands overstates the sustained rate) and reports the shader clock it saw
// (s_memtime cycles per s_memrealtime tick of 100 MHz).  Build and run:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, unsigned long long* clk, int iters, float seed) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  f32x4 acc4[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  // operand sets that differ per lane and per use (pseudo-random mantissas)
  float af[4], bf[4];
  bf16x8 ab[4], bb[4];
  h8 ah[4], bh[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    unsigned s = (threadIdx.x * 2654435761u) ^ (q * 40503u + 12345u);
    af[q] = seed * (1.f + (float)(s & 0xffff) / 65536.f), bf[q] = seed * (1.f + (float)((s >> 16) & 0xffff) / 65536.f);
#pragma unroll
    for (int r = 0; r < 8; r++) {
      s = s * 1664525u + 1013904223u;
      const float v = seed * ((float)(s >> 8) / 16777216.f - 0.5f);
      ab[q][r] = (__bf16)v, bb[q][r] = (__bf16)(v * 0.7f), ah[q][r] = (_Float16)v, bh[q][r] = (_Float16)(v * 0.7f);
    }
  }
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[(k + i) & 3], bf[(k ^ i) & 3], acc[i], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 6; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[(k + i) & 3], bb[(k ^ i) & 3], acc[i], 0, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(k + i) & 3], bh[(k ^ i) & 3], acc[i], 0, 0, 0);
    } else if (MODE == 4) {
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const h4 a4 = {ah[(k + i) & 3][0], ah[(k + i) & 3][1], ah[(k + i) & 3][2], ah[(k + i) & 3][3]};
          const h4 b4 = {bh[(k ^ i) & 3][0], bh[(k ^ i) & 3][1], bh[(k ^ i) & 3][2], bh[(k ^ i) & 3][3]};
          acc4[(2 * i + k) & 15] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc4[(2 * i + k) & 15], 0, 0, 0);
        }
    } else {
      // the same FLOPs per iteration: 4 accumulators x 32x32x16 = 16 tiles of 16x16; per tile 3 products x (K = 16 -> half
      // an MFMA of K = 32): 24 MFMAs 16x16x32 per iteration
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc4[(2 * i + k) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[(k + i) & 3], bh[(k ^ i) & 3], acc4[(2 * i + k) & 15], 0, 0, 0);
    }
  }
  if (MODE >= 3) {
#pragma unroll
    for (int i = 0; i < 16; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[i >> 2][(i & 3) * 4 + r] = acc4[i][r];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[2 * blockIdx.x] = clock64() - c0, clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
}

template <int MODE>
double run(float* out, unsigned long long* clk, int iters, double* mhz, double* ms_out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int blocks = 256 * 2;   // 2 workgroups of 4 waves per CU: 2 waves per SIMD
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, clk, 10, 1e-3f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1e-3f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  *mhz = (double)h[0] / ((double)h[1] / 100.0);   // cycles per microsecond
  *ms_out = ms;
  // fp32-equivalent FLOPs: per iteration and accumulator 32 x 32 x 16 MACs
  const double flops = 2.0 * 32 * 32 * 16 * 4 * (double)iters * blocks * 4;
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, 256 * 2 * 256 * sizeof(float));
  hipMalloc(&clk, 512 * 2 * sizeof(unsigned long long));
  double mhz[5], ms[5];
  const double t0 = run<0>(out, clk, 40000, &mhz[0], &ms[0]);
  const double t1 = run<1>(out, clk, 100000, &mhz[1], &ms[1]);
  const double t2 = run<2>(out, clk, 200000, &mhz[2], &ms[2]);
  const double t3 = run<3>(out, clk, 200000, &mhz[3], &ms[3]);
  printf("fp32 MFMA 32x32x2      : %7.1f TFLOP/s                         %6.1f ms at %4.0f MHz\n", t0, ms[0], mhz[0]);
  printf("3 x bf16 (6 x 32x32x16): %7.1f TFLOP/s fp32-equivalent (%.2fx)  %6.1f ms at %4.0f MHz\n", t1, t1 / t0, ms[1], mhz[1]);
  printf("2 x f16  (3 x 32x32x16): %7.1f TFLOP/s fp32-equivalent (%.2fx)  %6.1f ms at %4.0f MHz\n", t2, t2 / t0, ms[2], mhz[2]);
  printf("2 x f16  (as 16x16x32) : %7.1f TFLOP/s fp32-equivalent (%.2fx)  %6.1f ms at %4.0f MHz\n", t3, t3 / t0, ms[3], mhz[3]);
  run<4>(out, clk, 200000, &mhz[4], &ms[4]);
  printf("same count of 16x16x16 : %6.1f ms at %4.0f MHz for as many instructions as the line above (half the FLOPs each): %.2f of its time\n",
         ms[4], mhz[4], ms[4] / ms[3]);
  return 0;
}
