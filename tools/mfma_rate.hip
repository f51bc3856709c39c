// Matrix-pipe rate probe for gfx950: fp32 MFMA (32x32x2) against the "3 x bf16" emulation of an fp32 product
// (6 bf16 MFMAs 32x32x16 per 16 k-values: hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid).  Register operands only:
// this measures what the matrix pipe can do, not a convolution.  Build and run:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
  bf16x8 ah, bh;
#pragma unroll
  for (int r = 0; r < 8; r++) ah[r] = (__bf16)(a + r), bh[r] = (__bf16)(b - r);
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
      // 16 k-values in fp32: 8 MFMAs of k = 2 per accumulator; 8 accumulators -> 64 MFMAs
#pragma unroll
      for (int k = 0; k < 8; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    } else {
      // the same 16 k-values as 6 bf16 products per accumulator -> 48 MFMAs
#pragma unroll
      for (int k = 0; k < 6; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int blocks = 256 * 2;   // 2 workgroups of 4 waves per CU: 2 waves per SIMD
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // fp32-equivalent FLOPs: per iteration and accumulator 32 x 32 x 16 MACs
  const double flops = 2.0 * 32 * 32 * 16 * 8 * (double)iters * blocks * 4;
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 2 * 256 * sizeof(float));
  const double t0 = run<0>(out, 4000), t1 = run<1>(out, 4000);
  printf("fp32 MFMA 32x32x2      : %.1f TFLOP/s\n", t0);
  printf("3 x bf16 (6 x 32x32x16): %.1f TFLOP/s fp32-equivalent (%.2fx)\n", t1, t1 / t0);
  return 0;
}
