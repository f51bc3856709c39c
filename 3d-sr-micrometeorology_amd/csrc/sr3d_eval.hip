// All evaluation metrics of the reference's final test pass in ONE streaming kernel (gfx950).
//
// The reference evaluates ten nn.Modules one after the other on every test sample (script/train_model.py:366-390,
// src/loss_maker.py:522-741): each re-reads prediction, target and mask, and the divergence / vorticity ones run
// 27-tap depthwise convolutions for 2-tap central differences.  Here one pass over (p, t, b) -- 36 B per voxel --
// accumulates every sum those modules need; a one-block kernel turns them into the metric values.
//
//   d = p - t (4 channels: T, u, v, w);  near = calc_mask_near_build_wall(b) (loss_maker.py:57-83)
//   interior = voxels 1 .. n-2 in z, y, x;  M = b * (1 - near) there (loss_maker.py:84-113, 133-160)
//   vel(q) = stds[1:] * q[1:];  div(q) = d_x vel_u + d_y vel_v + d_z vel_w,  central differences with step delta
//   omega(q) = (d_y w - d_z v, d_z u - d_x w, d_x v - d_y u)                (loss_maker.py:163-191)
//
// Reductions are two-stage with a fixed order (per-thread -> wave (DPP) -> block -> ordered final sum in double):
// bit-reproducible run to run.
#include "sr3d_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;
constexpr int kAcc = 20;

enum {
  A_ABS = 0,     // sum |d|, 4 channels
  A_B_ABS,       // sum b * |d|
  A_B,           // sum b
  A_NEAR_ABS,    // sum near * |d|
  A_NEAR,        // sum near
  A_RES_P,       // sum_int M |div(p)|
  A_RES_T,       // sum_int M |div(t)|
  A_B_INT,       // sum_int b
  A_NEAR_INT,    // sum_int near
  A_T,           // sum b |d_T| * std_T
  A_V,           // sum b ||std_v * d_v||
  A_T_LEV,       // the same two restricted to z == lev
  A_V_LEV,
  A_B_LEV,       // sum_{z == lev} b
  A_DDIV,        // sum_int |M div(p) - M div(t)|
  A_OMEGA,       // sum_int M ||omega(p) - omega(t)||
  A_SQ,          // sum d^2
  A_B_SQ,        // sum b d^2
  A_NEAR_SQ,     // sum near d^2
  A_UNUSED
};

struct EvalParams {
  const float* p;
  const float* t;
  const float* b;
  int B, Z, Y, X, lev;
  float s[4];
  float w;        // 1 / (2 delta)
  float* part;    // [blocks][kAcc]
  float* out;     // SR3D_EVAL_COUNT values
  long long vox;  // B * Z * Y * X
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// central difference exactly as the reference's depthwise convolution evaluates it: hi * w + lo * (-w)
// (contraction off: hipcc would otherwise fuse one product into an fma; the convolution rounds both products)
__device__ __forceinline__ float cdiff(float hi, float lo, float w) {
#pragma clang fp contract(off)
  const float a = hi * w, b = lo * -w;
  return a + b;
}

__global__ __launch_bounds__(kThreads) void eval_kernel(const EvalParams q) {
#pragma clang fp contract(off)
  float acc[kAcc];
#pragma unroll
  for (int i = 0; i < kAcc; i++) acc[i] = 0.f;
  const long long zyx = (long long)q.Z * q.Y * q.X;
  const long long sy = q.X, sz = (long long)q.Y * q.X;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < q.vox; i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int x = (int)(r % q.X);
    r /= q.X;
    const int y = (int)(r % q.Y);
    r /= q.Y;
    const int z = (int)(r % q.Z);
    const int bi = (int)(r / q.Z);
    const long long sp = i - (long long)bi * zyx;          // voxel inside the sample
    const float* pb = q.p + (long long)bi * 4 * zyx + sp;  // channel 0 of this voxel
    const float* tb = q.t + (long long)bi * 4 * zyx + sp;
    const float* mb = q.b + (long long)bi * zyx;
    const float bv = mb[sp];

    // near-wall flag: any building voxel (1 - b > 0) in the zero-padded 3x3x3 box, on a fluid voxel
    float box = 0.f;
#pragma unroll
    for (int dz = -1; dz <= 1; dz++)
#pragma unroll
      for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
          const int zz = z + dz, yy = y + dy, xx = x + dx;
          if ((unsigned)zz < (unsigned)q.Z && (unsigned)yy < (unsigned)q.Y && (unsigned)xx < (unsigned)q.X)
            box += 1.f - mb[sp + dz * sz + dy * sy + dx];
        }
    const float near = ((box > 0.f ? 1.f : 0.f) * bv > 0.f) ? 1.f : 0.f;

    float pc[4], tc[4], sa = 0.f, sq = 0.f;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      pc[c] = pb[c * zyx], tc[c] = tb[c * zyx];
      const float d = pc[c] - tc[c];
      sa += fabsf(d), sq += d * d;
    }
    acc[A_ABS] += sa, acc[A_SQ] += sq;
    acc[A_B] += bv, acc[A_B_ABS] += bv * sa, acc[A_B_SQ] += bv * sq;
    acc[A_NEAR] += near, acc[A_NEAR_ABS] += near * sa, acc[A_NEAR_SQ] += near * sq;

    // |T_p - T_t| * std_T and || std_v * v_p - std_v * v_t ||  (loss_maker.py:614-671)
    const float dT = fabsf(pc[0] - tc[0]) * q.s[0];
    float n2 = 0.f;
#pragma unroll
    for (int c = 1; c < 4; c++) {
      const float vp = pc[c] * q.s[c], vt = tc[c] * q.s[c];   // (contraction is off in this kernel)
      const float dv = vp - vt;
      n2 += dv * dv;
    }
    const float dV = sqrtf(n2);
    acc[A_T] += bv * dT, acc[A_V] += bv * dV;
    if (z == q.lev) acc[A_T_LEV] += bv * dT, acc[A_V_LEV] += bv * dV, acc[A_B_LEV] += bv;

    if (z >= 1 && z < q.Z - 1 && y >= 1 && y < q.Y - 1 && x >= 1 && x < q.X - 1) {
      acc[A_B_INT] += bv, acc[A_NEAR_INT] += near;
      const float M = bv * (1.f - near);
      float res[2], om[2][3];
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const float* f = k == 0 ? pb : tb;
        // g[c][axis]: derivative of velocity component c (0 = u, 1 = v, 2 = w) along axis (0 = x, 1 = y, 2 = z)
        float g[3][3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float* fc = f + (long long)(c + 1) * zyx;
          const float sc = q.s[c + 1];
          g[c][0] = cdiff(sc * fc[1], sc * fc[-1], q.w);
          g[c][1] = cdiff(sc * fc[sy], sc * fc[-sy], q.w);
          g[c][2] = cdiff(sc * fc[sz], sc * fc[-sz], q.w);
        }
        res[k] = (g[0][0] + g[1][1]) + g[2][2];
        om[k][0] = g[2][1] - g[1][2];   // dw/dy - dv/dz
        om[k][1] = g[0][2] - g[2][0];   // du/dz - dw/dx
        om[k][2] = g[1][0] - g[0][1];   // dv/dx - du/dy
      }
      const float rp = res[0] * M, rt = res[1] * M;
      acc[A_RES_P] += fabsf(rp), acc[A_RES_T] += fabsf(rt), acc[A_DDIV] += fabsf(rp - rt);
      float o2 = 0.f;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float dd = om[0][a] * M - om[1][a] * M;
        o2 += dd * dd;
      }
      acc[A_OMEGA] += sqrtf(o2);
    }
  }
  __shared__ float sm[kThreads / 64][kAcc];
#pragma unroll
  for (int i = 0; i < kAcc; i++) {
    const float v = wave_sum(acc[i]);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < kAcc) {
    float v = 0.f;
    for (int w = 0; w < kThreads / 64; w++) v += sm[w][threadIdx.x];
    q.part[(long long)blockIdx.x * kAcc + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(64) void eval_final_kernel(const EvalParams q, int nblk) {
  __shared__ double S[kAcc];
  if (threadIdx.x < kAcc) {
    double v = 0.0;
    for (int b = 0; b < nblk; b++) v += (double)q.part[(long long)b * kAcc + threadIdx.x];
    S[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const float eps = 1e-30f;   // the modules' default (loss_maker.py:523, 540, 557, 578, 638, 674)
  auto F = [&](int i) { return (float)S[i]; };
  float* o = q.out;
  const float n_all = (float)(4.0 * (double)q.vox);
  o[SR3D_EVAL_L1] = F(A_ABS) / n_all;
  o[SR3D_EVAL_L2] = F(A_SQ) / n_all;
  // masks are broadcast over the 4 channels before torch.sum: the denominator is 4 * sum(mask)
  o[SR3D_EVAL_MASKED_L1] = F(A_B_ABS) / (4.f * F(A_B) + eps);
  o[SR3D_EVAL_MASKED_L2] = F(A_B_SQ) / (4.f * F(A_B) + eps);
  o[SR3D_EVAL_MASKED_L1_NEAR_WALL] = F(A_NEAR_ABS) / (4.f * F(A_NEAR) + eps);
  o[SR3D_EVAL_MASKED_L2_NEAR_WALL] = F(A_NEAR_SQ) / (4.f * F(A_NEAR) + eps);
  const float n_grid = F(A_B_INT) - F(A_NEAR_INT);   // loss_maker.py:110 (no eps: 0 / 0 = nan there too)
  o[SR3D_EVAL_RESIDUAL_CONTINUITY] = F(A_RES_P) / n_grid;
  o[SR3D_EVAL_RESIDUAL_CONTINUITY_TARGET] = F(A_RES_T) / n_grid;
  o[SR3D_EVAL_ABS_DIFF_TEMPERATURE] = F(A_T) / (F(A_B) + eps);
  o[SR3D_EVAL_DIFF_VELOCITY_NORM] = F(A_V) / (F(A_B) + eps);
  o[SR3D_EVAL_ABS_DIFF_TEMPERATURE_LEV] = F(A_T_LEV) / (F(A_B_LEV) + eps);
  o[SR3D_EVAL_DIFF_VELOCITY_NORM_LEV] = F(A_V_LEV) / (F(A_B_LEV) + eps);
  o[SR3D_EVAL_ABS_DIFF_DIVERGENCE] = F(A_DDIV) / n_grid;
  o[SR3D_EVAL_DIFF_OMEGA_NORM] = F(A_OMEGA) / n_grid;
  // raw sums (the weighted losses of loss_maker.py:216-255 combine them with a run-time weight)
  o[SR3D_EVAL_SUM_ABS] = F(A_ABS), o[SR3D_EVAL_SUM_MASK_ABS] = F(A_B_ABS);
  o[SR3D_EVAL_SUM_SQ] = F(A_SQ), o[SR3D_EVAL_SUM_MASK_SQ] = F(A_B_SQ);
  o[SR3D_EVAL_SUM_MASK] = F(A_B);
}

int blocks_for(long long vox) {
  const long long b = (vox + kThreads - 1) / kThreads;
  return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}

}  // namespace

extern "C" {

size_t sr3d_eval_metrics_workspace_bytes(int B, int Z, int Y, int X) {
  (void)B, (void)Z, (void)Y, (void)X;
  return (size_t)kMaxBlocks * kAcc * sizeof(float);
}

int sr3d_eval_metrics(const void* p, const void* t, const void* b, int B, int Z, int Y, int X, const float stds[4],
                      float delta_meter, int lev, void* out, void* workspace, void* stream) {
  SR3D_CHECK(p && t && b && stds && out && workspace, SR3D_E_ARG, "eval_metrics: null pointer");
  SR3D_CHECK(B > 0 && Z > 0 && Y > 0 && X > 0, SR3D_E_ARG, "eval_metrics: non-positive dimension");
  SR3D_CHECK((long long)Z * Y * X < (1ll << 31), SR3D_E_ARG, "eval_metrics: grid has >= 2^31 voxels");
  SR3D_CHECK(delta_meter > 0.f, SR3D_E_ARG, "eval_metrics: delta_meter must be positive");
  SR3D_CHECK(lev >= 0 && lev < Z, SR3D_E_ARG, "eval_metrics: level %d is outside the grid (Z = %d)", lev, Z);
  EvalParams q{};
  q.p = (const float*)p, q.t = (const float*)t, q.b = (const float*)b;
  q.B = B, q.Z = Z, q.Y = Y, q.X = X, q.lev = lev;
  for (int i = 0; i < 4; i++) q.s[i] = stds[i];
  q.w = (float)(1.0 / (2.0 * (double)delta_meter));   // math_helper.py:17: the kernel weight is a python float, then an fp32 tensor
  q.part = (float*)workspace, q.out = (float*)out;
  q.vox = (long long)B * Z * Y * X;
  const int nb = blocks_for(q.vox);
  SrProfScope prof(SR3D_PROF_EVAL, 36.0 * (double)q.vox, (hipStream_t)stream);   // p, t (32 B) + b (4 B) per voxel
  hipLaunchKernelGGL(eval_kernel, dim3(nb), dim3(kThreads), 0, (hipStream_t)stream, q);
  SR3D_HIP(hipGetLastError());
  hipLaunchKernelGGL(eval_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, q, nb);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

}  // extern "C"
