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

// ---- the pass: a 2.5-D marching stencil --------------------------------------------------------------------------
// A workgroup owns a column of the grid: MY = 8 rows x MX = 62 columns, ZS planes, and walks it plane by plane.
//   x: one row = one wave; lanes 0 and 63 are halo columns, so the x neighbours of lanes 1..62 are two wave SHUFFLES
//      away and every global load is one contiguous 256-byte row segment;
//   y: waves 0 and 9 are halo rows; a plane's values go through an LDS tile [field][10 rows][64] (double-buffered by
//      plane parity: one barrier per plane) and the y neighbours are read back from the rows above and below;
//   z: each thread keeps the previous / current / next plane of its column in REGISTERS (rolling window), so a value is
//      loaded from HBM once per tile (plus the halo shell: 10/8 x 64/62 x (ZS+2)/ZS = 1.45x at ZS = 16).
// The near-wall flag (27-tap box of the mask in the reference) is the same separable walk: sum over x by shuffles,
// over y through LDS, over z in the rolling window.
constexpr int MY = 8, MX = 62, MROWS = MY + 2, MZS = 16;
constexpr int MF = 7;                            // fields that need y neighbours: 3 velocities x (p, t) + mask x-sum
constexpr int kMarchThreads = MROWS * 64;

__global__ __launch_bounds__(kMarchThreads) void eval_kernel(const EvalParams q) {
#pragma clang fp contract(off)
  __shared__ float tile[2][MF][MROWS][64];
  const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;          // row 0 / MROWS-1: halo rows
  int blk = blockIdx.x;
  const int ntx = (q.X + MX - 1) / MX, nty = (q.Y + MY - 1) / MY;
  const int tix = blk % ntx;
  blk /= ntx;
  const int tiy = blk % nty;
  const int tiz = blk / nty;
  const int bi = blockIdx.y;
  const int x = tix * MX - 1 + lane, y = tiy * MY - 1 + row;
  const int z_lo = tiz * MZS, z_hi = min(z_lo + MZS, q.Z);             // output planes [z_lo, z_hi)
  const bool in_xy = (unsigned)x < (unsigned)q.X && (unsigned)y < (unsigned)q.Y;
  const bool owner = in_xy && lane >= 1 && lane <= MX && row >= 1 && row <= MY;    // this thread reports outputs
  const bool inner_xy = owner && x >= 1 && x < q.X - 1 && y >= 1 && y < q.Y - 1;
  const long long zyx = (long long)q.Z * q.Y * q.X, sz = (long long)q.Y * q.X;
  const long long col = (long long)y * q.X + x;
  const float* pb = q.p + (long long)bi * 4 * zyx + col;
  const float* tb = q.t + (long long)bi * 4 * zyx + col;
  const float* mb = q.b + (long long)bi * zyx + col;

  float acc[kAcc];
#pragma unroll
  for (int i = 0; i < kAcc; i++) acc[i] = 0.f;

  // rolling window (index 0 = plane j-1, 1 = plane j, 2 = plane j+1 when plane j is reported)
  float vel[3][2][3];      // scaled velocity [plane][p / t][u, v, w]
  float pc1[4], tc1[4];    // all four channels of plane j (non-stencil sums)
  float b1 = 0.f, sxy[3] = {0.f, 0.f, 0.f};
  float gx1[2][3], gy1[2][3];   // x / y derivatives of plane j, formed when plane j arrived
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
      for (int c = 0; c < 3; c++) vel[a][k][c] = 0.f;
#pragma unroll
  for (int c = 0; c < 4; c++) pc1[c] = tc1[c] = 0.f;
#pragma unroll
  for (int k = 0; k < 2; k++)
#pragma unroll
    for (int c = 0; c < 3; c++) gx1[k][c] = gy1[k][c] = 0.f;

  for (int zz = z_lo - 1; zz <= z_hi; zz++) {          // plane zz arrives; plane j = zz - 1 is reported
    const bool in_z = (unsigned)zz < (unsigned)q.Z;
    const bool live = in_xy && in_z;
    float pn[4], tn[4], bn = 1.f;                       // outside the grid: zero field, "fluid" mask (1 - b = 0)
#pragma unroll
    for (int c = 0; c < 4; c++) pn[c] = tn[c] = 0.f;
    if (live) {
      const long long o = (long long)zz * sz;
#pragma unroll
      for (int c = 0; c < 4; c++) pn[c] = pb[c * zyx + o], tn[c] = tb[c * zyx + o];
      bn = mb[o];
    }
    float vn[2][3];
#pragma unroll
    for (int c = 0; c < 3; c++) vn[0][c] = q.s[c + 1] * pn[c + 1], vn[1][c] = q.s[c + 1] * tn[c + 1];
    // x neighbours: shuffles (lanes 0 / 63 get garbage from the wrap-around; they are halo lanes and never report)
    float gxn[2][3];
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
      for (int c = 0; c < 3; c++)
        gxn[k][c] = cdiff(__shfl_down(vn[k][c], 1, 64), __shfl_up(vn[k][c], 1, 64), q.w);
    const float ib = 1.f - bn;
    const float sx = (__shfl_up(ib, 1, 64) + ib) + __shfl_down(ib, 1, 64);
    // y neighbours through LDS
    float (*T)[MROWS][64] = tile[zz & 1];
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
      for (int c = 0; c < 3; c++) T[k * 3 + c][row][lane] = vn[k][c];
    T[6][row][lane] = sx;
    __syncthreads();
    float gyn[2][3], sxy_n = 0.f;
    if (row >= 1 && row <= MY) {
#pragma unroll
      for (int k = 0; k < 2; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) gyn[k][c] = cdiff(T[k * 3 + c][row + 1][lane], T[k * 3 + c][row - 1][lane], q.w);
      sxy_n = (T[6][row - 1][lane] + sx) + T[6][row + 1][lane];
    } else {
#pragma unroll
      for (int k = 0; k < 2; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) gyn[k][c] = 0.f;
    }
    sxy[2] = sxy_n;
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
      for (int c = 0; c < 3; c++) vel[2][k][c] = vn[k][c];

    // ---- report plane j = zz - 1 (its z+1 neighbour has just arrived)
    const int j = zz - 1;
    if (owner && j >= z_lo && j < z_hi) {
      const float bv = b1;
      const float box = (sxy[0] + sxy[1]) + sxy[2];
      const float near = ((box > 0.f ? 1.f : 0.f) * bv > 0.f) ? 1.f : 0.f;
      float sa = 0.f, sq = 0.f;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const float d = pc1[c] - tc1[c];
        sa += fabsf(d), sq += d * d;
      }
      acc[A_ABS] += sa, acc[A_SQ] += sq;
      acc[A_B] += bv, acc[A_B_ABS] += bv * sa, acc[A_B_SQ] += bv * sq;
      acc[A_NEAR] += near, acc[A_NEAR_ABS] += near * sa, acc[A_NEAR_SQ] += near * sq;
      // |T_p - T_t| * std_T and || std_v * v_p - std_v * v_t ||  (loss_maker.py:636-703)
      const float dT = fabsf(pc1[0] - tc1[0]) * q.s[0];
      float n2 = 0.f;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float dv = vel[1][0][c] - vel[1][1][c];
        n2 += dv * dv;
      }
      const float dV = sqrtf(n2);
      acc[A_T] += bv * dT, acc[A_V] += bv * dV;
      if (j == q.lev) acc[A_T_LEV] += bv * dT, acc[A_V_LEV] += bv * dV, acc[A_B_LEV] += bv;
      if (inner_xy && j >= 1 && j < q.Z - 1) {
        acc[A_B_INT] += bv, acc[A_NEAR_INT] += near;
        const float M = bv * (1.f - near);
        float res[2], om[2][3];
#pragma unroll
        for (int k = 0; k < 2; k++) {
          float gz[3];
#pragma unroll
          for (int c = 0; c < 3; c++) gz[c] = cdiff(vel[2][k][c], vel[0][k][c], q.w);
          res[k] = (gx1[k][0] + gy1[k][1]) + gz[2];
          om[k][0] = gy1[k][2] - gz[1];       // dw/dy - dv/dz
          om[k][1] = gz[0] - gx1[k][2];       // du/dz - dw/dx
          om[k][2] = gx1[k][1] - gy1[k][0];   // dv/dx - du/dy
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
    // ---- roll the window: plane zz becomes "plane j" of the next iteration
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
      for (int c = 0; c < 3; c++) {
        vel[0][k][c] = vel[1][k][c], vel[1][k][c] = vel[2][k][c];
        gx1[k][c] = gxn[k][c], gy1[k][c] = gyn[k][c];
      }
#pragma unroll
    for (int c = 0; c < 4; c++) pc1[c] = pn[c], tc1[c] = tn[c];
    b1 = bn;
    sxy[0] = sxy[1], sxy[1] = sxy[2];
  }

  __shared__ float sm[kMarchThreads / 64][kAcc];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kAcc; i++) {
    const float v = wave_sum(acc[i]);
    if (lane == 0) sm[row][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < kAcc) {
    float v = 0.f;
    for (int w = 0; w < kMarchThreads / 64; w++) v += sm[w][threadIdx.x];
    q.part[((long long)blockIdx.y * gridDim.x + blockIdx.x) * kAcc + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(kThreads) void eval_final_kernel(const EvalParams q, int nblk) {
  // every accumulator: 256 strided partial sums (independent loads, all in flight), then a fixed-order tree in LDS
  __shared__ double S[kAcc];
  __shared__ double red[kThreads];
  for (int k = 0; k < kAcc; k++) {
    double v = 0.0;
    for (int b = threadIdx.x; b < nblk; b += kThreads) v += (double)q.part[(long long)b * kAcc + k];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) S[k] = red[0];
    __syncthreads();
  }
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

int march_blocks(int Z, int Y, int X) { return ((X + MX - 1) / MX) * ((Y + MY - 1) / MY) * ((Z + MZS - 1) / MZS); }

}  // namespace

extern "C" {

size_t sr3d_eval_metrics_workspace_bytes(int B, int Z, int Y, int X) {
  if (B <= 0 || Z <= 0 || Y <= 0 || X <= 0) return 0;
  return (size_t)B * march_blocks(Z, Y, X) * kAcc * sizeof(float);
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
  const int nbs = march_blocks(Z, Y, X), nb = nbs * B;
  SR3D_CHECK(B <= 65535, SR3D_E_ARG, "eval_metrics: batch too large");
  SrProfScope prof(SR3D_PROF_EVAL, 36.0 * (double)q.vox, (hipStream_t)stream);   // p, t (32 B) + b (4 B) per voxel
  hipLaunchKernelGGL(eval_kernel, dim3(nbs, B), dim3(kMarchThreads), 0, (hipStream_t)stream, q);
  SR3D_HIP(hipGetLastError());
  hipLaunchKernelGGL(eval_final_kernel, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, q, nb);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

}  // extern "C"
