// Fused training losses (forward value + dL/dprediction) for gfx950.
//
//  - MyL1Loss                         (reference pytorch/src/loss_maker.py:194-202)
//  - MixedDivergenceGradientL2Loss    (loss_maker.py:358-450) with its helpers
//    calc_mask_near_build_wall (:57-83), differentiate_along_{x,y,z}
//    (math_helper.py:6-60, padding 0) and _calc_residual_continuity_eq (:115-130).
//
// The reference evaluates the mixed loss with ~40 elementwise launches and ten
// 27-tap depthwise convolutions; here it is two stencil passes:
//   pass 1  reads p, t, b once (neighbours through L1/L2), accumulates the four
//           sums (sum d^2, sum M*|grad d|^2, sum M*dd^2, sum M) and stores the
//           two per-voxel fields the adjoint needs (M and E = M*dd);
//   pass 2  applies the adjoint of the two-tap stencils analytically and writes
//           dL/dp.  Reductions are two-stage with a fixed order (deterministic).
#include "sr3d_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;

__device__ __forceinline__ float block_sum(float v, float* red) {
  // wave64 shuffle reduction, then the 4 wave partials through LDS
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------- L1
__global__ __launch_bounds__(kThreads) void l1_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                      long long n, float* __restrict__ part,
                                                      float* __restrict__ dldp, float inv_n) {
  __shared__ float red[4];
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  for (long long i = i0; i < n4; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(p)[i];
    const float4 b = reinterpret_cast<const float4*>(t)[i];
    float4 g;
#define ONE(q)                                               \
  {                                                          \
    const float d = a.q - b.q;                               \
    s += fabsf(d);                                           \
    g.q = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);        \
  }
    ONE(x) ONE(y) ONE(z) ONE(w)
#undef ONE
    if (dldp) reinterpret_cast<float4*>(dldp)[i] = g;
  }
  for (long long i = n4 * 4 + i0; i < n; i += stride) {
    const float d = p[i] - t[i];
    s += fabsf(d);
    if (dldp) dldp[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
  }
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ void l1_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out, double inv_n) {
  __shared__ double red[kThreads];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += kThreads) s += (double)part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = kThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] * inv_n);
}

// ---------------------------------------------------------------- mixed loss
struct MixParams {
  const float* p;
  const float* t;
  const float* b;
  int B, Z, Y, X;
  float s[3];        // velocity scales (u, v, w)
  float w_g, w_d;
  float delta, mean_scale;
  float* fieldM;     // [B][Z][Y][X]
  float* fieldE;
  float* part;       // [4][nblocks]
  float* sums;       // 4 floats: sum d^2, grd numerator, div numerator, sum M
  float* terms;      // out: mse, grd, div, total
  float* dldp;
  const float* wts;  // device, 3 floats: d(result)/d(mse, grd, div); null = (1, w_g, w_d)
};

__device__ __forceinline__ float cdiff(float hi, float lo, float w) { return hi * w + lo * (-w); }

// Pass 1 as a 2.5-D marching stencil (the layout of csrc/sr3d_eval.hip): a workgroup owns a column of the grid, 8 rows
// x 62 columns x 16 planes, and walks it plane by plane.  x neighbours: wave shuffles (one row = one wave, lanes 0 and
// 63 are halo columns, every load is one contiguous 256-byte row segment); y neighbours: an LDS tile [field][10 rows][64]
// (waves 0 and 9 are halo rows; double-buffered by plane parity, one barrier per plane); z neighbours: a rolling 3-plane
// window in registers.  The near-wall flag's 27-tap box over the mask is the same separable walk.  (The first version
// -- one thread per voxel, 27 + 56 cached gathers and three integer divisions -- took 1.19 ms at batch 4.)
constexpr int MY = 8, MX = 62, MROWS = MY + 2, MZS = 16;
constexpr int MF = 7;                            // LDS fields: d (4 channels), s_v * v of p and of t, mask x-sum
constexpr int kMarchThreads = MROWS * 64;

__host__ __device__ inline int march_blocks(int Z, int Y, int X) {
  return ((X + MX - 1) / MX) * ((Y + MY - 1) / MY) * ((Z + MZS - 1) / MZS);
}

__global__ __launch_bounds__(kMarchThreads) void mix_pass1_kernel(const MixParams q) {
  __shared__ float tile[2][MF][MROWS][64];
  __shared__ float red4[kMarchThreads / 64][4];
  const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
  int blk = blockIdx.x;
  const int ntx = (q.X + MX - 1) / MX, nty = (q.Y + MY - 1) / MY;
  const int tix = blk % ntx;
  blk /= ntx;
  const int tiy = blk % nty;
  const int tiz = blk / nty;
  const int bi = blockIdx.y;
  const int x = tix * MX - 1 + lane, y = tiy * MY - 1 + row;
  const int z_lo = tiz * MZS, z_hi = min(z_lo + MZS, q.Z);
  const bool in_xy = (unsigned)x < (unsigned)q.X && (unsigned)y < (unsigned)q.Y;
  const bool owner = in_xy && lane >= 1 && lane <= MX && row >= 1 && row <= MY;
  const bool inner_xy = owner && x >= 1 && x < q.X - 1 && y >= 1 && y < q.Y - 1;
  const long long zyx = (long long)q.Z * q.Y * q.X, sz = (long long)q.Y * q.X;
  const long long col = (long long)y * q.X + x;
  const float* pb = q.p + (long long)bi * 4 * zyx + col;
  const float* tb = q.t + (long long)bi * 4 * zyx + col;
  const float* mb = q.b + (long long)bi * zyx + col;
  const float w5 = 1.f / (2.f * q.delta);
  const bool do_g = q.w_g != 0.f, do_d = q.w_d != 0.f;
  float a_mse = 0.f, a_grd = 0.f, a_div = 0.f, a_m = 0.f;

  // rolling window: index 0 = plane j-1, 1 = plane j (the one being reported), 2 = plane j+1
  float d[3][4], wv[3][2];          // p - t (4 channels); s_w * w of p and of t
  float b1 = 0.f, sxy[3] = {0.f, 0.f, 0.f};
  float g2xy1 = 0.f;                // sum_c (d_x^2 + d_y^2) of plane j
  float dxy1[2] = {0.f, 0.f};       // d_x(s_u u) + d_y(s_v v) of plane j, for p and for t
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int c = 0; c < 4; c++) d[a][c] = 0.f;
    wv[a][0] = wv[a][1] = 0.f;
  }

  for (int zz = z_lo - 1; zz <= z_hi; zz++) {
    const bool live = in_xy && (unsigned)zz < (unsigned)q.Z;
    float pn[4], tn[4], bn = 1.f;    // outside the grid: zero field, "fluid" mask (1 - b = 0: zero padding of the box)
#pragma unroll
    for (int c = 0; c < 4; c++) pn[c] = tn[c] = 0.f;
    if (live) {
      const long long o = (long long)zz * sz;
#pragma unroll
      for (int c = 0; c < 4; c++) pn[c] = pb[c * zyx + o], tn[c] = tb[c * zyx + o];
      bn = mb[o];
    }
    float dn[4];
#pragma unroll
    for (int c = 0; c < 4; c++) dn[c] = pn[c] - tn[c];
    const float un[2] = {q.s[0] * pn[1], q.s[0] * tn[1]}, vn[2] = {q.s[1] * pn[2], q.s[1] * tn[2]};
    // x direction: shuffles (lanes 0 / 63 receive wrapped values; they are halo lanes and never report)
    float g2 = 0.f;
    if (do_g) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const float gx = cdiff(__shfl_down(dn[c], 1, 64), __shfl_up(dn[c], 1, 64), 0.5f);
        g2 += gx * gx;
      }
    }
    float dux[2] = {0.f, 0.f};
    if (do_d) {
#pragma unroll
      for (int k = 0; k < 2; k++) dux[k] = cdiff(__shfl_down(un[k], 1, 64), __shfl_up(un[k], 1, 64), w5);
    }
    const float ib = 1.f - bn;
    const float sx = (__shfl_up(ib, 1, 64) + ib) + __shfl_down(ib, 1, 64);
    // y direction: through LDS
    float (*T)[MROWS][64] = tile[zz & 1];
#pragma unroll
    for (int c = 0; c < 4; c++) T[c][row][lane] = dn[c];
    T[4][row][lane] = vn[0], T[5][row][lane] = vn[1], T[6][row][lane] = sx;
    __syncthreads();
    float dvy[2] = {0.f, 0.f}, sxy_n = 0.f;
    if (row >= 1 && row <= MY) {
      if (do_g) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const float gy = cdiff(T[c][row + 1][lane], T[c][row - 1][lane], 0.5f);
          g2 += gy * gy;
        }
      }
      if (do_d) {
#pragma unroll
        for (int k = 0; k < 2; k++) dvy[k] = cdiff(T[4 + k][row + 1][lane], T[4 + k][row - 1][lane], w5);
      }
      sxy_n = (T[6][row - 1][lane] + sx) + T[6][row + 1][lane];
    }
    sxy[2] = sxy_n;
#pragma unroll
    for (int c = 0; c < 4; c++) d[2][c] = dn[c];
    wv[2][0] = q.s[2] * pn[3], wv[2][1] = q.s[2] * tn[3];

    // ---- report plane j = zz - 1
    const int j = zz - 1;
    if (owner && j >= z_lo && j < z_hi) {
#pragma unroll
      for (int c = 0; c < 4; c++) a_mse += d[1][c] * d[1][c];
      float M = 0.f, E = 0.f;
      if (inner_xy && j >= 1 && j < q.Z - 1) {
        const float box = (sxy[0] + sxy[1]) + sxy[2];
        const float near = ((box > 0.f ? 1.f : 0.f) * b1 > 0.f) ? 1.f : 0.f;
        M = b1 * (1.f - near);
        a_m += M;
        if (do_g) {
          float gg = g2xy1;
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const float gz = cdiff(d[2][c], d[0][c], 0.5f);
            gg += gz * gz;
          }
          a_grd += gg * M;
        }
        if (do_d) {
          const float div_t = dxy1[1] + cdiff(wv[2][1], wv[0][1], w5);
          const float div_p = dxy1[0] + cdiff(wv[2][0], wv[0][0], w5);
          const float dd = (div_t - div_p) * q.delta / q.mean_scale;
          a_div += dd * dd * M;
          E = M * dd;
        }
      }
      if (q.fieldM) {
        const long long o = (long long)bi * zyx + (long long)j * sz + col;
        q.fieldM[o] = M;
        q.fieldE[o] = E;
      }
    }
    // ---- roll
#pragma unroll
    for (int c = 0; c < 4; c++) d[0][c] = d[1][c], d[1][c] = d[2][c];
#pragma unroll
    for (int k = 0; k < 2; k++) wv[0][k] = wv[1][k], wv[1][k] = wv[2][k], dxy1[k] = dux[k] + dvy[k];
    g2xy1 = g2;
    b1 = bn;
    sxy[0] = sxy[1], sxy[1] = sxy[2];
  }

  // block sums in a fixed order: wave shuffles, then the 10 wave partials
  float v4[4] = {a_mse, a_grd, a_div, a_m};
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    float v = v4[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red4[row][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = 0.f;
    for (int w = 0; w < kMarchThreads / 64; w++) v += red4[w][threadIdx.x];
    const int nb = gridDim.x * gridDim.y;
    q.part[threadIdx.x * nb + blockIdx.y * gridDim.x + blockIdx.x] = v;
  }
}

// one block: finish the four sums in double, emit the loss terms
__global__ __launch_bounds__(kThreads) void mix_final_kernel(const MixParams q, int nb) {
  __shared__ double red[kThreads];
  double tot[4];
  for (int k = 0; k < 4; k++) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += kThreads) s += (double)q.part[k * nb + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    tot[k] = red[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double n = 4.0 * q.B * (double)q.Z * q.Y * q.X;
    const float mse = (float)(tot[0] / n);
    const float sumM = (float)tot[3];
    const float grd = q.w_g != 0.f ? (float)tot[1] / (4.f * sumM + 1.f) : 0.f;
    const float div = q.w_d != 0.f ? (float)tot[2] / (sumM + 1.f) : 0.f;
    q.terms[0] = mse, q.terms[1] = grd, q.terms[2] = div;
    q.terms[3] = mse + q.w_g * grd + q.w_d * div;
    q.sums[0] = (float)tot[0], q.sums[1] = (float)tot[1], q.sums[2] = (float)tot[2], q.sums[3] = sumM;
  }
}

__global__ __launch_bounds__(kThreads) void mix_pass2_kernel(const MixParams q) {
  const long long zyx = (long long)q.Z * q.Y * q.X;
  const long long total = (long long)q.B * zyx;
  const long long sy = q.X, sz = (long long)q.Y * q.X;
  const float sumM = q.sums[3];
  // weights of the three terms in the scalar whose gradient is wanted (autograd may ask for any mix)
  const float u_mse = q.wts ? q.wts[0] : 1.f;
  const float u_grd = q.w_g != 0.f ? (q.wts ? q.wts[1] : q.w_g) : 0.f;   // a skipped term has no gradient
  const float u_div = q.w_d != 0.f ? (q.wts ? q.wts[2] : q.w_d) : 0.f;
  const float c_mse = u_mse * 2.f / (4.f * (float)q.B * (float)zyx);
  const float c_grd = u_grd / (4.f * sumM + 1.f);
  const float w5 = 1.f / (2.f * q.delta);
  const float k = q.delta / q.mean_scale;
  const float c_div = u_div / (sumM + 1.f) * (-2.f * k * w5);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int x = (int)(r % q.X);
    r /= q.X;
    const int y = (int)(r % q.Y);
    r /= q.Y;
    const int z = (int)(r % q.Z);
    const int b = (int)(r / q.Z);
    const long long sp = i - (long long)b * zyx;
    const float* P = q.p + (long long)b * 4 * zyx + sp;
    const float* T = q.t + (long long)b * 4 * zyx + sp;
    const float* Mf = q.fieldM + i;
    const float* Ef = q.fieldE + i;
    // neighbours o = q -+ e that can carry a stencil (M is zero outside the interior, but the
    // far point q -+ 2e must exist to recompute the difference)
    const bool xm = x >= 2, xp = x < q.X - 2, ym = y >= 2, yp = y < q.Y - 2, zm = z >= 2, zp = z < q.Z - 2;
    const float Mxm = xm ? Mf[-1] : 0.f, Mxp = xp ? Mf[1] : 0.f;
    const float Mym = ym ? Mf[-sy] : 0.f, Myp = yp ? Mf[sy] : 0.f;
    const float Mzm = zm ? Mf[-sz] : 0.f, Mzp = zp ? Mf[sz] : 0.f;
    float out[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const float* Pc = P + c * zyx;
      const float* Tc = T + c * zyx;
      const float d0 = Pc[0] - Tc[0];
      float g = c_mse * d0;
      if (c_grd != 0.f) {
        float acc = 0.f;
        if (Mxm != 0.f) acc += Mxm * cdiff(d0, Pc[-2] - Tc[-2], 0.5f);
        if (Mxp != 0.f) acc -= Mxp * cdiff(Pc[2] - Tc[2], d0, 0.5f);
        if (Mym != 0.f) acc += Mym * cdiff(d0, Pc[-2 * sy] - Tc[-2 * sy], 0.5f);
        if (Myp != 0.f) acc -= Myp * cdiff(Pc[2 * sy] - Tc[2 * sy], d0, 0.5f);
        if (Mzm != 0.f) acc += Mzm * cdiff(d0, Pc[-2 * sz] - Tc[-2 * sz], 0.5f);
        if (Mzp != 0.f) acc -= Mzp * cdiff(Pc[2 * sz] - Tc[2 * sz], d0, 0.5f);
        g += c_grd * acc;
      }
      out[c] = g;
    }
    if (c_div != 0.f) {
      const float ex = (x >= 1 ? Ef[-1] : 0.f) - (x < q.X - 1 ? Ef[1] : 0.f);
      const float ey = (y >= 1 ? Ef[-sy] : 0.f) - (y < q.Y - 1 ? Ef[sy] : 0.f);
      const float ez = (z >= 1 ? Ef[-sz] : 0.f) - (z < q.Z - 1 ? Ef[sz] : 0.f);
      out[1] += c_div * q.s[0] * ex;
      out[2] += c_div * q.s[1] * ey;
      out[3] += c_div * q.s[2] * ez;
    }
    float* D = q.dldp + (long long)b * 4 * zyx + sp;
#pragma unroll
    for (int c = 0; c < 4; c++) D[c * zyx] = out[c];
  }
}

inline int grid_for(long long n, int per_thread) {
  long long b = (n + (long long)kThreads * per_thread - 1) / ((long long)kThreads * per_thread);
  return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}

__global__ __launch_bounds__(kThreads) void weighted_lp_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                                   const float* __restrict__ m, long long total, long long vox,
                                                                   int Cc, int power, const float* __restrict__ coef,
                                                                   float* __restrict__ g) {
  const float c_in = coef[0], c_out = coef[1];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long ch = i / vox;
    const float mv = m[(ch / Cc) * vox + (i - ch * vox)];
    const float d = p[i] - t[i];
    const float e = power == 1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d;
    g[i] = e * (mv * c_in + (1.f - mv) * c_out);
  }
}

// ---------------------------------------------------------------- SSIM3D (reference src/ssim.py:52-115)
// The reference filters six fields (x1 m, x2 m, m, (x1 m)^2, (x2 m)^2, x1 x2 m^2) with a dense w x w x w window
// (1331 taps at the default size 11).  The window is an outer product, so three 1-D passes do the same: x (fused with
// forming the six fields), y, and z (fused with the SSIM formula and the mean).
constexpr int kSsimMaxTaps = 15;
struct SsimParams {
  const float* a;      // img1 (B, C, Z, Y, X)
  const float* b;      // img2
  const float* m;      // mask (B, Cm, Z, Y, X), Cm = 1 or C
  float* f0;           // 6 fields after the x pass   [6][B*C*vox]
  float* f1;           // 6 fields after the y pass
  float* map;          // optional ssim map
  float* part;         // per-block partial sums
  int B, C, Cm, Z, Y, X, n;
  float w[kSsimMaxTaps];
  float c1, c2, eps;
  long long total;     // B * C * Z * Y * X
};

__global__ __launch_bounds__(kThreads) void ssim_x_kernel(const SsimParams q) {
  const int r = q.n / 2;
  const long long vox = (long long)q.Z * q.Y * q.X;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < q.total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % q.X);
    const long long ch = i / vox;                       // b * C + c
    const long long mrow = (q.Cm == 1 ? (ch / q.C) : ch) * vox + (i - ch * vox) - x;
    const long long row = i - x;
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < q.n; k++) {
      const int xx = x + k - r;
      if ((unsigned)xx >= (unsigned)q.X) continue;
      const float mv = q.m[mrow + xx];
      const float u = q.a[row + xx] * mv, v = q.b[row + xx] * mv, wk = q.w[k];
      s[0] += wk * u, s[1] += wk * v, s[2] += wk * mv, s[3] += wk * (u * u), s[4] += wk * (v * v), s[5] += wk * (u * v);
    }
#pragma unroll
    for (int f = 0; f < 6; f++) q.f0[f * q.total + i] = s[f];
  }
}

__global__ __launch_bounds__(kThreads) void ssim_y_kernel(const SsimParams q) {
  const int r = q.n / 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < q.total; i += (long long)gridDim.x * blockDim.x) {
    const int y = (int)((i / q.X) % q.Y);
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < q.n; k++) {
      const int yy = y + k - r;
      if ((unsigned)yy >= (unsigned)q.Y) continue;
      const long long o = i + (long long)(yy - y) * q.X;
      const float wk = q.w[k];
#pragma unroll
      for (int f = 0; f < 6; f++) s[f] += wk * q.f0[f * q.total + o];
    }
#pragma unroll
    for (int f = 0; f < 6; f++) q.f1[f * q.total + i] = s[f];
  }
}

__global__ __launch_bounds__(kThreads) void ssim_z_kernel(const SsimParams q) {
  __shared__ float red[4];
  const int r = q.n / 2;
  const long long yx = (long long)q.Y * q.X;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < q.total; i += (long long)gridDim.x * blockDim.x) {
    const int z = (int)((i / yx) % q.Z);
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < q.n; k++) {
      const int zz = z + k - r;
      if ((unsigned)zz >= (unsigned)q.Z) continue;
      const long long o = i + (long long)(zz - z) * yx;
      const float wk = q.w[k];
#pragma unroll
      for (int f = 0; f < 6; f++) s[f] += wk * q.f1[f * q.total + o];
    }
    const float wt = s[2] + q.eps;
    const float mu1 = s[0] / wt, mu2 = s[1] / wt;
    const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = s[3] / wt - mu1s, s2 = s[4] / wt - mu2s, s12 = s[5] / wt - mu12;
    const float v = ((2.f * mu12 + q.c1) * (2.f * s12 + q.c2)) / ((mu1s + mu2s + q.c1) * (s1 + s2 + q.c2));
    if (q.map) q.map[i] = v;
    acc += v;
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) q.part[blockIdx.x] = tot;
}

}  // namespace

extern "C" {

static size_t loss_part_floats(int B, int Z, int Y, int X) {   // 4 partial sums per workgroup of pass 1
  const size_t nb = (size_t)(B > 0 ? B : 1) * march_blocks(Z > 0 ? Z : 1, Y > 0 ? Y : 1, X > 0 ? X : 1);
  return 4 * (nb > (size_t)kMaxBlocks ? nb : (size_t)kMaxBlocks);
}

size_t sr3d_loss_workspace_bytes(int B, int Z, int Y, int X) {
  const size_t vox = (size_t)B * Z * Y * X;
  return (2 * vox + loss_part_floats(B, Z, Y, X) + 16) * sizeof(float);
}

int sr3d_l1_fwd_bwd(const void* p, const void* t, long long n, void* loss_out, void* dLdp, void* workspace,
                    void* stream) {
  SR3D_CHECK(p && t && loss_out && workspace && n > 0, SR3D_E_ARG, "l1_fwd_bwd: bad argument");
  SR3D_CHECK(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(dLdp)) &
              15) == 0,
             SR3D_E_ARG, "l1_fwd_bwd: pointers must be 16-byte aligned");
  SrProfScope prof(SR3D_PROF_LOSS, (dLdp ? 12.0 : 8.0) * (double)n, (hipStream_t)stream);
  const int nb = grid_for(n, 8);
  hipLaunchKernelGGL(l1_kernel, dim3(nb), dim3(kThreads), 0, (hipStream_t)stream, (const float*)p, (const float*)t, n,
                     (float*)workspace, (float*)dLdp, (float)(1.0 / (double)n));
  SR3D_HIP(hipGetLastError());
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, (const float*)workspace, nb,
                     (float*)loss_out, 1.0 / (double)n);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_mixed_div_grad_l2_fwd_bwd(const void* p, const void* t, const void* b, int B, int Z, int Y, int X,
                                   const float scales[3], float delta_meter, float w_g, float w_d, void* terms_out,
                                   void* dLdp, void* workspace, void* stream) {
  SR3D_CHECK(p && t && b && terms_out && workspace && scales, SR3D_E_ARG, "mixed_loss: null pointer");
  SR3D_CHECK(B > 0 && Z >= 3 && Y >= 3 && X >= 3, SR3D_E_ARG, "mixed_loss: grid must be at least 3^3 (got %d,%d,%d)",
             Z, Y, X);
  SR3D_CHECK(delta_meter > 0.f, SR3D_E_ARG, "mixed_loss: delta_meter must be positive");
  const long long vox = (long long)B * Z * Y * X;
  MixParams q{};
  q.p = (const float*)p, q.t = (const float*)t, q.b = (const float*)b;
  q.B = B, q.Z = Z, q.Y = Y, q.X = X;
  q.s[0] = scales[0], q.s[1] = scales[1], q.s[2] = scales[2];
  q.w_g = w_g, q.w_d = w_d, q.delta = delta_meter;
  // np.mean(scales) in double, then used as a python float against fp32 tensors (loss_maker.py:375,430)
  q.mean_scale = (float)(((double)scales[0] + (double)scales[1] + (double)scales[2]) / 3.0);
  float* ws = (float*)workspace;
  q.fieldM = ws, q.fieldE = ws + vox;
  q.part = ws + 2 * vox, q.sums = ws + 2 * vox + loss_part_floats(B, Z, Y, X);
  q.terms = (float*)terms_out, q.dldp = (float*)dLdp;
  SR3D_CHECK(B <= 65535, SR3D_E_ARG, "mixed_loss: batch too large");
  // read p, t (32 B) + b (4 B) per voxel, write dL/dp (16 B)
  SrProfScope prof(SR3D_PROF_LOSS, (dLdp ? 52.0 : 36.0) * (double)vox, (hipStream_t)stream);
  const int nbs = march_blocks(Z, Y, X), nb = nbs * B;
  hipLaunchKernelGGL(mix_pass1_kernel, dim3(nbs, B), dim3(kMarchThreads), 0, (hipStream_t)stream, q);
  SR3D_HIP(hipGetLastError());
  hipLaunchKernelGGL(mix_final_kernel, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, q, nb);
  SR3D_HIP(hipGetLastError());
  if (dLdp) {
    hipLaunchKernelGGL(mix_pass2_kernel, dim3(grid_for(vox, 1)), dim3(kThreads), 0, (hipStream_t)stream, q);
    SR3D_HIP(hipGetLastError());
  }
  return SR3D_OK;
}

int sr3d_weighted_lp_bwd(const void* p, const void* t, const void* b, int B, int C, long long voxels, int power,
                         const void* coef, void* dLdp, void* stream) {
  SR3D_CHECK(p && t && b && coef && dLdp && B > 0 && C > 0 && voxels > 0, SR3D_E_ARG, "weighted_lp_bwd: bad argument");
  SR3D_CHECK(power == 1 || power == 2, SR3D_E_ARG, "weighted_lp_bwd: power must be 1 or 2");
  const long long total = (long long)B * C * voxels;
  SrProfScope prof(SR3D_PROF_LOSS, (12.0 + 4.0 / C) * (double)total, (hipStream_t)stream);
  hipLaunchKernelGGL(weighted_lp_bwd_kernel, dim3(grid_for(total, 1)), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float*)p, (const float*)t, (const float*)b, total, voxels, C, power, (const float*)coef,
                     (float*)dLdp);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

size_t sr3d_ssim3d_workspace_bytes(int B, int C, int Z, int Y, int X) {
  return ((size_t)12 * B * C * Z * Y * X + kMaxBlocks) * sizeof(float);
}

int sr3d_ssim3d(const void* img1, const void* img2, const void* mask, int B, int C, int Cm, int Z, int Y, int X,
                const float* window, int n, float max_val, float eps, void* mean_out, void* ssim_map, void* workspace,
                void* stream) {
  SR3D_CHECK(img1 && img2 && mask && window && mean_out && workspace, SR3D_E_ARG, "ssim3d: null pointer");
  SR3D_CHECK(B > 0 && C > 0 && Z > 0 && Y > 0 && X > 0 && (Cm == 1 || Cm == C), SR3D_E_ARG, "ssim3d: bad shape");
  SR3D_CHECK(n >= 1 && n <= kSsimMaxTaps && (n & 1), SR3D_E_ARG, "ssim3d: window size must be odd and <= %d", kSsimMaxTaps);
  SsimParams q{};
  q.a = (const float*)img1, q.b = (const float*)img2, q.m = (const float*)mask;
  q.B = B, q.C = C, q.Cm = Cm, q.Z = Z, q.Y = Y, q.X = X, q.n = n;
  for (int k = 0; k < n; k++) q.w[k] = window[k];
  q.c1 = (max_val * 0.01f) * (max_val * 0.01f), q.c2 = (max_val * 0.03f) * (max_val * 0.03f), q.eps = eps;
  q.total = (long long)B * C * Z * Y * X;
  float* ws = (float*)workspace;
  q.f0 = ws, q.f1 = ws + 6 * q.total, q.part = ws + 12 * q.total, q.map = (float*)ssim_map;
  const int nb = grid_for(q.total, 1);
  hipStream_t st = (hipStream_t)stream;
  SrProfScope prof(SR3D_PROF_EVAL, (12.0 + 4.0 * Cm / C + 4.0) * (double)q.total, st);
  hipLaunchKernelGGL(ssim_x_kernel, dim3(nb), dim3(kThreads), 0, st, q);
  SR3D_HIP(hipGetLastError());
  hipLaunchKernelGGL(ssim_y_kernel, dim3(nb), dim3(kThreads), 0, st, q);
  SR3D_HIP(hipGetLastError());
  hipLaunchKernelGGL(ssim_z_kernel, dim3(nb), dim3(kThreads), 0, st, q);
  SR3D_HIP(hipGetLastError());
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(kThreads), 0, st, (const float*)q.part, nb, (float*)mean_out,
                     1.0 / (double)q.total);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

/* gradient of  wts[0]*mse + wts[1]*grd_mse + wts[2]*div_mse  w.r.t. p, from the fields a previous
 * sr3d_mixed_div_grad_l2_fwd_bwd call (same arguments, same workspace) left in the workspace */
int sr3d_mixed_div_grad_l2_bwd(const void* p, const void* t, int B, int Z, int Y, int X, const float scales[3],
                               float delta_meter, float w_g, float w_d, const void* term_weights, void* dLdp,
                               void* workspace, void* stream) {
  SR3D_CHECK(p && t && dLdp && workspace && scales && term_weights, SR3D_E_ARG, "mixed_loss_bwd: null pointer");
  SR3D_CHECK(B > 0 && Z >= 3 && Y >= 3 && X >= 3 && delta_meter > 0.f, SR3D_E_ARG, "mixed_loss_bwd: bad argument");
  const long long vox = (long long)B * Z * Y * X;
  MixParams q{};
  q.p = (const float*)p, q.t = (const float*)t;
  q.B = B, q.Z = Z, q.Y = Y, q.X = X;
  q.s[0] = scales[0], q.s[1] = scales[1], q.s[2] = scales[2];
  q.w_g = w_g, q.w_d = w_d, q.delta = delta_meter;
  q.mean_scale = (float)(((double)scales[0] + (double)scales[1] + (double)scales[2]) / 3.0);
  float* ws = (float*)workspace;
  q.fieldM = ws, q.fieldE = ws + vox;
  q.part = ws + 2 * vox, q.sums = ws + 2 * vox + loss_part_floats(B, Z, Y, X);
  q.dldp = (float*)dLdp, q.wts = (const float*)term_weights;
  SrProfScope prof(SR3D_PROF_LOSS, 48.0 * (double)vox, (hipStream_t)stream);   // p, t read, dL/dp written
  hipLaunchKernelGGL(mix_pass2_kernel, dim3(grid_for(vox, 1)), dim3(kThreads), 0, (hipStream_t)stream, q);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

}  // extern "C"
