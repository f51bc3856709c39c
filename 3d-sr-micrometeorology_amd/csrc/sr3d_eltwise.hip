// HBM-bound helper kernels of the voxel-SR hot path (gfx950): activation
// backward, voxel shuffle backward, nearest-upsample+concat, mask pyramid,
// near-wall mask, fused Adam.  All are streaming kernels: 16 B per lane where the
// layout allows, one pass over the data, no LDS.
#include "sr3d_common.h"

namespace {

constexpr int kThreads = 256;

inline int blocks_for(long long n, int per_thread = 1) {
  long long b = (n + (long long)kThreads * per_thread - 1) / ((long long)kThreads * per_thread);
  const long long cap = 256 * 16;  // 16 workgroups per CU, grid-stride beyond that
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// derivative of the activation expressed through its OUTPUT f = act(pre)
__device__ __forceinline__ float act_slope(float f, int act) {
  if (act == SR3D_ACT_RELU) return f > 0.f ? 1.f : 0.f;
  if (act == SR3D_ACT_LRELU) return f > 0.f ? 1.f : 0.01f;
  return 1.f;
}

// max |.| of what a thread wrote -> one atomicMax per wave into slot[blockIdx & 63] (bits of a non-negative float order
// like unsigned integers).  The split-f16 weight gradient needs max |dY| of its operands (sr3d_hwgrad.hip); taking it here,
// where every element passes through registers anyway, replaces a separate sweep of the tensor.
__device__ __forceinline__ void export_absmax(float m, unsigned* slots) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(slots + (blockIdx.x & 63), __float_as_uint(m));
}

// d_feat = dy * s * act'(f);  d_gate = dy * f * s * (1 - s)        (T: storage type, float or bf16raw; fp32 arithmetic)
// FROM_Y: the second operand is the layer's OUTPUT y = s * act(f) instead of act(f) (SR3D_ACT_FROM_Y): s > 0, so y has
// the sign of act(f) -- the same act' -- and f * s = y: d_gate = dy * y * (1 - s).  The forward then stores no act(f) at all.
// dy2 (may be null): a SECOND incoming gradient of the same output -- the skip connection's -- added on the fly (in fp32),
// instead of a separate elementwise add of the two gradient tensors in front of this pass (3 tensor passes less).
template <typename T, bool FROM_Y>
__global__ __launch_bounds__(kThreads) void gated_act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ f,
                                                                 const T* __restrict__ s, T* __restrict__ df,
                                                                 T* __restrict__ dg, long long n, int act, unsigned* amax) {
  float m1 = 0.f, m2 = 0.f;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = i0; i < n4; i += stride) {
    f32x4 a = ActIo<T>::ld4(dy + 4 * i);
    const f32x4 ff = ActIo<T>::ld4(f + 4 * i), ss = ActIo<T>::ld4(s + 4 * i);
    if (dy2 != nullptr) a += ActIo<T>::ld4(dy2 + 4 * i);   // (kernel argument: uniform)
    f32x4 o1, o2;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      o1[q] = a[q] * ss[q] * act_slope(ff[q], act);
      o2[q] = FROM_Y ? a[q] * ff[q] * (1.f - ss[q]) : a[q] * ff[q] * (ss[q] * (1.f - ss[q]));
      m1 = fmaxf(m1, fabsf(o1[q])), m2 = fmaxf(m2, fabsf(o2[q]));
    }
    ActIo<T>::st4(df + 4 * i, o1);
    ActIo<T>::st4(dg + 4 * i, o2);
  }
  for (long long i = n4 * 4 + i0; i < n; i += stride) {
    const float a = ActIo<T>::ld(dy + i) + (dy2 != nullptr ? ActIo<T>::ld(dy2 + i) : 0.f), ff = ActIo<T>::ld(f + i), ss = ActIo<T>::ld(s + i);
    const float v1 = a * ss * act_slope(ff, act), v2 = FROM_Y ? a * ff * (1.f - ss) : a * ff * (ss * (1.f - ss));
    m1 = fmaxf(m1, fabsf(v1)), m2 = fmaxf(m2, fabsf(v2));
    ActIo<T>::st(df + i, v1);
    ActIo<T>::st(dg + i, v2);
  }
  if (amax != nullptr) {
    export_absmax(m1, amax);
    export_absmax(m2, amax + 64);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void lrelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                             T* __restrict__ dp, long long n, unsigned* amax) {
  float m = 0.f;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = i0; i < n4; i += stride) {
    const f32x4 a = ActIo<T>::ld4(dy + 4 * i), yy = ActIo<T>::ld4(y + 4 * i);
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      o[q] = yy[q] > 0.f ? a[q] : 0.01f * a[q];
      m = fmaxf(m, fabsf(o[q]));
    }
    ActIo<T>::st4(dp + 4 * i, o);
  }
  for (long long i = n4 * 4 + i0; i < n; i += stride) {
    const float v = ActIo<T>::ld(y + i) > 0.f ? ActIo<T>::ld(dy + i) : 0.01f * ActIo<T>::ld(dy + i);
    m = fmaxf(m, fabsf(v));
    ActIo<T>::st(dp + i, v);
  }
  if (amax != nullptr) export_absmax(m, amax);
}

// dpre[b][f*C + c][z][y][x] = dy[b][c][2z+fz][2y+fy][2x+fx] * lrelu'(y[same])
// one thread per (b, c, fine z, fine y, coarse x): reads a float2 (fx = 0, 1), writes two channels
template <typename T>
__global__ __launch_bounds__(kThreads) void unshuffle_lrelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                                       T* __restrict__ dp, int B, int C, int Z,
                                                                       int Y, int X, unsigned* amax) {
  float m = 0.f;
  const long long total = (long long)B * C * (2 * Z) * (2 * Y) * X;
  const long long czyx = (long long)Z * Y * X;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int x = (int)(r % X);
    r /= X;
    const int fy_ = (int)(r % (2 * Y));
    r /= 2 * Y;
    const int fz_ = (int)(r % (2 * Z));
    r /= 2 * Z;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    float2 g, v;
    if constexpr (sizeof(T) == 4) {
      g = reinterpret_cast<const float2*>(dy)[i];
      v = reinterpret_cast<const float2*>(y)[i];
    } else {   // two adjacent bf16: one 4-byte load each
      const unsigned gu = reinterpret_cast<const unsigned*>(dy)[i], vu = reinterpret_cast<const unsigned*>(y)[i];
      g = float2{__builtin_bit_cast(float, gu << 16), __builtin_bit_cast(float, gu & 0xffff0000u)};
      v = float2{__builtin_bit_cast(float, vu << 16), __builtin_bit_cast(float, vu & 0xffff0000u)};
    }
    const int fz = fz_ & 1, z = fz_ >> 1, fy = fy_ & 1, yy = fy_ >> 1;
    const int f0 = (fz * 2 + fy) * 2;
    const long long o = (((long long)b * 8 * C + (long long)f0 * C + c) * Z + z) * Y * X + (long long)yy * X + x;
    const float o0 = v.x > 0.f ? g.x : 0.01f * g.x, o1 = v.y > 0.f ? g.y : 0.01f * g.y;
    m = fmaxf(m, fmaxf(fabsf(o0), fabsf(o1)));
    ActIo<T>::st(dp + o, o0);
    ActIo<T>::st(dp + o + (long long)C * czyx, o1);
  }
  if (amax != nullptr) export_absmax(m, amax);
}

// x0[b][c][z][y][x] = c < C ? x[b][c][z/s][y/s][x/s] : mask[b][0][z][y][x]
__global__ __launch_bounds__(kThreads) void upsample_cat_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ m, float* __restrict__ out,
                                                                int B, int C, int Z, int Y, int X, int s) {
  const long long total = (long long)B * (C + 1) * Z * Y * X;
  const int zl = Z / s, yl = Y / s, xl = X / s;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int xx = (int)(r % X);
    r /= X;
    const int yy = (int)(r % Y);
    r /= Y;
    const int zz = (int)(r % Z);
    r /= Z;
    const int c = (int)(r % (C + 1));
    const int b = (int)(r / (C + 1));
    float v;
    if (c < C)
      v = x[((((long long)b * C + c) * zl + zz / s) * yl + yy / s) * xl + xx / s];
    else
      v = m[(((long long)b * Z + zz) * Y + yy) * X + xx];
    out[i] = v;
  }
}

__global__ __launch_bounds__(kThreads) void avgpool2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int B, int Z, int Y, int X) {
  const int oz = Z / 2, oy = Y / 2, ox = X / 2;
  const long long total = (long long)B * oz * oy * ox;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int xx = (int)(r % ox);
    r /= ox;
    const int yy = (int)(r % oy);
    r /= oy;
    const int zz = (int)(r % oz);
    const int b = (int)(r / oz);
    const float* p = in + (((long long)b * Z + 2 * zz) * Y + 2 * yy) * X + 2 * xx;
    const long long sy = X, sz = (long long)Y * X;
    float s = 0.f;
    s += p[0] + p[1];
    s += p[sy] + p[sy + 1];
    s += p[sz] + p[sz + 1];
    s += p[sz + sy] + p[sz + sy + 1];
    out[i] = s * 0.125f;
  }
}

// near = 1[ (sum over the zero-padded 3x3x3 box of (1 - b) > 0) * b > 0 ]
__global__ __launch_bounds__(kThreads) void near_wall_kernel(const float* __restrict__ m, float* __restrict__ near,
                                                             int B, int Z, int Y, int X) {
  const long long total = (long long)B * Z * Y * X;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int xx = (int)(r % X);
    r /= X;
    const int yy = (int)(r % Y);
    r /= Y;
    const int zz = (int)(r % Z);
    const int b = (int)(r / Z);
    const float* p = m + (long long)b * Z * Y * X;
    float s = 0.f;
    for (int dz = -1; dz <= 1; dz++)
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          const int z = zz + dz, y = yy + dy, x = xx + dx;
          if ((unsigned)z < (unsigned)Z && (unsigned)y < (unsigned)Y && (unsigned)x < (unsigned)X)
            s += 1.f - p[((long long)z * Y + y) * X + x];
        }
    const float filt = s > 0.f ? 1.f : 0.f;
    near[i] = (filt * m[i] > 0.f) ? 1.f : 0.f;
  }
}

// torch.optim.Adam (single-tensor path): exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2*v + (1-b2) g*g;
// denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m / denom
__global__ __launch_bounds__(kThreads) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long long n,
                                                        float step_size, float inv_bc2_sqrt, float w1, float b2,
                                                        float w2, float eps, float gscale) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = i0; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define ONE(q)                                              \
  {                                                         \
    const float gr = gg.q * gscale;                         \
    mm.q = mm.q + w1 * (gr - mm.q);                         \
    vv.q = vv.q * b2 + w2 * gr * gr;                        \
    const float den = sqrtf(vv.q) * inv_bc2_sqrt + eps;     \
    pp.q = pp.q - step_size * (mm.q / den);                 \
  }
    ONE(x) ONE(y) ONE(z) ONE(w)
#undef ONE
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (long long i = n4 * 4 + i0; i < n; i += stride) {
    const float gr = g[i] * gscale;
    const float mm = m[i] + w1 * (gr - m[i]);
    const float vv = v[i] * b2 + w2 * gr * gr;
    const float den = sqrtf(vv) * inv_bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mm / den);
    m[i] = mm;
    v[i] = vv;
  }
}

// Sample preprocessing of the reference's Dataset (pytorch/src/dataset.py:139-161,174,191-195) on the device:
// out = nan_to_num(clamp?((scale * x - mean[c]) / std[c], 0, 1), nan = nan_value), then z levels < discard_z filled.
// Same fp32 operations in the same order as the reference's torch expressions (IEEE subtract and divide), so the
// result is bit-identical to the CPU path.
struct PrepParams {
  const float* x;
  float* out;
  long long total, vox;   // elements; voxels per channel
  int C, YX, discard_z, clip;
  float mean[8], stdv[8];
  float scale, nan_value;
};

__global__ __launch_bounds__(kThreads) void preprocess_kernel(const PrepParams q) {
  // hipcc contracts a * b - c into one fma by default (and __fmul_rn / __fsub_rn are plain operators in HIP, no
  // barrier against it); torch rounds the product first, so contraction is switched off for this function
#pragma clang fp contract(off)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < q.total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long ch = i / q.vox;
    const int c = (int)(ch % q.C);
    const int z = (int)((i - ch * q.vox) / q.YX);
    const float scaled = q.scale * q.x[i];   // plain operators: the pragma above governs THESE (the __f*_rn wrappers
    float v = (scaled - q.mean[c]) / q.stdv[c];   // are header functions compiled with contraction allowed)
    if (q.clip) v = v != v ? v : fminf(fmaxf(v, 0.f), 1.f);               // torch.clamp keeps NaN
    if (v != v) v = q.nan_value;                                          // torch.nan_to_num(nan = nan_value)
    else if (isinf(v)) v = v > 0.f ? 3.402823466e+38f : -3.402823466e+38f;   // ... and its +-inf defaults
    if (z < q.discard_z) v = q.nan_value;
    q.out[i] = v;
  }
}

// ---- PartialConv3d (reference model/custom_conv.py:129-234) around the engine's convolution --------------------
// mask statistics of every output voxel: s = sum over the (zero-padded) 3x3x3 window and the mask channels,
// update = clamp(s, 0, 1), ratio = slide_winsize / (s + 1e-8) * update            (custom_conv.py:203-216)
__global__ __launch_bounds__(kThreads) void pconv_mask_kernel(const float* __restrict__ m, float* __restrict__ upd,
                                                              float* __restrict__ ratio, int Bm, int Cm, int Z, int Y,
                                                              int X, int stride, float winsize) {
  const int oz_ = (Z - 1) / stride + 1, oy_ = (Y - 1) / stride + 1, ox_ = (X - 1) / stride + 1;
  const long long total = (long long)Bm * oz_ * oy_ * ox_;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int ox = (int)(r % ox_);
    r /= ox_;
    const int oy = (int)(r % oy_);
    r /= oy_;
    const int oz = (int)(r % oz_);
    const int b = (int)(r / oz_);
    float sacc = 0.f;
    for (int c = 0; c < Cm; c++) {
      const float* pm = m + ((long long)b * Cm + c) * Z * Y * X;
      for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
          for (int dx = -1; dx <= 1; dx++) {
            const int z = oz * stride + dz, y = oy * stride + dy, x = ox * stride + dx;
            if ((unsigned)z < (unsigned)Z && (unsigned)y < (unsigned)Y && (unsigned)x < (unsigned)X)
              sacc += pm[((long long)z * Y + y) * X + x];
          }
    }
    const float u = fminf(fmaxf(sacc, 0.f), 1.f);
    upd[i] = u;
    ratio[i] = winsize / (sacc + 1e-8f) * u;
  }
}

// out[b][c][v] = x[b][c][v] * m[b % Bm][c % Cm][v]   (mask broadcast over batch and / or channels)
__global__ __launch_bounds__(kThreads) void mul_mask_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                            float* __restrict__ out, int B, int C, long long vox, int Bm,
                                                            int Cm) {
  const long long total = (long long)B * C * vox;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long ch = i / vox, v = i - ch * vox;
    const int b = (int)(ch / C), c = (int)(ch % C);
    out[i] = x[i] * m[((long long)(Bm == 1 ? 0 : b) * Cm + (Cm == 1 ? 0 : c)) * vox + v];
  }
}

// forward : out = bias ? ((raw - bias[c]) * ratio + bias[c]) * upd : raw * ratio          (custom_conv.py:224-229)
// backward: d_raw = dy * ratio * (bias ? upd : 1);  t = dy * upd * (1 - ratio)  (its per-channel sum is d bias)
__global__ __launch_bounds__(kThreads) void pconv_scale_kernel(const float* __restrict__ in, const float* __restrict__ bias,
                                                               const float* __restrict__ upd, const float* __restrict__ ratio,
                                                               float* __restrict__ out, float* __restrict__ tb, int B, int C,
                                                               long long vox, int Bm, int has_bias, int backward) {
  const long long total = (long long)B * C * vox;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long ch = i / vox, v = i - ch * vox;
    const int b = (int)(ch / C), c = (int)(ch % C);
    const long long mi = (long long)(Bm == 1 ? 0 : b) * vox + v;
    const float u = upd[mi], rt = ratio[mi], a = in[i];
    if (!backward) {
      out[i] = has_bias ? ((a - bias[c]) * rt + bias[c]) * u : a * rt;
    } else {
      out[i] = has_bias ? a * rt * u : a * rt;
      if (tb) tb[i] = a * u * (1.f - rt);
    }
  }
}

// Step bookkeeping on the DEVICE for a captured (hipGraph) training step: the host cannot advance a counter between
// replays, so one thread increments it and derives the two bias-correction scalars exactly as the host path does
// (double precision, then rounded to float).
__global__ void adam_prepare_kernel(int* __restrict__ step, float* __restrict__ scalars, double lr, double beta1,
                                    double beta2) {
  const int t = *step + 1;
  *step = t;
  const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
  scalars[0] = (float)(lr / bc1);
  scalars[1] = (float)(1.0 / sqrt(bc2));
}

__global__ __launch_bounds__(kThreads) void adam_indirect_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                 float* __restrict__ m, float* __restrict__ v, long long n,
                                                                 const float* __restrict__ scalars, float w1, float b2,
                                                                 float w2, float eps, float gscale) {
  const float step_size = scalars[0], inv_bc2_sqrt = scalars[1];
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = i0; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define ONE(q)                                              \
  {                                                         \
    const float gr = gg.q * gscale;                         \
    mm.q = mm.q + w1 * (gr - mm.q);                         \
    vv.q = vv.q * b2 + w2 * gr * gr;                        \
    const float den = sqrtf(vv.q) * inv_bc2_sqrt + eps;     \
    pp.q = pp.q - step_size * (mm.q / den);                 \
  }
    ONE(x) ONE(y) ONE(z) ONE(w)
#undef ONE
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (long long i = n4 * 4 + i0; i < n; i += stride) {
    const float gr = g[i] * gscale;
    const float mm = m[i] + w1 * (gr - m[i]);
    const float vv = v[i] * b2 + w2 * gr * gr;
    const float den = sqrtf(vv) * inv_bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mm / den);
    m[i] = mm;
    v[i] = vv;
  }
}

}  // namespace

#define SR3D_ALIGN_CHECK(ptr, what) \
  SR3D_CHECK((reinterpret_cast<uintptr_t>(ptr) & 15) == 0, SR3D_E_ARG, what ": pointer must be 16-byte aligned")

extern "C" {

#define SR3D_DTYPE_CHECK(dtype, what) \
  SR3D_CHECK((dtype) == SR3D_DTYPE_F32 || (dtype) == SR3D_DTYPE_BF16, SR3D_E_ARG, what ": unknown dtype %d", (dtype))

int sr3d_gated_act_bwd(const void* dy, const void* save_f, const void* save_s, void* d_feat, void* d_gate,
                       long long n, int act, int dtype, void* absmax_out, void* stream) {
  return sr3d_gated_act_bwd_sum(dy, nullptr, save_f, save_s, d_feat, d_gate, n, act, dtype, absmax_out, stream);
}

int sr3d_gated_act_bwd_sum(const void* dy, const void* dy2, const void* save_f, const void* save_s, void* d_feat, void* d_gate,
                           long long n, int act, int dtype, void* absmax_out, void* stream) {
  SR3D_DTYPE_CHECK(dtype, "gated_act_bwd");
  SR3D_CHECK(dy && save_f && save_s && d_feat && d_gate && n > 0, SR3D_E_ARG, "gated_act_bwd: bad argument");
  const bool from_y = (act & SR3D_ACT_FROM_Y) != 0;
  act &= ~SR3D_ACT_FROM_Y;
  SR3D_CHECK(act >= 0 && act <= 2, SR3D_E_ARG, "gated_act_bwd: unknown activation %d", act);
  SR3D_ALIGN_CHECK(dy, "gated_act_bwd");
  if (dy2 != nullptr) SR3D_ALIGN_CHECK(dy2, "gated_act_bwd");
  SR3D_ALIGN_CHECK(save_f, "gated_act_bwd");
  SR3D_ALIGN_CHECK(save_s, "gated_act_bwd");
  SR3D_ALIGN_CHECK(d_feat, "gated_act_bwd");
  SR3D_ALIGN_CHECK(d_gate, "gated_act_bwd");
  const double esz = dtype == SR3D_DTYPE_BF16 ? 2.0 : 4.0;
  SrProfScope prof(SR3D_PROF_ACT_BWD, (dy2 ? 6.0 : 5.0) * esz * (double)n, (hipStream_t)stream);   // 3 (4) reads + 2 writes
  const dim3 grid(blocks_for(n, 4)), block(kThreads);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SR3D_DTYPE_BF16) {
    const bf16raw *a = (const bf16raw*)dy, *a2 = (const bf16raw*)dy2, *f = (const bf16raw*)save_f, *sg = (const bf16raw*)save_s;
    if (from_y)
      hipLaunchKernelGGL((gated_act_bwd_kernel<bf16raw, true>), grid, block, 0, st, a, a2, f, sg, (bf16raw*)d_feat, (bf16raw*)d_gate, n, act, (unsigned*)nullptr);
    else
      hipLaunchKernelGGL((gated_act_bwd_kernel<bf16raw, false>), grid, block, 0, st, a, a2, f, sg, (bf16raw*)d_feat, (bf16raw*)d_gate, n, act, (unsigned*)nullptr);
  } else {
    const float *a = (const float*)dy, *a2 = (const float*)dy2, *f = (const float*)save_f, *sg = (const float*)save_s;
    if (from_y)
      hipLaunchKernelGGL((gated_act_bwd_kernel<float, true>), grid, block, 0, st, a, a2, f, sg, (float*)d_feat, (float*)d_gate, n, act, (unsigned*)absmax_out);
    else
      hipLaunchKernelGGL((gated_act_bwd_kernel<float, false>), grid, block, 0, st, a, a2, f, sg, (float*)d_feat, (float*)d_gate, n, act, (unsigned*)absmax_out);
  }
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_lrelu_bwd(const void* dy, const void* y, void* dpre, long long n, int dtype, void* absmax_out, void* stream) {
  SR3D_CHECK(dy && y && dpre && n > 0, SR3D_E_ARG, "lrelu_bwd: bad argument");
  SR3D_DTYPE_CHECK(dtype, "lrelu_bwd");
  SR3D_ALIGN_CHECK(dy, "lrelu_bwd");
  SR3D_ALIGN_CHECK(y, "lrelu_bwd");
  SR3D_ALIGN_CHECK(dpre, "lrelu_bwd");
  SrProfScope prof(SR3D_PROF_ACT_BWD, (dtype == SR3D_DTYPE_BF16 ? 6.0 : 12.0) * (double)n, (hipStream_t)stream);
  if (dtype == SR3D_DTYPE_BF16)
    hipLaunchKernelGGL(lrelu_bwd_kernel<bf16raw>, dim3(blocks_for(n, 4)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const bf16raw*)dy, (const bf16raw*)y, (bf16raw*)dpre, n, (unsigned*)nullptr);
  else
    hipLaunchKernelGGL(lrelu_bwd_kernel<float>, dim3(blocks_for(n, 4)), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float*)dy, (const float*)y, (float*)dpre, n, (unsigned*)absmax_out);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_unshuffle_lrelu_bwd(const void* dy, const void* y, void* dpre, int B, int C, int Z, int Y, int X, int dtype,
                             void* absmax_out, void* stream) {
  SR3D_CHECK(dy && y && dpre && B > 0 && C > 0 && Z > 0 && Y > 0 && X > 0, SR3D_E_ARG,
             "unshuffle_lrelu_bwd: bad argument");
  SR3D_DTYPE_CHECK(dtype, "unshuffle_lrelu_bwd");
  SR3D_CHECK((reinterpret_cast<uintptr_t>(dy) & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0, SR3D_E_ARG,
             "unshuffle_lrelu_bwd: pointers must be 8-byte aligned");
  const long long total = (long long)B * C * 4 * Z * Y * X;
  SrProfScope prof(SR3D_PROF_ACT_BWD, (dtype == SR3D_DTYPE_BF16 ? 6.0 : 12.0) * 2.0 * (double)total, (hipStream_t)stream);
  if (dtype == SR3D_DTYPE_BF16)
    hipLaunchKernelGGL(unshuffle_lrelu_bwd_kernel<bf16raw>, dim3(blocks_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const bf16raw*)dy, (const bf16raw*)y, (bf16raw*)dpre, B, C, Z, Y, X, (unsigned*)nullptr);
  else
    hipLaunchKernelGGL(unshuffle_lrelu_bwd_kernel<float>, dim3(blocks_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float*)dy, (const float*)y, (float*)dpre, B, C, Z, Y, X, (unsigned*)absmax_out);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_upsample_cat(const void* x, const void* b, void* x0, int B, int C, int Z, int Y, int X, int scale,
                      void* stream) {
  SR3D_CHECK(x && b && x0 && B > 0 && C > 0 && Z > 0 && Y > 0 && X > 0 && scale >= 1, SR3D_E_ARG,
             "upsample_cat: bad argument");
  SR3D_CHECK(Z % scale == 0 && Y % scale == 0 && X % scale == 0, SR3D_E_ARG,
             "upsample_cat: grid (%d,%d,%d) is not a multiple of the scale %d", Z, Y, X, scale);
  const long long total = (long long)B * (C + 1) * Z * Y * X;
  SrProfScope prof(SR3D_PROF_DATA, 4.0 * ((double)total + (double)total / (C + 1) * (1.0 + C / (double)(scale * scale * scale))), (hipStream_t)stream);
  hipLaunchKernelGGL(upsample_cat_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float*)x, (const float*)b, (float*)x0, B, C, Z, Y, X, scale);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_avgpool2(const void* in, void* out, int B, int Z, int Y, int X, void* stream) {
  SR3D_CHECK(in && out && B > 0 && Z >= 2 && Y >= 2 && X >= 2, SR3D_E_ARG, "avgpool2: bad argument");
  const long long total = (long long)B * (Z / 2) * (Y / 2) * (X / 2);
  SrProfScope prof(SR3D_PROF_DATA, 4.0 * 9.0 * (double)total, (hipStream_t)stream);
  hipLaunchKernelGGL(avgpool2_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float*)in, (float*)out, B, Z, Y, X);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_near_wall(const void* b, void* near, int B, int Z, int Y, int X, void* stream) {
  SR3D_CHECK(b && near && B > 0 && Z > 0 && Y > 0 && X > 0, SR3D_E_ARG, "near_wall: bad argument");
  const long long total = (long long)B * Z * Y * X;
  SrProfScope prof(SR3D_PROF_DATA, 8.0 * (double)total, (hipStream_t)stream);
  hipLaunchKernelGGL(near_wall_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float*)b, (float*)near, B, Z, Y, X);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_adam_step(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, long long n, double lr, double beta1,
                   double beta2, double eps, int step, double grad_scale, void* stream) {
  SR3D_CHECK(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, SR3D_E_ARG, "adam_step: bad argument");
  SR3D_ALIGN_CHECK(param, "adam_step");
  SR3D_ALIGN_CHECK(grad, "adam_step");
  SR3D_ALIGN_CHECK(exp_avg, "adam_step");
  SR3D_ALIGN_CHECK(exp_avg_sq, "adam_step");
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  const float step_size = (float)(lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  SrProfScope prof(SR3D_PROF_ADAM, 28.0 * (double)n, (hipStream_t)stream);   // p, m, v read+write, g read
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n, 4)), dim3(kThreads), 0, (hipStream_t)stream, (float*)param,
                     (const float*)grad, (float*)exp_avg, (float*)exp_avg_sq, n, step_size, inv_bc2_sqrt,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)grad_scale);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_pconv_mask_update(const void* mask, int Bm, int Cm, int Z, int Y, int X, int stride, float slide_winsize,
                           void* update_mask, void* mask_ratio, void* stream) {
  SR3D_CHECK(mask && update_mask && mask_ratio && Bm > 0 && Cm > 0 && Z > 0 && Y > 0 && X > 0, SR3D_E_ARG,
             "pconv_mask_update: bad argument");
  SR3D_CHECK(stride == 1 || stride == 2, SR3D_E_ARG, "pconv_mask_update: stride must be 1 or 2");
  const long long total = (long long)Bm * ((Z - 1) / stride + 1) * ((Y - 1) / stride + 1) * ((X - 1) / stride + 1);
  hipLaunchKernelGGL(pconv_mask_kernel, dim3(blocks_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                     (const float*)mask, (float*)update_mask, (float*)mask_ratio, Bm, Cm, Z, Y, X, stride, slide_winsize);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_mul_mask(const void* x, const void* mask, void* out, int B, int C, long long voxels, int Bm, int Cm,
                  void* stream) {
  SR3D_CHECK(x && mask && out && B > 0 && C > 0 && voxels > 0, SR3D_E_ARG, "mul_mask: bad argument");
  SR3D_CHECK((Bm == 1 || Bm == B) && (Cm == 1 || Cm == C), SR3D_E_ARG, "mul_mask: mask must broadcast to the input");
  hipLaunchKernelGGL(mul_mask_kernel, dim3(blocks_for((long long)B * C * voxels)), dim3(kThreads), 0,
                     (hipStream_t)stream, (const float*)x, (const float*)mask, (float*)out, B, C, voxels, Bm, Cm);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_pconv_scale(const void* in, const void* bias, const void* update_mask, const void* mask_ratio, void* out,
                     void* bias_terms, int B, int C, long long voxels, int Bm, int backward, void* stream) {
  SR3D_CHECK(in && update_mask && mask_ratio && out && B > 0 && C > 0 && voxels > 0, SR3D_E_ARG,
             "pconv_scale: bad argument");
  SR3D_CHECK(Bm == 1 || Bm == B, SR3D_E_ARG, "pconv_scale: mask batch must be 1 or B");
  SR3D_CHECK(!bias_terms || (backward && bias), SR3D_E_ARG, "pconv_scale: bias_terms only in the backward of a biased conv");
  hipLaunchKernelGGL(pconv_scale_kernel, dim3(blocks_for((long long)B * C * voxels)), dim3(kThreads), 0,
                     (hipStream_t)stream, (const float*)in, (const float*)bias, (const float*)update_mask,
                     (const float*)mask_ratio, (float*)out, (float*)bias_terms, B, C, voxels, Bm, bias != nullptr,
                     backward != 0);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_preprocess(const void* x, void* out, int B, int C, int Z, int Y, int X, const float* means, const float* stds,
                    float scaling, int clip, float nan_value, int discard_z, void* stream) {
  SR3D_CHECK(x && out && means && stds && B > 0 && C > 0 && C <= 8 && Z > 0 && Y > 0 && X > 0, SR3D_E_ARG,
             "preprocess: bad argument (1..8 channels)");
  SR3D_CHECK(discard_z >= 0 && discard_z <= Z, SR3D_E_ARG, "preprocess: discard_z outside the grid");
  PrepParams q{};
  q.x = (const float*)x, q.out = (float*)out;
  q.vox = (long long)Z * Y * X, q.total = (long long)B * C * q.vox;
  q.C = C, q.YX = Y * X, q.discard_z = discard_z, q.clip = clip != 0;
  for (int c = 0; c < C; c++) {
    SR3D_CHECK(stds[c] != 0.f, SR3D_E_ARG, "preprocess: stds[%d] is zero", c);
    q.mean[c] = means[c], q.stdv[c] = stds[c];
  }
  q.scale = scaling, q.nan_value = nan_value;
  SrProfScope prof(SR3D_PROF_DATA, 8.0 * (double)q.total, (hipStream_t)stream);
  hipLaunchKernelGGL(preprocess_kernel, dim3(blocks_for(q.total)), dim3(kThreads), 0, (hipStream_t)stream, q);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_adam_step_device_counter(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, long long n, double lr,
                                  double beta1, double beta2, double eps, void* step_counter, void* scalars,
                                  double grad_scale, void* stream) {
  SR3D_CHECK(param && grad && exp_avg && exp_avg_sq && step_counter && scalars && n > 0, SR3D_E_ARG,
             "adam_step_device_counter: bad argument");
  SR3D_ALIGN_CHECK(param, "adam_step_device_counter");
  SR3D_ALIGN_CHECK(grad, "adam_step_device_counter");
  SR3D_ALIGN_CHECK(exp_avg, "adam_step_device_counter");
  SR3D_ALIGN_CHECK(exp_avg_sq, "adam_step_device_counter");
  SrProfScope prof(SR3D_PROF_ADAM, 28.0 * (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)step_counter, (float*)scalars,
                     lr, beta1, beta2);
  SR3D_HIP(hipGetLastError());
  hipLaunchKernelGGL(adam_indirect_kernel, dim3(blocks_for(n, 4)), dim3(kThreads), 0, (hipStream_t)stream, (float*)param,
                     (const float*)grad, (float*)exp_avg, (float*)exp_avg_sq, n, (const float*)scalars,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)grad_scale);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

}  // extern "C"
