// Weight gradient of a 3x3x3 stride-1 convolution when ONE side has only a handful of channels (VALU kernel).
//
// The Winograd weight gradient (sr3d_wino_wgrad.hip) works on blocks of 32 input channels; layers with 128 + 1 or
// 192 + 2 input channels (a building mask concatenated to the features) would spend a whole block on the 1-2 odd
// channels.  Those channels take this kernel instead:
//
//   dW[n][c][tap] = sum_v dy[n][v] * x[c][v + tap - 1]          n: MANY rows (one per lane), c: FEW channels
//
// A lane owns one row n and keeps FEW x 27 sums in registers; it streams its own dy row with 16-byte loads while
// the x window of the few channels sits in LDS (zero padded halo tile) and is read as BROADCASTS (all lanes read
// the same address: one LDS cycle), 2 LDS reads per 12 FMAs.  No MFMA: with 1-2 columns the matrix pipe would
// run at 3-6 % utilisation, the vector ALUs finish sooner.
#include <algorithm>

#include "sr3d_common.h"

namespace {

constexpr int FTZ = 4, FTY = 8, FTX = 32;        // voxel tile: one z plane per wave
constexpr int FHX = 36;                          // halo row pitch (34 used; 16-byte aligned rows)
constexpr int FHP = (FTY + 2) * FHX;             // halo plane
constexpr int FHC = (FTZ + 2) * FHP;             // halo floats per few-channel

struct FewParams {
  ChanCat many, few;
  int M, few_c0, few_n;     // many-side channels; first few-side channel, number of valid ones (<= F)
  int Z, Y, X;
  int ntz, nty, ntx;
  long long ntiles, per_split;
  float* slab;              // [S][Mpad][F * 27]
  int Mpad, vec;
};

// T: storage type of both operands (float, or bf16raw: widened on load, fp32 arithmetic)
template <int F, typename T>
__global__ __launch_bounds__(256, F == 2 ? 2 : 1) void wgrad_few_kernel(const FewParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* halo = lds;                  // [F][6][10][36]
  float* red = lds + F * FHC;         // [64][F * 27]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x, mb = blockIdx.y;
  const int m = mb * 64 + lane;
  const long long ZYX = (long long)p.Z * p.Y * p.X;

  float acc[F][27];
#pragma unroll
  for (int f = 0; f < F; f++)
#pragma unroll
    for (int t = 0; t < 27; t++) acc[f][t] = 0.f;

  const T* mbase0 = nullptr;   // this lane's channel, sample 0
  long long mbs = 0;
  if (m < p.M) {
    const int si = cat_find(p.many, m);
    mbase0 = reinterpret_cast<const T*>(cat_ptr(p.many, si)) + (long long)(m - cat_cbeg(p.many, si)) * ZYX;
    mbs = cat_bstride(p.many, si);
  }

  const long long t_begin = (long long)split * p.per_split;
  const long long t_end = std::min(t_begin + p.per_split, p.ntiles);
  for (long long tile = t_begin; tile < t_end; tile++) {
    long long r = tile;
    const int tix = (int)(r % p.ntx);
    r /= p.ntx;
    const int tiy = (int)(r % p.nty);
    r /= p.nty;
    const int tiz = (int)(r % p.ntz);
    const int b = (int)(r / p.ntz);
    const int z0 = tiz * FTZ, y0 = tiy * FTY, x0 = tix * FTX;
    __syncthreads();   // previous tile's halo is no longer read
    for (int e = tid; e < F * (FTZ + 2) * (FTY + 2) * 34; e += 256) {
      int q = e;
      const int hx = q % 34;
      q /= 34;
      const int hy = q % (FTY + 2);
      q /= (FTY + 2);
      const int hz = q % (FTZ + 2), f = q / (FTZ + 2);
      const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      float v = 0.f;
      if (f < p.few_n && (unsigned)gz < (unsigned)p.Z && (unsigned)gy < (unsigned)p.Y && (unsigned)gx < (unsigned)p.X) {
        const int c = p.few_c0 + f;
        const int si = cat_find(p.few, c);
        v = ActIo<T>::ld(reinterpret_cast<const T*>(cat_ptr(p.few, si)) + (long long)b * cat_bstride(p.few, si) +
                         (long long)(c - cat_cbeg(p.few, si)) * ZYX + ((long long)gz * p.Y + gy) * p.X + gx);
      }
      halo[f * FHC + hz * FHP + hy * FHX + hx] = v;
    }
    __syncthreads();
    const int z = z0 + wave;
    if (z >= p.Z || mbase0 == nullptr) continue;
    const T* mrow0 = mbase0 + (long long)b * mbs + (long long)z * p.Y * p.X;
    for (int y = 0; y < FTY; y++) {
      const int gy = y0 + y;
      if (gy >= p.Y) break;
      const T* mrow = mrow0 + (long long)gy * p.X + x0;
      auto load_m = [&](const int xq) {
        const int gx = x0 + 4 * xq;
        f32x4 v;
        if (p.vec && gx + 3 < p.X) {
          v = ActIo<T>::ld4(mrow + 4 * xq);
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) v[j] = gx + j < p.X ? ActIo<T>::ld(mrow + 4 * xq + j) : 0.f;
        }
        return v;
      };
      auto fma_piece = [&](const f32x4 mv, const int xq) {
        const float* hrow = halo + wave * FHP + y * FHX + 4 * xq;   // wave-uniform: LDS broadcasts
#pragma unroll
        for (int f = 0; f < F; f++)
#pragma unroll
          for (int dz = 0; dz < 3; dz++)
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
              const float* hp = hrow + f * FHC + dz * FHP + dy * FHX;
              const f32x4 h0 = *reinterpret_cast<const f32x4*>(hp);
              const float h4 = hp[4], h5 = hp[5];
              const float h[6] = {h0[0], h0[1], h0[2], h0[3], h4, h5};
#pragma unroll
              for (int dx = 0; dx < 3; dx++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[f][(dz * 3 + dy) * 3 + dx] += mv[j] * h[j + dx];
            }
      };
      if constexpr (F == 1) {
        // 27 sums leave room for the whole row: 8 loads in flight, then 8 x 108 FMAs (measured best)
        f32x4 mv[FTX / 4];
#pragma unroll
        for (int xq = 0; xq < FTX / 4; xq++) mv[xq] = load_m(xq);
#pragma unroll
        for (int xq = 0; xq < FTX / 4; xq++) fma_piece(mv[xq], xq);
      } else {
        // 54+ sums: one piece and its prefetch, two or more waves per SIMD hide the latencies instead
        f32x4 nxt = load_m(0);
#pragma unroll 2
        for (int xq = 0; xq < FTX / 4; xq++) {
          const f32x4 mv = nxt;
          if (xq + 1 < FTX / 4) nxt = load_m(xq + 1);
          fma_piece(mv, xq);
        }
      }
    }
  }

  // ---- the 4 waves' sums, added in a fixed order
  for (int w = 0; w < 4; w++) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int f = 0; f < F; f++)
#pragma unroll
        for (int t = 0; t < 27; t++) {
          float* dst = red + (f * 27 + t) * 64 + lane;
          *dst = w == 0 ? acc[f][t] : *dst + acc[f][t];
        }
    }
  }
  __syncthreads();
  for (int e = tid; e < 64 * F * 27; e += 256) {
    const int l = e & 63, ft = e >> 6;
    p.slab[((long long)split * p.Mpad + mb * 64 + l) * (F * 27) + ft] = red[ft * 64 + l];
  }
}

// dw[m * ldw + (c0 + f) * 27 + t] = sum over the splits.  One wave per output element: lane l adds splits l, l+64, ...
// in order, then a fixed shuffle tree (deterministic); a single thread per element would walk up to 2048 splits
// serially.
__global__ __launch_bounds__(256) void wgrad_few_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                              int S, int M, int Mpad, int F, int few_n, int few_c0,
                                                              long long ldw) {
  const int total = M * F * 27;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= total) return;
  const int mm = e / (F * 27), ft = e - mm * (F * 27), f = ft / 27, t = ft - f * 27;
  if (f >= few_n) return;
  const float* src = slab + (long long)mm * (F * 27) + ft;
  float s = 0.f;
  for (int k = lane; k < S; k += 64) s += src[(long long)k * Mpad * (F * 27)];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) dw[(long long)mm * ldw + (few_c0 + f) * 27 + t] = s;
}

struct FewPlan {
  int F, mblk, Mpad, ntz, nty, ntx, S;
  long long ntiles, per_split;
};

FewPlan few_plan(const sr3d_conv_desc_t* d, int M, int few_n) {
  FewPlan pl;
  pl.F = few_n <= 1 ? 1 : (few_n <= 2 ? 2 : 4);
  pl.mblk = ceil_div(M, 64), pl.Mpad = pl.mblk * 64;
  pl.ntz = ceil_div(d->Z, FTZ), pl.nty = ceil_div(d->Y, FTY), pl.ntx = ceil_div(d->X, FTX);
  pl.ntiles = (long long)d->B * pl.ntz * pl.nty * pl.ntx;
  long long want = std::max(1, 2048 / pl.mblk);
  want = std::min(want, pl.ntiles);
  pl.per_split = (pl.ntiles + want - 1) / want;
  pl.S = (int)((pl.ntiles + pl.per_split - 1) / pl.per_split);
  return pl;
}

template <int F, typename T>
int launch_few(const FewParams& p, const FewPlan& pl, hipStream_t st) {
  constexpr int kLds = (F * FHC + 64 * F * 27) * 4;
  static SrPerDevice setup;   // (the attribute is per device, not per thread)
  if (int rc = setup.once([&]() -> int {
        SR3D_HIP(hipFuncSetAttribute((const void*)wgrad_few_kernel<F, T>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
        return SR3D_OK;
      }))
    return rc;
  hipLaunchKernelGGL((wgrad_few_kernel<F, T>), dim3(pl.S, pl.mblk), dim3(256), kLds, st, p);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

}  // namespace

size_t sr3d_wgrad_few_ws_bytes(const sr3d_conv_desc_t* d, int M, int few_n) {
  const FewPlan pl = few_plan(d, M, few_n);
  return (size_t)pl.S * pl.Mpad * pl.F * 27 * 4;
}

int sr3d_wgrad_few(const sr3d_conv_desc_t* d, const ChanCat& many, int M, const ChanCat& few, int few_c0, int few_n,
                   float* dw, long long ldw, float* ws, hipStream_t st) {
  SR3D_CHECK(few_n >= 1 && few_n <= 4, SR3D_E_ARG, "wgrad_few: 1..4 channels on the small side (got %d)", few_n);
  SR3D_CHECK(d->stride == 1, SR3D_E_ARG, "wgrad_few: stride 1 only");
  const FewPlan pl = few_plan(d, M, few_n);
  SR3D_CHECK(pl.mblk <= 65535, SR3D_E_ARG, "wgrad_few: too many row blocks");
  FewParams p{};
  p.many = many, p.few = few, p.M = M, p.few_c0 = few_c0, p.few_n = few_n;
  p.Z = d->Z, p.Y = d->Y, p.X = d->X;
  p.ntz = pl.ntz, p.nty = pl.nty, p.ntx = pl.ntx, p.ntiles = pl.ntiles, p.per_split = pl.per_split;
  p.slab = ws, p.Mpad = pl.Mpad;
  uintptr_t bits = 0;
  for (int i = 0; i < many.n; i++) bits |= reinterpret_cast<uintptr_t>(many.ptr[i]);
  p.vec = (d->X % 4 == 0) && (bits & 15) == 0;
  int rc;
  if (d->dtype == SR3D_DTYPE_BF16)
    rc = pl.F == 1 ? launch_few<1, bf16raw>(p, pl, st) : (pl.F == 2 ? launch_few<2, bf16raw>(p, pl, st) : launch_few<4, bf16raw>(p, pl, st));
  else
    rc = pl.F == 1 ? launch_few<1, float>(p, pl, st) : (pl.F == 2 ? launch_few<2, float>(p, pl, st) : launch_few<4, float>(p, pl, st));
  if (rc) return rc;
  const int total = M * pl.F * 27;
  hipLaunchKernelGGL(wgrad_few_reduce_kernel, dim3(ceil_div(total, 4)), dim3(256), 0, st, (const float*)ws, dw, pl.S, M,
                     pl.Mpad, pl.F, few_n, few_c0, ldw);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}
