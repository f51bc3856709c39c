// Weight gradient of the 3x3x3 Conv3d on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   dW[n][c][t] = sum_{b, voxel o} dY[b][n][o] * X[b][c][o*stride + d_t]
//
// GEMM view: rows n (A = dY), cols c for one tap (B = X shifted by the tap),
// reduction over voxels (2 per MFMA).  One workgroup owns a 32(n) x 32(c) x 27(tap)
// block of dW in registers -- wave g holds taps 7g..7g+6 -- and walks a
// contiguous range of spatial tiles (z fastest, so consecutive tiles re-read two
// of their three input planes from L2).  The voxel reduction is split over S
// workgroups; partial blocks go to a slab [S][27][Npad][Cpad] that a second,
// fixed-order kernel sums and transposes into PyTorch's (n, c, kz, ky, kx)
// layout, so the result is bit-reproducible (the reference runs with
// use_deterministic=True: utils.py:70-92).
// X and dY are virtual channel concatenations (sr3d_common.h), e.g. dY =
// [d_feat ; d_gate] of a gated layer yields dW = [dWf ; dWg] in one pass.
#include "sr3d_common.h"

namespace {

struct WgradParams {
  ChanCat x;
  ChanCat dy;
  int Cin, N;
  int IZ, IY, IX;
  int OZ, OY, OX;
  int nty, ntx;        // tiles per (y, x); z tiles = OZ
  long long ntiles;    // B * nty * ntx * OZ
  long long per_split; // tiles per workgroup
  float* slab;
  int Npad, Cpad;
};

template <int S_IN, int TY>
struct WgradCfg {
  static constexpr int HZ = 3;
  static constexpr int HY = (TY - 1) * S_IN + 3;
  static constexpr int HX = 31 * S_IN + 3;
  static constexpr int HCH = HZ * HY * HX;
  static constexpr int PH = HCH | 1;      // odd pitch: 32 channels hit 32 banks
  static constexpr int VT = TY * 32;      // voxels per tile
  static constexpr int PV = VT + 1;
  static constexpr int XS = 32 * PH;
  static constexpr size_t lds_bytes = (size_t)(XS + 32 * PV) * 4;
};

template <int S_IN, int TY>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
  using C = WgradCfg<S_IN, TY>;
  constexpr int HY = C::HY, HX = C::HX, HCH = C::HCH, PH = C::PH, PV = C::PV, VT = C::VT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;
  float* Ds = lds + C::XS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, cb = blockIdx.y, nb = blockIdx.z;
  const long long IZYX = (long long)p.IZ * p.IY * p.IX;
  const long long OZYX = (long long)p.OZ * p.OY * p.OX;

  f32x16 acc[7];
#pragma unroll
  for (int t = 0; t < 7; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

  // LDS read bases: A = dY[n = lane&31][v + (lane>>5)], B = X[c = lane&31][pos + (lane>>5)*S + tap]
  const int a_base = (lane & 31) * PV + (lane >> 5);
  int b_base[7];
#pragma unroll
  for (int t = 0; t < 7; t++) {
    int tap = wave * 7 + t;
    tap = tap > 26 ? 26 : tap;  // wave 3 has 6 real taps; the 7th is a dummy that is never stored
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    b_base[t] = (lane & 31) * PH + (lane >> 5) * S_IN + (kz * HY + ky) * HX + kx;
  }

  constexpr int NIX = (HCH + 255) / 256;

  const long long t_begin = (long long)split * p.per_split;
  long long t_end = t_begin + p.per_split;
  if (t_end > p.ntiles) t_end = p.ntiles;

  for (long long tile = t_begin; tile < t_end; tile++) {
    long long r = tile;
    const int oz = (int)(r % p.OZ);
    r /= p.OZ;
    const int tix = (int)(r % p.ntx);
    r /= p.ntx;
    const int tiy = (int)(r % p.nty);
    const int b = (int)(r / p.nty);
    const int oy0 = tiy * TY, ox0 = tix * 32;
    const int gz0 = oz * S_IN - 1, gy0 = oy0 * S_IN - 1, gx0 = ox0 * S_IN - 1;

    __syncthreads();  // previous tile fully consumed
    // ---- X halo [32 c][3][HY][HX]
    {
      int hoff[NIX];
#pragma unroll
      for (int i = 0; i < NIX; i++) {
        const int e = tid + i * 256;
        const int hz = e / (HY * HX);
        const int r2 = e - hz * (HY * HX);
        const int hy = r2 / HX;
        const int hx = r2 - hy * HX;
        const int gz = gz0 + hz, gy = gy0 + hy, gx = gx0 + hx;
        const bool ok = e < HCH && (unsigned)gz < (unsigned)p.IZ && (unsigned)gy < (unsigned)p.IY &&
                        (unsigned)gx < (unsigned)p.IX;
        hoff[i] = ok ? (gz * p.IY + gy) * p.IX + gx : -1;
      }
      for (int c0 = 0; c0 < 32; c0 += 8) {
        float v[8][NIX];
#pragma unroll
        for (int cc = 0; cc < 8; cc++) {
          const int gc = cb * 32 + c0 + cc;
          const float* base = nullptr;
          if (gc < p.Cin) {
            const int si = cat_find(p.x, gc);
            base = p.x.ptr[si] + (long long)b * p.x.bstride[si] + (long long)(gc - p.x.cbeg[si]) * IZYX;
          }
#pragma unroll
          for (int i = 0; i < NIX; i++) v[cc][i] = (base != nullptr && hoff[i] >= 0) ? base[hoff[i]] : 0.f;
        }
#pragma unroll
        for (int cc = 0; cc < 8; cc++)
#pragma unroll
          for (int i = 0; i < NIX; i++)
            if (tid + i * 256 < HCH) Xs[(c0 + cc) * PH + tid + i * 256] = v[cc][i];
      }
    }
    // ---- dY tile [32 n][TY*32 voxels]
    {
      constexpr int PER = (32 * VT) / 256;  // elements per thread
      float v[PER];
#pragma unroll
      for (int i = 0; i < PER; i++) {
        const int e = tid + i * 256;
        const int n = e / VT, vv = e % VT;
        const int oy = oy0 + vv / 32, ox = ox0 + (vv & 31);
        const int gn = nb * 32 + n;
        float val = 0.f;
        if (gn < p.N && oy < p.OY && ox < p.OX) {
          const int si = cat_find(p.dy, gn);
          const float* base = cat_ptr(p.dy, si) + (long long)b * cat_bstride(p.dy, si) +
                              (long long)(gn - cat_cbeg(p.dy, si)) * OZYX;
          val = base[((long long)oz * p.OY + oy) * p.OX + ox];
        }
        v[i] = val;
      }
#pragma unroll
      for (int i = 0; i < PER; i++) {
        const int e = tid + i * 256;
        Ds[(e / VT) * PV + (e % VT)] = v[i];
      }
    }
    __syncthreads();

#pragma unroll
    for (int row = 0; row < TY; row++) {
#pragma unroll 8
      for (int xx = 0; xx < 32; xx += 2) {
        const float a = Ds[a_base + row * 32 + xx];
        float bv[7];
#pragma unroll
        for (int t = 0; t < 7; t++) bv[t] = Xs[b_base[t] + (row * S_IN) * HX + xx * S_IN];
#pragma unroll
        for (int t = 0; t < 7; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc[t], 0, 0, 0);
      }
    }
  }

  // ---- partial block -> slab[split][tap][n][c]
  const int c = cb * 32 + (lane & 31);
#pragma unroll
  for (int t = 0; t < 7; t++) {
    const int tap = wave * 7 + t;
    if (tap > 26) continue;
    float* dst = p.slab + (((long long)split * 27 + tap) * p.Npad + nb * 32) * p.Cpad + c;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      dst[(long long)n * p.Cpad] = acc[t][r];
    }
  }
}

// dW[n][c][t] = sum_s slab[s][t][n][c]; one workgroup per (n, 32 channels)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                          int S, int N, int Cin, int Npad, int Cpad) {
  __shared__ float tile[32 * 27];
  const int n = blockIdx.y, c0 = blockIdx.x * 32;
  const long long plane = (long long)Npad * Cpad;
  for (int e = threadIdx.x; e < 27 * 32; e += 256) {
    const int t = e >> 5, cc = e & 31;
    const float* src = slab + (long long)t * plane + (long long)n * Cpad + c0 + cc;
    float s = 0.f;
    for (int k = 0; k < S; k++) s += src[(long long)k * 27 * plane];
    tile[cc * 27 + t] = s;
  }
  __syncthreads();
  const int nvalid = (Cin - c0 < 32 ? Cin - c0 : 32) * 27;
  float* dst = dw + ((long long)n * Cin + c0) * 27;
  for (int e = threadIdx.x; e < nvalid; e += 256) dst[e] = tile[e];
}

// db[c] = sum dy[b][c][:]  -- two deterministic stages
__global__ __launch_bounds__(256) void bias_partial_kernel(const float* __restrict__ dy, float* __restrict__ part,
                                                          int B, int C, long long vox, int nsplit) {
  const int c = blockIdx.y, sp = blockIdx.x;
  const long long per = (vox + nsplit - 1) / nsplit;
  const long long v0 = sp * per, v1 = (v0 + per < vox ? v0 + per : vox);
  float s = 0.f;
  for (int b = 0; b < B; b++) {
    const float* src = dy + ((long long)b * C + c) * vox;
    for (long long v = v0 + threadIdx.x; v < v1; v += 256) s += src[v];
  }
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(long long)c * nsplit + sp] = red[0];
}

__global__ void bias_final_kernel(const float* __restrict__ part, float* __restrict__ db, int C, int nsplit) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; k++) s += part[(long long)c * nsplit + k];
  db[c] = s;
}

inline int out_dim(int z, int s) { return (z - 1) / s + 1; }

struct Plan {
  int Npad, Cpad, nblk, cblk, ty, nty, ntx;
  long long ntiles, per_split;
  int S;
};

Plan make_plan(const sr3d_conv_desc_t* d, int n_total) {
  Plan pl;
  pl.nblk = ceil_div(n_total, 32), pl.cblk = ceil_div(d->Cin, 32);
  pl.Npad = pl.nblk * 32, pl.Cpad = pl.cblk * 32;
  const int OZ = out_dim(d->Z, d->stride), OY = out_dim(d->Y, d->stride), OX = out_dim(d->X, d->stride);
  pl.ty = d->stride == 1 ? 2 : 1;
  pl.nty = ceil_div(OY, pl.ty), pl.ntx = ceil_div(OX, 32);
  pl.ntiles = (long long)d->B * pl.nty * pl.ntx * OZ;
  // enough workgroups to fill 256 CUs x 2 a few times over, but cap the slab at ~192 MiB
  long long want = ceil_div(2048, pl.nblk * pl.cblk);
  const long long slab_one = (long long)27 * pl.Npad * pl.Cpad * 4;
  const long long cap = (192ll << 20) / slab_one;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  if (want > pl.ntiles) want = pl.ntiles;
  pl.per_split = (pl.ntiles + want - 1) / want;
  pl.S = (int)((pl.ntiles + pl.per_split - 1) / pl.per_split);
  return pl;
}

int bias_splits(long long vox) {
  long long s = vox / 16384;
  return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

}  // namespace

extern "C" {

size_t sr3d_conv3d_bwd_weight_workspace_bytes(const sr3d_conv_desc_t* d, int n_total) {
  if (!d || d->Cin <= 0 || n_total <= 0 || (d->stride != 1 && d->stride != 2)) return 0;
  const Plan pl = make_plan(d, n_total);
  return (size_t)pl.S * 27 * pl.Npad * pl.Cpad * 4;
}

int sr3d_conv3d_bwd_weight(const sr3d_conv_desc_t* d, const sr3d_slice_t* x_srcs, int n_src,
                           const sr3d_slice_t* dy_srcs, int n_dy, void* dw, void* workspace, size_t workspace_bytes,
                           void* stream) {
  SR3D_CHECK(d && dw && workspace, SR3D_E_ARG, "conv3d_bwd_weight: null pointer");
  SR3D_CHECK(d->stride == 1 || d->stride == 2, SR3D_E_ARG, "conv3d_bwd_weight: stride must be 1 or 2");
  SR3D_CHECK((long long)d->Z * d->Y * d->X < (1ll << 31), SR3D_E_ARG, "conv3d_bwd_weight: grid too large");
  int n_total = 0;
  for (int i = 0; i < n_dy && i < SR3D_MAX_SRC; i++) n_total += dy_srcs[i].channels;
  SR3D_CHECK(n_total > 0, SR3D_E_ARG, "conv3d_bwd_weight: dy_srcs hold no channels");
  const Plan pl = make_plan(d, n_total);
  SR3D_CHECK(workspace_bytes >= (size_t)pl.S * 27 * pl.Npad * pl.Cpad * 4, SR3D_E_WORKSPACE,
             "conv3d_bwd_weight: workspace of %zu bytes is too small", workspace_bytes);
  WgradParams p{};
  const int OZ = out_dim(d->Z, d->stride), OY = out_dim(d->Y, d->stride), OX = out_dim(d->X, d->stride);
  if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &p.x, "x_srcs")) return rc;
  if (int rc = sr3d_make_cat(dy_srcs, n_dy, (long long)OZ * OY * OX, n_total, &p.dy, "dy_srcs")) return rc;
  for (int i = 0; i < p.x.n; i++) SR3D_CHECK(p.x.ptr[i], SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
  for (int i = 0; i < p.dy.n; i++) SR3D_CHECK(p.dy.ptr[i], SR3D_E_ARG, "dy_srcs[%d].ptr is null", i);
  p.Cin = d->Cin, p.N = n_total;
  p.IZ = d->Z, p.IY = d->Y, p.IX = d->X;
  p.OZ = OZ, p.OY = OY, p.OX = OX;
  p.nty = pl.nty, p.ntx = pl.ntx, p.ntiles = pl.ntiles, p.per_split = pl.per_split;
  p.slab = (float*)workspace, p.Npad = pl.Npad, p.Cpad = pl.Cpad;
  dim3 grid(pl.S, pl.cblk, pl.nblk);
  SR3D_CHECK(pl.cblk <= 65535 && pl.nblk <= 65535, SR3D_E_ARG, "conv3d_bwd_weight: too many channel blocks");
  hipStream_t st = (hipStream_t)stream;
  constexpr int kLds1 = (int)WgradCfg<1, 2>::lds_bytes;
  constexpr int kLds2 = (int)WgradCfg<2, 1>::lds_bytes;
  void* tok = nullptr;
  if (sr3d_prof_active())
    sr3d_prof_begin(SR3D_PROF_WGRAD, 2.0 * 27 * d->Cin * (double)n_total * (double)OZ * OY * OX * d->B, st, &tok);
  if (d->stride == 1) {
    auto kern = wgrad_kernel<1, 2>;
    static thread_local bool cfg = false;
    if (!cfg) {
      SR3D_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLds1));
      cfg = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), kLds1, st, p);
  } else {
    auto kern = wgrad_kernel<2, 1>;
    static thread_local bool cfg = false;
    if (!cfg) {
      SR3D_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLds2));
      cfg = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), kLds2, st, p);
  }
  sr3d_prof_end(tok, st);
  SR3D_HIP(hipGetLastError());
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(pl.cblk, n_total), dim3(256), 0, st, (const float*)workspace,
                     (float*)dw, pl.S, n_total, d->Cin, pl.Npad, pl.Cpad);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

size_t sr3d_bias_grad_workspace_bytes(int B, int C, long long voxels) {
  (void)B;
  return (size_t)C * bias_splits(voxels) * 4;
}

int sr3d_bias_grad(const void* dy, int B, int C, long long voxels, void* db, void* workspace, void* stream) {
  SR3D_CHECK(dy && db && workspace && B > 0 && C > 0 && voxels > 0, SR3D_E_ARG, "bias_grad: bad argument");
  SR3D_CHECK(C <= 65535, SR3D_E_ARG, "bias_grad: too many channels");
  const int ns = bias_splits(voxels);
  hipLaunchKernelGGL(bias_partial_kernel, dim3(ns, C), dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                     (float*)workspace, B, C, voxels, ns);
  SR3D_HIP(hipGetLastError());
  hipLaunchKernelGGL(bias_final_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, (float*)db, C, ns);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

}  // extern "C"
