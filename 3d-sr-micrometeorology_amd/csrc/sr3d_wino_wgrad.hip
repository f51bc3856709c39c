// Weight gradient of the stride-1 3x3x3 convolutions in the Winograd F(2x2, 3x3) domain (fp32 MFMA).
//
//   dU[kz][xi][n][c] = sum over 2x2 output tiles  dM[xi][n][tile] * V[kz][xi][c][tile]
//   dM = A dY_tile A^T  (2x2 -> 4x4),   V = B^T d_tile B  (4x4 input patch of input plane z + kz - 1)
//   dW[n][c][kz] = G^T dU[kz] G         (4x4 -> 3x3, done by the reduce kernel after the split-K sum)
//
// 16 products per tile, channel pair and kz instead of 36: 2.25x fewer MFMA issues than the direct kernel
// (sr3d_wgrad.hip), same fp32 arithmetic.
//
// One 512-thread workgroup owns a 32(n) x 32(c) block of dU for all 3 x 16 (kz, xi): wave w keeps xi = 2w, 2w+1
// for the three kz (6 accumulators).  The reduction runs over strips of 8 tiles (2 x 16 voxels of one output
// plane), z fastest.  Per strip: the raw rows prefetched during the previous strip go to LDS, 256 threads
// transform the ONE new input plane (c, tile) -> V (the other two planes of the 3-plane window are re-used in
// place, rotating slots), 256 threads transform dY (n, tile) -> dM, then the MFMAs read both as plain
// conflict-free fragments (2 LDS reads per MFMA, no transform work in the MFMA phase).
#include "sr3d_common.h"

namespace {

constexpr int GT = 8;                       // Winograd tiles per strip (1 tile row x 8 tile columns)
constexpr int GXW = 2 * GT + 2;             // raw input columns per row (18)
constexpr int GXP = 4 * GXW + 2;            // raw X pitch per channel (74: even, (c*74) mod 64 distinct evens)
constexpr int GDP = 2 * 2 * GT + 2;         // raw dY pitch per channel (34)
constexpr int GVS = 16 * GT * 32;           // one V plane slot / the dM buffer: [xi][tile][32] floats (4096)
constexpr int kGLdsFloats = 3 * GVS + GVS + 32 * GXP + 32 * GDP;
constexpr size_t kGLds = (size_t)kGLdsFloats * 4;

typedef const __attribute__((address_space(1))) float* gfloat_p;
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WinoWgradParams {
  ChanCat x;
  ChanCat dy;
  int Cin, N;
  int Z, Y, X;
  int nty, ntx;
  long long ntiles, per_split;
  float* slab;   // [S][48][Npad][Cpad]
  int Npad, Cpad;
};

__global__ __launch_bounds__(512, 2) void wino_wgrad_kernel(const WinoWgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Vs = lds;                    // 3 slots [xi][tile][c]
  float* Ms = lds + 3 * GVS;          // [xi][tile][n]
  float* Xr = Ms + GVS;               // raw input rows of the new plane [c][4 rows][18]
  float* Dr = Xr + 32 * GXP;          // raw dY rows [n][2 rows][16]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x, cb = blockIdx.y, nb = blockIdx.z;
  const long long ZYX = (long long)p.Z * p.Y * p.X;

  f32x16 acc[6];   // [kz][xi - 2*wave]
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;

  // ---- prefetch bookkeeping: this thread's raw elements (fixed positions inside a strip)
  // X: 32 c x 4 rows x 18 cols = 2304 -> 5 per thread (the last partly);  dY: 32 n x 2 x 16 = 1024 -> 2 per thread
  constexpr int NXE = (32 * 4 * GXW + 511) / 512;   // 5
  int xe_c[NXE], xe_r[NXE], xe_x[NXE];
#pragma unroll
  for (int i = 0; i < NXE; i++) {
    const int e = tid + i * 512;
    xe_c[i] = e / (4 * GXW);
    const int r2 = e - xe_c[i] * (4 * GXW);
    xe_r[i] = r2 / GXW, xe_x[i] = r2 - xe_r[i] * GXW;
    if (e >= 32 * 4 * GXW) xe_c[i] = -1;
  }
  float px[NXE], pd[2];

  const long long t_begin = (long long)split * p.per_split;
  long long t_end = t_begin + p.per_split;
  if (t_end > p.ntiles) t_end = p.ntiles;
  int n_oz, n_tix, n_tiy, n_b;
  {
    long long r = t_begin;
    n_oz = (int)(r % p.Z);
    r /= p.Z;
    n_tix = (int)(r % p.ntx);
    r /= p.ntx;
    n_tiy = (int)(r % p.nty);
    n_b = (int)(r / p.nty);
  }
  int c_b = 0, c_oz = 0, c_y0 = 0, c_x0 = 0;
  auto prep_next = [&]() {
    c_oz = n_oz, c_b = n_b, c_y0 = n_tiy * 2, c_x0 = n_tix * (2 * GT);
    if (++n_oz == p.Z) {
      n_oz = 0;
      if (++n_tix == p.ntx) {
        n_tix = 0;
        if (++n_tiy == p.nty) n_tiy = 0, ++n_b;
      }
    }
  };
  // raw rows of input plane gz of the strip (c_*) -> px
  auto load_x = [&](const int gz) {
#pragma unroll
    for (int i = 0; i < NXE; i++) {
      float v = 0.f;
      const int gc = cb * 32 + xe_c[i];
      const int gy = c_y0 - 1 + xe_r[i], gx = c_x0 - 1 + xe_x[i];
      if (xe_c[i] >= 0 && gc < p.Cin && (unsigned)gz < (unsigned)p.Z && (unsigned)gy < (unsigned)p.Y &&
          (unsigned)gx < (unsigned)p.X) {
        const int si = cat_find(p.x, gc);
        v = ((gfloat_p)cat_ptr(p.x, si))[(long long)c_b * cat_bstride(p.x, si) + (long long)(gc - cat_cbeg(p.x, si)) * ZYX +
                                          ((long long)gz * p.Y + gy) * p.X + gx];
      }
      px[i] = v;
    }
  };
  auto store_x = [&]() {
#pragma unroll
    for (int i = 0; i < NXE; i++)
      if (xe_c[i] >= 0) Xr[xe_c[i] * GXP + xe_r[i] * GXW + xe_x[i]] = px[i];
  };
  auto load_dy = [&]() {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int e = tid + i * 512;            // 32 n x 32 voxels
      const int n = e >> 5, r = (e >> 4) & 1, xx = e & 15;
      const int gn = nb * 32 + n, gy = c_y0 + r, gx = c_x0 + xx;
      float v = 0.f;
      if (gn < p.N && gy < p.Y && gx < p.X) {
        const int si = cat_find(p.dy, gn);
        v = ((gfloat_p)cat_ptr(p.dy, si))[(long long)c_b * cat_bstride(p.dy, si) + (long long)(gn - cat_cbeg(p.dy, si)) * ZYX +
                                           ((long long)c_oz * p.Y + gy) * p.X + gx];
      }
      pd[i] = v;
    }
  };
  auto store_dy = [&]() {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int e = tid + i * 512;
      Dr[(e >> 5) * GDP + ((e >> 4) & 1) * (2 * GT) + (e & 15)] = pd[i];
    }
  };
  // threads 0..255: (c, tile) patch of the raw plane -> V slot;   [xi][tile][c]
  auto transform_v = [&](float* V) {
    const int c = tid & 31, tl = (tid >> 5) & 7;
    const float* rp = Xr + c * GXP + 2 * tl;
    float d[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const f32x2 a = *reinterpret_cast<const f32x2*>(rp + i * GXW);
      const f32x2 b2 = *reinterpret_cast<const f32x2*>(rp + i * GXW + 2);
      d[i][0] = a.x, d[i][1] = a.y, d[i][2] = b2.x, d[i][3] = b2.y;
    }
    float tt[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      tt[i][0] = d[i][0] - d[i][2];
      tt[i][1] = d[i][1] + d[i][2];
      tt[i][2] = d[i][2] - d[i][1];
      tt[i][3] = d[i][1] - d[i][3];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      V[((0 * 4 + j) * GT + tl) * 32 + c] = tt[0][j] - tt[2][j];
      V[((1 * 4 + j) * GT + tl) * 32 + c] = tt[1][j] + tt[2][j];
      V[((2 * 4 + j) * GT + tl) * 32 + c] = tt[2][j] - tt[1][j];
      V[((3 * 4 + j) * GT + tl) * 32 + c] = tt[1][j] - tt[3][j];
    }
  };
  // threads 256..511: (n, tile) 2x2 block of dY -> dM = A dY A^T with A = [1 0; 1 1; 1 -1; 0 -1]
  auto transform_m = [&]() {
    const int n = tid & 31, tl = (tid >> 5) & 7;
    const float* rp = Dr + n * GDP + 2 * tl;
    const f32x2 r0 = *reinterpret_cast<const f32x2*>(rp);
    const f32x2 r1 = *reinterpret_cast<const f32x2*>(rp + 2 * GT);
    float t0[4] = {r0.x, r0.x + r0.y, r0.x - r0.y, -r0.y};
    float t1[4] = {r1.x, r1.x + r1.y, r1.x - r1.y, -r1.y};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      Ms[((0 * 4 + j) * GT + tl) * 32 + n] = t0[j];
      Ms[((1 * 4 + j) * GT + tl) * 32 + n] = t0[j] + t1[j];
      Ms[((2 * 4 + j) * GT + tl) * 32 + n] = t0[j] - t1[j];
      Ms[((3 * 4 + j) * GT + tl) * 32 + n] = -t1[j];
    }
  };

  bool fresh = true;
  int s0 = 0;   // slot of the plane kz = 0 of the current strip
  if (t_begin < t_end) {
    prep_next();
    load_dy();
  }
  const int fb = (lane >> 5) * 32 + (lane & 31);   // fragment offset inside [tile pair][32]

  for (long long tile = t_begin; tile < t_end; tile++) {
    __syncthreads();   // previous strip's MFMAs are done: raw buffers, dM and the oldest V slot are free
    if (fresh) {
      // start of a z column: planes oz-1 and oz are transformed here, oz+1 joins the normal path below
      s0 = 0;
      for (int k = 0; k < 2; k++) {
        load_x(c_oz - 1 + k);
        store_x();
        __syncthreads();
        if (tid < 256) transform_v(Vs + k * GVS);
        __syncthreads();
      }
      load_x(c_oz + 1);
    } else {
      s0 = (s0 + 1) % 3;
    }
    store_x();    // plane oz+1 (prefetched during the previous strip, or just loaded)
    store_dy();
    __syncthreads();
    if (tid < 256)
      transform_v(Vs + ((s0 + 2) % 3) * GVS);
    else
      transform_m();
    __syncthreads();

    const bool more = tile + 1 < t_end;
    const int cur_oz = c_oz;
    if (more) {
      prep_next();
      fresh = c_oz == 0;
      if (!fresh) load_x(c_oz + 1);   // in flight during the MFMA phase
      load_dy();
    }
    (void)cur_oz;

    const float* v0p = Vs + s0 * GVS + fb;
    const float* v1p = Vs + ((s0 + 1) % 3) * GVS + fb;
    const float* v2p = Vs + ((s0 + 2) % 3) * GVS + fb;
    const float* mp = Ms + fb;
#pragma unroll
    for (int ks = 0; ks < GT / 2; ks++) {
#pragma unroll
      for (int xl = 0; xl < 2; xl++) {
        const int xi = 2 * wave + xl;
        const int o = (xi * GT + 2 * ks) * 32;
        const float a = mp[o];
        const float b0 = v0p[o], b1 = v1p[o], b2 = v2p[o];
        acc[0 * 2 + xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0 * 2 + xl], 0, 0, 0);
        acc[1 * 2 + xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1 * 2 + xl], 0, 0, 0);
        acc[2 * 2 + xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2 * 2 + xl], 0, 0, 0);
      }
    }
  }

  // ---- partial dU block -> slab[split][kz*16 + xi][n][c]
  const int c = cb * 32 + (lane & 31);
#pragma unroll
  for (int kz = 0; kz < 3; kz++)
#pragma unroll
    for (int xl = 0; xl < 2; xl++) {
      const int q = kz * 16 + 2 * wave + xl;
      float* dst = p.slab + (((long long)split * 48 + q) * p.Npad + nb * 32) * p.Cpad + c;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        dst[(long long)n * p.Cpad] = acc[kz * 2 + xl][r];
      }
    }
}

// dW[n][c][kz][ky][kx] = sum_{xi} G[xi_y][ky] G[xi_x][kx] * (sum_s slab[s][kz*16 + xi][n][c])
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                               int S, int N, int Cin, int Npad, int Cpad) {
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  const long long plane = (long long)Npad * Cpad;
  const long long total = (long long)N * Cin * 3;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    // c fastest so that slab reads are coalesced
    const int c = (int)(e % Cin);
    const long long r = e / Cin;
    const int n = (int)(r % N), kz = (int)(r / N);
    float u[16];
#pragma unroll
    for (int xi = 0; xi < 16; xi++) {
      const float* src = slab + (long long)(kz * 16 + xi) * plane + (long long)n * Cpad + c;
      float s = 0.f;
      for (int k = 0; k < S; k++) s += src[(long long)k * 48 * plane];
      u[xi] = s;
    }
    float* out = dw + ((long long)n * Cin + c) * 27 + kz * 9;
#pragma unroll
    for (int ky = 0; ky < 3; ky++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 4; b++) s += G[a][ky] * G[b][kx] * u[a * 4 + b];
        out[ky * 3 + kx] = s;
      }
  }
}

struct GPlan {
  int nblk, cblk, Npad, Cpad, nty, ntx, S;
  long long ntiles, per_split;
};

GPlan gplan(const sr3d_conv_desc_t* d, int n_total) {
  GPlan g;
  g.nblk = ceil_div(n_total, 32), g.cblk = ceil_div(d->Cin, 32);
  g.Npad = g.nblk * 32, g.Cpad = g.cblk * 32;
  g.nty = ceil_div(d->Y, 2), g.ntx = ceil_div(d->X, 2 * GT);
  g.ntiles = (long long)d->B * g.nty * g.ntx * d->Z;
  long long want = ceil_div(1280, g.nblk * g.cblk);
  const long long slab_one = (long long)48 * g.Npad * g.Cpad * 4;
  const long long cap = (256ll << 20) / slab_one;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  if (want > g.ntiles) want = g.ntiles;
  g.per_split = (g.ntiles + want - 1) / want;
  g.S = (int)((g.ntiles + g.per_split - 1) / g.per_split);
  return g;
}

}  // namespace

size_t sr3d_wino_wgrad_ws_bytes(const sr3d_conv_desc_t* d, int n_total) {
  const GPlan g = gplan(d, n_total);
  return (size_t)g.S * 48 * g.Npad * g.Cpad * 4;
}

int sr3d_wino_wgrad(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy, int n_total, float* dw, float* ws,
                    hipStream_t st) {
  const GPlan g = gplan(d, n_total);
  WinoWgradParams p{};
  p.x = x, p.dy = dy, p.Cin = d->Cin, p.N = n_total;
  p.Z = d->Z, p.Y = d->Y, p.X = d->X;
  p.nty = g.nty, p.ntx = g.ntx, p.ntiles = g.ntiles, p.per_split = g.per_split;
  p.slab = ws, p.Npad = g.Npad, p.Cpad = g.Cpad;
  SR3D_CHECK(g.cblk <= 65535 && g.nblk <= 65535, SR3D_E_ARG, "winograd wgrad: too many blocks");
  static thread_local bool configured = false;
  if (!configured) {
    SR3D_HIP(hipFuncSetAttribute((const void*)wino_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGLds));
    configured = true;
  }
  void* tok = nullptr;
  if (sr3d_prof_active())
    sr3d_prof_begin(SR3D_PROF_WGRAD, 2.0 * 27 * d->Cin * (double)n_total * (double)d->Z * d->Y * d->X * d->B, st, &tok);
  hipLaunchKernelGGL(wino_wgrad_kernel, dim3(g.S, g.cblk, g.nblk), dim3(512), kGLds, st, p);
  sr3d_prof_end(tok, st);
  SR3D_HIP(hipGetLastError());
  const long long total = (long long)n_total * d->Cin * 3;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw, g.S, n_total,
                     d->Cin, g.Npad, g.Cpad);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}
