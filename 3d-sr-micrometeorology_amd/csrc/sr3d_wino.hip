// Winograd F(2x2, 3x3) in the (y, x) plane x direct 3 taps in z, on the fp32 MFMA, for the stride-1
// 3x3x3 convolutions (forward and input gradient).
//
// Why: gfx950's f32-input MFMA runs at the fp32 VALU rate (1/16 of bf16), so the convolutions are bound by
// the NUMBER of fp32 multiplications.  The 2-D Winograd transform needs 16 products per 2x2 outputs and kz-tap
// instead of 36, i.e. 48 instead of 108 per 2x2x1 outputs: 2.25x fewer MFMA issues at unchanged fp32 accuracy
// (measured normwise error vs fp64 3e-7, the same as the direct fp32 kernel; all arithmetic is still fp32).
//
//   U[kz][xi][c][n] = (G w[n][c][kz] G^T)[xi]              weights, transformed once per call (wino_pack_kernel)
//   V[xi][c][tile]  = (B^T d B)[xi]                         4x4 input patch of a 2x2 output tile, on the fly
//   M[xi][n][tile] += sum_c U[kz][xi][c][n] * V[xi][c][tile]     <- 16 independent GEMMs = the MFMA work
//   Y[n][tile 2x2]  = A^T M A                               epilogue
//
// Mapping: MFMA rows = 32 output channels (A operand = U, read from LDS), MFMA columns = 32 tiles (2 tile
// rows x 16 tile columns = 4 x 32 voxels of one z plane).  The lane that owns column `tile` and k-slot
// `lane>>5` transforms exactly that (tile, channel) patch, so the 16 transformed values ARE its B operands
// for the 16 MFMAs of the k-step -- V never goes through LDS.  A wave owns one z plane of the 4x4x32-voxel
// workgroup block and all 16 xi accumulators (256 VGPRs); one workgroup per CU, one wave per SIMD.
// Staging is double buffered: raw input halo through registers (issued before, written after the MFMA
// phase of the previous chunk), transformed weights by LDS-DMA.
#include "sr3d_common.h"

#include <limits.h>
#include <stdlib.h>

namespace {

constexpr int WKC = 4;                 // input channels per chunk
constexpr int WHZ = 6, WHY = 6, WHX = 34;
constexpr int WHCH = WHZ * WHY * WHX;  // halo floats per channel (1224)
constexpr int WHS = WKC * WHCH;        // 4896 floats
constexpr int WUS = 3 * 16 * WKC * 32; // 6144 floats: [kz][xi][kc][32 rows]
constexpr int WNI = (WHCH + 255) / 256;
constexpr size_t kWinoLds = (size_t)2 * (WHS + WUS) * 4;

typedef const __attribute__((address_space(1))) float* gfloat_p;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wact(float v, int act) {
  if (act == SR3D_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == SR3D_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
  return v;
}

__global__ __launch_bounds__(256, 1) void wino_kernel(const SrWinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Hs0 = lds;
  float* Us0 = lds + 2 * WHS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // workgroup -> (row block, spatial block); row blocks of one spatial block adjacent on one XCD
  int v;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nblk = v % p.nblk;
  int blk = v / p.nblk;
  const int tix = blk % p.ntx;
  blk /= p.ntx;
  const int tiy = blk % p.nty;
  const int tiz = blk / p.nty;
  const int b = blockIdx.y;
  const int z0 = tiz * 4, y0 = tiy * 4, x0 = tix * 32;
  const long long ZYX = (long long)p.Z * p.Y * p.X;

  // spatial offsets of this thread's halo elements; -1 = outside the grid (zero padding)
  int hoff[WNI];
#pragma unroll
  for (int i = 0; i < WNI; i++) {
    const int e = tid + i * 256;
    const int hz = e / (WHY * WHX), r2 = e - hz * (WHY * WHX);
    const int hy = r2 / WHX, hx = r2 - hy * WHX;
    const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = e < WHCH && (unsigned)gz < (unsigned)p.Z && (unsigned)gy < (unsigned)p.Y &&
                    (unsigned)gx < (unsigned)p.X;
    hoff[i] = ok ? (gz * p.Y + gy) * p.X + gx : -1;
  }

  f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;

  const int t = lane & 31, ty = t >> 4, tx = t & 15;
  const int hb = (lane >> 5) * WHCH + (wave * WHY + 2 * ty) * WHX + 2 * tx;  // this lane's patch origin
  const int ub = (lane >> 5) * 32 + (lane & 31);

  float hv[WKC][WNI];
  auto load_halo = [&](const int chunk) {
#pragma unroll
    for (int c = 0; c < WKC; c++) {
      const int gc = chunk * WKC + c;  // wave-uniform
      gfloat_p base = nullptr;
      if (gc < p.K) {
        const int si = cat_find(p.in, gc);
        base = (gfloat_p)cat_ptr(p.in, si) + ((long long)b * cat_bstride(p.in, si) + (long long)(gc - cat_cbeg(p.in, si)) * ZYX);
      }
#pragma unroll
      for (int i = 0; i < WNI; i++) hv[c][i] = (base != nullptr && hoff[i] >= 0) ? base[hoff[i]] : 0.f;
    }
  };
  auto store_halo = [&](float* H) {
#pragma unroll
    for (int c = 0; c < WKC; c++)
#pragma unroll
      for (int i = 0; i < WNI; i++)
        if (tid + i * 256 < WHCH) H[c * WHCH + tid + i * 256] = hv[c][i];
  };
  auto dma_u = [&](const int chunk, float* U) {
    const float* gw = p.up + (size_t)(nblk * p.nchunks + chunk) * WUS;
    constexpr int NINSTR = WUS / 256;  // 24 pieces of 1 KiB
    for (int i = wave; i < NINSTR; i += 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw + i * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(U + i * 256), 16, 0, 0);
  };

  // one k-step = (kz, channel pair): this lane's A fragments (U, from LDS) and B fragments (transformed patch)
  auto fetch = [&](const float* H, const float* U, const int ks, float (&u)[16], float (&vv)[16]) {
    const int kz = ks / (WKC / 2), cp = ks % (WKC / 2);
    const float* hp = H + hb + (2 * cp) * WHCH + kz * (WHY * WHX);
    float d[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const f32x2 a = *reinterpret_cast<const f32x2*>(hp + i * WHX);
      const f32x2 c2 = *reinterpret_cast<const f32x2*>(hp + i * WHX + 2);
      d[i][0] = a.x, d[i][1] = a.y, d[i][2] = c2.x, d[i][3] = c2.y;
    }
    float tt[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++) {   // along x:  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
      tt[i][0] = d[i][0] - d[i][2];
      tt[i][1] = d[i][1] + d[i][2];
      tt[i][2] = d[i][2] - d[i][1];
      tt[i][3] = d[i][1] - d[i][3];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {   // along y
      vv[0 * 4 + j] = tt[0][j] - tt[2][j];
      vv[1 * 4 + j] = tt[1][j] + tt[2][j];
      vv[2 * 4 + j] = tt[2][j] - tt[1][j];
      vv[3 * 4 + j] = tt[1][j] - tt[3][j];
    }
    const float* up = U + ((kz * 16) * WKC + 2 * cp) * 32 + ub;
#pragma unroll
    for (int xi = 0; xi < 16; xi++) u[xi] = up[xi * (WKC * 32)];
  };

  // ---- prologue: chunk 0
  load_halo(0);
  dma_u(0, Us0);
  store_halo(Hs0);

  constexpr int KSTEPS = 3 * (WKC / 2);
  for (int chunk = 0; chunk < p.nchunks; chunk++) {
    const int cur = chunk & 1;
    const float* H = Hs0 + cur * WHS;
    const float* U = Us0 + cur * WUS;
    __syncthreads();  // buffers `cur` are complete (the compiler drains vmcnt here); buffers `cur^1` are free
    const bool more = chunk + 1 < p.nchunks;
    if (more) {
      load_halo(chunk + 1);                       // into registers, written to LDS after the MFMA phase
      dma_u(chunk + 1, Us0 + (cur ^ 1) * WUS);    // straight into the other weight buffer
    }
    float u0[16], v0[16], u1[16], v1[16];
    fetch(H, U, 0, u0, v0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks += 2) {
      fetch(H, U, ks + 1, u1, v1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int xi = 0; xi < 16; xi++) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0[xi], v0[xi], acc[xi], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 2 < KSTEPS) fetch(H, U, ks + 2, u0, v0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int xi = 0; xi < 16; xi++) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1[xi], v1[xi], acc[xi], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) store_halo(Hs0 + (cur ^ 1) * WHS);
  }

  // ---- epilogue: Y = A^T M A per accumulator element, then bias / activation / gate / scatter
  const int oz = z0 + wave;
  if (oz >= p.Z) return;
  const int oy = y0 + 2 * ty, ox = x0 + 2 * tx;
  const long long TZYX = (long long)p.TZ_ * p.TY_ * p.TX_;
  const int row0 = p.n_off + nblk * 32 + 4 * (lane >> 5);

#pragma unroll
  for (int e = 0; e < 16; e++) {
    // gated: element e < 8 is the feature row, e + 8 the gate row of the same channel
    if (p.epi == SR3D_EPI_GATED && e >= 8) break;
    float yv[2][2], gv[2][2];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      if (pass == 1 && p.epi != SR3D_EPI_GATED) break;
      const int ee = pass == 0 ? e : e + 8;
      float s[4][2];
#pragma unroll
      for (int i = 0; i < 4; i++) {   // along x: A^T = [1 1 1 0; 0 1 -1 -1]
        s[i][0] = acc[i * 4 + 0][ee] + acc[i * 4 + 1][ee] + acc[i * 4 + 2][ee];
        s[i][1] = acc[i * 4 + 1][ee] - acc[i * 4 + 2][ee] - acc[i * 4 + 3][ee];
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const float a0 = s[0][j] + s[1][j] + s[2][j], a1 = s[1][j] - s[2][j] - s[3][j];
        if (pass == 0)
          yv[0][j] = a0, yv[1][j] = a1;
        else
          gv[0][j] = a0, gv[1][j] = a1;
      }
    }
    const int rl = (e & 3) + 8 * (e >> 2);  // row inside the 32-row block (before + 4*(lane>>5))
    if (p.epi == SR3D_EPI_GATED) {
      const int co = (p.n_off + nblk * 32) / 2 + 4 * (lane >> 5) + rl;  // 16 channels per row block
      if (co >= p.Cg) continue;
      const float bf = p.bias ? p.bias[co] : 0.f, bg = p.bias2 ? p.bias2[co] : 0.f;
#pragma unroll
      for (int yo = 0; yo < 2; yo++)
#pragma unroll
        for (int xo = 0; xo < 2; xo++) {
          if (oy + yo >= p.Y || ox + xo >= p.X) continue;
          const float f = wact(yv[yo][xo] + bf, p.act);
          const float sg = 1.f / (1.f + expf(-(gv[yo][xo] + bg)));
          const long long o = ((long long)b * p.Cg + co) * TZYX + ((long long)oz * p.TY_ + oy + yo) * p.TX_ + ox + xo;
          p.y[o] = sg * f;
          if (p.save_f) p.save_f[o] = f, p.save_s[o] = sg;
        }
    } else {
      const int n = row0 + rl;
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
      if (p.epi == SR3D_EPI_UNSHUFFLE) {
        const int f = n / p.unsh_C, c = n - f * p.unsh_C;
        float* base = p.y + ((long long)b * p.unsh_C + c) * TZYX;
#pragma unroll
        for (int yo = 0; yo < 2; yo++)
#pragma unroll
          for (int xo = 0; xo < 2; xo++) {
            if (oy + yo >= p.Y || ox + xo >= p.X) continue;
            base[((long long)(2 * oz + (f >> 2)) * p.TY_ + 2 * (oy + yo) + ((f >> 1) & 1)) * p.TX_ + 2 * (ox + xo) + (f & 1)] =
                wact(yv[yo][xo] + bv, p.act);
          }
      } else {
        const int si = cat_find(p.out, n);
        float* base = cat_ptr(p.out, si);
        if (base == nullptr) continue;
        base += (long long)b * cat_bstride(p.out, si) + (long long)(n - cat_cbeg(p.out, si)) * TZYX;
#pragma unroll
        for (int yo = 0; yo < 2; yo++)
#pragma unroll
          for (int xo = 0; xo < 2; xo++) {
            if (oy + yo >= p.Y || ox + xo >= p.X) continue;
            base[((long long)oz * p.TY_ + oy + yo) * p.TX_ + ox + xo] = wact(yv[yo][xo] + bv, p.act);
          }
      }
    }
  }
}

// ---- weight transform + packing: image [nblk][chunk][kz][xi][kc][32 rows]
struct WinoPackParams {
  const float* w1;
  const float* w2;
  float* up;
  int Cout, Cin, kind, K, N, nchunks, nblk;
  int rbeg[SR3D_MAX_SRC + 1];
  int cbeg[SR3D_MAX_SRC];
};

__global__ void wino_pack_kernel(const WinoPackParams p) {
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  const long long total = (long long)p.nblk * p.nchunks * WUS;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    long long r = e;
    const int rr = r % 32;
    r /= 32;
    const int kc = r % WKC;
    r /= WKC;
    const int xi = r % 16;
    r /= 16;
    const int kz = r % 3;
    r /= 3;
    const int chunk = r % p.nchunks;
    const int nb = r / p.nchunks;
    const int n = nb * 32 + rr, k = chunk * WKC + kc;
    float val = 0.f;
    if (n < p.N && k < p.K) {
      const float* w = nullptr;  // -> w[co][ci][0][0][0]
      bool flip = false;
      if (p.kind == SR3D_PACK_FWD) {
        w = p.w1 + ((long long)n * p.Cin + k) * 27;
      } else if (p.kind == SR3D_PACK_FWD_GATED) {
        const int co = (n >> 5) * 16 + (n & 15);  // rows 0..15 of a block: features, 16..31: gates
        if (co < p.Cout) w = ((n & 16) ? p.w2 : p.w1) + ((long long)co * p.Cin + k) * 27;
      } else {  // input gradient: rows = input channels that need a gradient, K = output channels, taps mirrored
        const int si = (n >= p.rbeg[1]) + (n >= p.rbeg[2]) + (n >= p.rbeg[3]);
        const int ci = p.cbeg[si] + (n - p.rbeg[si]);
        w = (k < p.Cout ? p.w1 + (long long)k * p.Cin * 27 : p.w2 + (long long)(k - p.Cout) * p.Cin * 27) + ci * 27;
        flip = true;
      }
      if (w != nullptr) {
        const int xy = xi >> 2, xx = xi & 3;
        const int kzz = flip ? 2 - kz : kz;
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
          for (int kx = 0; kx < 3; kx++) {
            const int kyy = flip ? 2 - ky : ky, kxx = flip ? 2 - kx : kx;
            val += G[xy][ky] * G[xx][kx] * w[(kzz * 3 + kyy) * 3 + kxx];
          }
      }
    }
    p.up[e] = val;
  }
}

}  // namespace

bool sr3d_wino_enabled() {
  static const bool on = getenv("SR3D_WINOGRAD") ? atoi(getenv("SR3D_WINOGRAD")) != 0 : true;
  return on;
}

size_t sr3d_wino_image_floats(int rows, int K) { return (size_t)ceil_div(rows, 32) * ceil_div(K, WKC) * WUS; }

int sr3d_wino_pack(int kind, int Cout, int Cin, int rows, int K, const float* w1, const float* w2, const int* rbeg,
                   const int* cbeg, float* image, hipStream_t st) {
  WinoPackParams p{};
  p.w1 = w1, p.w2 = w2, p.up = image;
  p.Cout = Cout, p.Cin = Cin, p.kind = kind, p.K = K, p.N = rows;
  p.nchunks = ceil_div(K, WKC), p.nblk = ceil_div(rows, 32);
  for (int i = 0; i <= SR3D_MAX_SRC; i++) p.rbeg[i] = rbeg ? rbeg[i] : INT_MAX;
  for (int i = 0; i < SR3D_MAX_SRC; i++) p.cbeg[i] = cbeg ? cbeg[i] : 0;
  const long long total = (long long)p.nblk * p.nchunks * WUS;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, st, p);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

int sr3d_wino_launch(SrWinoParams& p, int B, hipStream_t st) {
  p.ntz = ceil_div(p.Z, 4), p.nty = ceil_div(p.Y, 4), p.ntx = ceil_div(p.X, 32);
  p.nblk = ceil_div(p.N, 32);
  p.nchunks = ceil_div(p.K, WKC);
  const long long nwg = (long long)p.ntz * p.nty * p.ntx * p.nblk;
  SR3D_CHECK(nwg < (1ll << 31) && B <= 65535, SR3D_E_ARG, "winograd conv: grid too large");
  static thread_local bool configured = false;
  if (!configured) {
    SR3D_HIP(hipFuncSetAttribute((const void*)wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLds));
    configured = true;
  }
  void* tok = nullptr;
  if (sr3d_prof_active()) {
    const double rows = p.epi == SR3D_EPI_GATED ? 2.0 * p.Cg : (double)p.N;
    sr3d_prof_begin(SR3D_PROF_IGEMM_S1, 2.0 * 27 * p.K * rows * (double)p.Z * p.Y * p.X * B, st, &tok);
  }
  hipLaunchKernelGGL(wino_kernel, dim3((unsigned)nwg, B), dim3(256), kWinoLds, st, p);
  sr3d_prof_end(tok, st);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}
