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
// for the MFMAs of the k-step -- V never goes through LDS.  A 512-thread workgroup covers a 4x4x32-voxel
// block: wave w works on z plane w&3 and on the xi rows 2*(w>>2), 2*(w>>2)+1 (8 accumulators = 128 VGPRs),
// so the two waves that share a SIMD (w, w+4) split the 16 transform points of the same tiles and their
// partial output transforms are added through LDS in the epilogue.  One workgroup per CU, two waves per SIMD.
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
constexpr int WNT = 512;               // threads per workgroup
constexpr int WNI = (WHCH + WNT - 1) / WNT;
constexpr size_t kWinoLds = (size_t)2 * (WHS + WUS) * 4;

typedef const __attribute__((address_space(1))) float* gfloat_p;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wact(float v, int act) {
  if (act == SR3D_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == SR3D_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
  return v;
}

__global__ __launch_bounds__(512, 2) void wino_kernel(const SrWinoParams p) {
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
    const int e = tid + i * WNT;
    const int hz = e / (WHY * WHX), r2 = e - hz * (WHY * WHX);
    const int hy = r2 / WHX, hx = r2 - hy * WHX;
    const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = e < WHCH && (unsigned)gz < (unsigned)p.Z && (unsigned)gy < (unsigned)p.Y &&
                    (unsigned)gx < (unsigned)p.X;
    hoff[i] = ok ? (gz * p.Y + gy) * p.X + gx : -1;
  }

  const int plane = wave & 3, half = wave >> 2;   // z plane of the block; xi rows {2*half, 2*half+1}
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;

  const int t = lane & 31, ty = t >> 4, tx = t & 15;
  const int hb = (lane >> 5) * WHCH + (plane * WHY + 2 * ty) * WHX + 2 * tx;  // this lane's patch origin
  const int ub = (lane >> 5) * 32 + (lane & 31) + half * (8 * WKC * 32);

  // Branch-free halo prefetch (so that it can be scheduled between MFMAs): out-of-grid elements and channels
  // beyond K load a valid dummy address and are zeroed by a select.
  float hv[WKC][WNI];
  int hclamp[WNI];
#pragma unroll
  for (int i = 0; i < WNI; i++) hclamp[i] = hoff[i] >= 0 ? hoff[i] : 0;
  const gfloat_p dummy = (gfloat_p)p.in.ptr[0];
  auto load_halo = [&](const int chunk) {
#pragma unroll
    for (int c = 0; c < WKC; c++) {
      const int gc = chunk * WKC + c;  // wave-uniform
      const bool cok = gc < p.K;
      gfloat_p base = dummy;
      if (cok) {
        const int si = cat_find(p.in, gc);
        base = (gfloat_p)cat_ptr(p.in, si) + ((long long)b * cat_bstride(p.in, si) + (long long)(gc - cat_cbeg(p.in, si)) * ZYX);
      }
#pragma unroll
      for (int i = 0; i < WNI; i++) hv[c][i] = base[hclamp[i]];   // masked when written to LDS (no wait here)
    }
  };
  auto store_halo = [&](float* H, const int chunk) {
#pragma unroll
    for (int c = 0; c < WKC; c++) {
      const bool cok = chunk * WKC + c < p.K;
#pragma unroll
      for (int i = 0; i < WNI; i++)
        if (tid + i * WNT < WHCH) H[c * WHCH + tid + i * WNT] = (cok && hoff[i] >= 0) ? hv[c][i] : 0.f;
    }
  };
  auto dma_u = [&](const int chunk, float* U) {
    const float* gw = p.up + (size_t)(nblk * p.nchunks + chunk) * WUS;
    constexpr int NINSTR = WUS / 256;  // 24 pieces of 1 KiB
    for (int i = wave; i < NINSTR; i += 8)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw + i * 256 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(U + i * 256), 16, 0, 0);
  };

  // one k-step = (kz, channel pair): this lane's A fragments (U, from LDS) and B fragments (transformed patch)
  auto fetch = [&](const float* H, const float* U, const int ks, float (&u)[8], float (&vv)[8]) {
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
    for (int j = 0; j < 4; j++) {   // along y: only this wave's two rows of B^T
      vv[j] = half == 0 ? tt[0][j] - tt[2][j] : tt[2][j] - tt[1][j];
      vv[4 + j] = half == 0 ? tt[1][j] + tt[2][j] : tt[1][j] - tt[3][j];
    }

    const float* up = U + ((kz * 16) * WKC + 2 * cp) * 32 + ub;
#pragma unroll
    for (int q = 0; q < 8; q++) u[q] = up[q * (WKC * 32)];
  };

  // ---- prologue: chunk 0
  load_halo(0);
  dma_u(0, Us0);
  store_halo(Hs0, 0);

  constexpr int KSTEPS = 3 * (WKC / 2);
  for (int chunk = 0; chunk < p.nchunks; chunk++) {
    const int cur = chunk & 1;
    const float* H = Hs0 + cur * WHS;
    const float* U = Us0 + cur * WUS;
    __syncthreads();  // buffers `cur` are complete (the compiler drains vmcnt here); buffers `cur^1` are free
    const bool more = chunk + 1 < p.nchunks;
    if (more) dma_u(chunk + 1, Us0 + (cur ^ 1) * WUS);    // straight into the other weight buffer
    // The fetch of k-step s+1 (8 ds_read_b64 + 16 ds_read_b32 + ~40 VALU) is interleaved with the 16 MFMAs
    // of k-step s: an MFMA occupies the matrix pipe for 64 cycles but the issue port only for 8, and with one
    // wave per SIMD nobody else would use the gaps.  sched_group_barrier pins "1 MFMA, 2 LDS reads, 3 VALU".
    float u0[8], v0[8], u1[8], v1[8];
    fetch(H, U, 0, u0, v0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks += 2) {
      fetch(H, U, ks + 1, u1, v1);
      if (ks == 0) load_halo(more ? chunk + 1 : chunk);   // next chunk's raw input -> registers (the last chunk
                                                          // re-reads its own: harmless and keeps this branch-free)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 8; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0[q], v0[q], acc[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 2 < KSTEPS) fetch(H, U, ks + 2, u0, v0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 8; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1[q], v1[q], acc[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) store_halo(Hs0 + (cur ^ 1) * WHS, chunk + 1);
  }

  // ---- epilogue: Y = A^T M A.  Each wave transforms its two xi rows; the partial results of the upper half
  // (waves 4..7) go through LDS to the lower half, which adds its own, applies bias / activation / gate and stores.
  float part[16][2][2];   // [element][yo][xo]
#pragma unroll
  for (int e = 0; e < 16; e++) {
    float s[2][2];        // rows (this wave's xi_y = 2*half + r) after the x transform A^T = [1 1 1 0; 0 1 -1 -1]
#pragma unroll
    for (int r = 0; r < 2; r++) {
      s[r][0] = acc[r * 4 + 0][e] + acc[r * 4 + 1][e] + acc[r * 4 + 2][e];
      s[r][1] = acc[r * 4 + 1][e] - acc[r * 4 + 2][e] - acc[r * 4 + 3][e];
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      // y transform: rows 0,1 contribute (m0 + m1, m1); rows 2,3 contribute (m2, -m2 - m3)
      part[e][0][j] = half == 0 ? s[0][j] + s[1][j] : s[0][j];
      part[e][1][j] = half == 0 ? s[1][j] : -s[0][j] - s[1][j];
    }
  }
  __syncthreads();   // all MFMA phases are done: the staging buffers can carry the exchange
  float* X = lds + (plane * 64 + lane) * 65;   // 64 floats per lane, odd pitch
  if (half == 1) {
#pragma unroll
    for (int e = 0; e < 16; e++)
#pragma unroll
      for (int q = 0; q < 4; q++) X[e * 4 + q] = part[e][q >> 1][q & 1];
  }
  __syncthreads();
  if (half == 1) return;
#pragma unroll
  for (int e = 0; e < 16; e++)
#pragma unroll
    for (int q = 0; q < 4; q++) part[e][q >> 1][q & 1] += X[e * 4 + q];

  const int oz = z0 + plane;
  if (oz >= p.Z) return;
  const int oy = y0 + 2 * ty, ox = x0 + 2 * tx;
  const long long TZYX = (long long)p.TZ_ * p.TY_ * p.TX_;
  const int row0 = p.n_off + nblk * 32 + 4 * (lane >> 5);
  const bool pair_ok = (p.TX_ & 1) == 0 && p.pair_aligned;   // (x, x+1) pairs are 8-byte aligned in every destination

#pragma unroll
  for (int e = 0; e < 16; e++) {
    // gated: element e < 8 is the feature row, e + 8 the gate row of the same channel
    if (p.epi == SR3D_EPI_GATED && e >= 8) break;
    const int rl = (e & 3) + 8 * (e >> 2);  // row inside the 32-row block (before + 4*(lane>>5))
    if (p.epi == SR3D_EPI_GATED) {
      const int co = (p.n_off + nblk * 32) / 2 + 4 * (lane >> 5) + rl;  // 16 channels per row block
      if (co >= p.Cg) continue;
      const float bf = p.bias ? p.bias[co] : 0.f, bg = p.bias2 ? p.bias2[co] : 0.f;
      const int eg = e + 8 < 16 ? e + 8 : e;
#pragma unroll
      for (int yo = 0; yo < 2; yo++) {
        if (oy + yo >= p.Y) continue;
        float f[2], sg[2];
#pragma unroll
        for (int xo = 0; xo < 2; xo++) {
          f[xo] = wact(part[e][yo][xo] + bf, p.act);
          sg[xo] = 1.f / (1.f + expf(-(part[eg][yo][xo] + bg)));
        }
        const long long o = ((long long)b * p.Cg + co) * TZYX + ((long long)oz * p.TY_ + oy + yo) * p.TX_ + ox;
        if (pair_ok && ox + 1 < p.X) {   // both voxels of the tile row: one 8-byte store per tensor
          *reinterpret_cast<f32x2*>(p.y + o) = f32x2{sg[0] * f[0], sg[1] * f[1]};
          if (p.save_f) {
            *reinterpret_cast<f32x2*>(p.save_f + o) = f32x2{f[0], f[1]};
            *reinterpret_cast<f32x2*>(p.save_s + o) = f32x2{sg[0], sg[1]};
          }
        } else {
#pragma unroll
          for (int xo = 0; xo < 2; xo++) {
            if (ox + xo >= p.X) continue;
            p.y[o + xo] = sg[xo] * f[xo];
            if (p.save_f) p.save_f[o + xo] = f[xo], p.save_s[o + xo] = sg[xo];
          }
        }
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
                wact(part[e][yo][xo] + bv, p.act);
          }
      } else {
        const int si = cat_find(p.out, n);
        float* base = cat_ptr(p.out, si);
        if (base == nullptr) continue;
        base += (long long)b * cat_bstride(p.out, si) + (long long)(n - cat_cbeg(p.out, si)) * TZYX;
#pragma unroll
        for (int yo = 0; yo < 2; yo++) {
          if (oy + yo >= p.Y) continue;
          float* o = base + ((long long)oz * p.TY_ + oy + yo) * p.TX_ + ox;
          const float r0 = wact(part[e][yo][0] + bv, p.act), r1 = wact(part[e][yo][1] + bv, p.act);
          if (pair_ok && ox + 1 < p.X) {
            *reinterpret_cast<f32x2*>(o) = f32x2{r0, r1};
          } else {
            if (ox < p.X) o[0] = r0;
            if (ox + 1 < p.X) o[1] = r1;
          }
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
  {
    uintptr_t bits = reinterpret_cast<uintptr_t>(p.y) | reinterpret_cast<uintptr_t>(p.save_f) | reinterpret_cast<uintptr_t>(p.save_s);
    for (int i = 0; i < p.out.n; i++) bits |= reinterpret_cast<uintptr_t>(p.out.ptr[i]);
    p.pair_aligned = (bits & 7) == 0;
  }
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
  hipLaunchKernelGGL(wino_kernel, dim3((unsigned)nwg, B), dim3(WNT), kWinoLds, st, p);
  sr3d_prof_end(tok, st);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}
