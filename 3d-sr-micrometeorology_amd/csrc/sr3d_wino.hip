// Winograd F(2x2, 3x3) in the (y, x) plane x direct 3 taps in z, on the fp32 MFMA, for the stride-1
// 3x3x3 convolutions (forward and input gradient).
//
// Why: gfx950's f32-input MFMA runs at the fp32 VALU rate (1/16 of bf16), so the convolutions are bound by
// the NUMBER of fp32 multiplications.  The 2-D Winograd transform needs 16 products per 2x2 outputs and kz-tap
// instead of 36, i.e. 48 instead of 108 per 2x2x1 outputs: 2.25x fewer MFMA issues at unchanged fp32 accuracy
// (measured normwise error vs fp64 3e-7, the same as the direct fp32 kernel; all arithmetic is still fp32).
//
//   U[kz][xi][c][n] = (G w[n][c][kz] G^T)[xi]              weights, transformed once per call (wino_pack_kernel)
//   V[z][xi][c][tile] = (B^T d B)[xi]                       4x4 input patch of a 2x2 output tile
//   M[xi][n][z][tile] += sum_{c,kz} U[kz][xi][c][n] * V[z+kz][xi][c][tile]    <- 16 independent GEMMs = MFMA work
//   Y[n][z][tile 2x2]  = A^T M A                            epilogue
//
// One 512-thread workgroup (alone on its CU) owns 64 output rows x 3 z planes x (4 x 32 voxels = 32 tiles).
// Wave w keeps the transform points xi = 2w, 2w+1 for both 32-row tiles and the 3 planes: 12 accumulators.  Both
// operands come from LDS as plain 64-float fragments: U by LDS-DMA from the packed image, V written ONCE per
// workgroup and chunk by a cooperative transform (5 input planes x 2 channels x 32 tiles, half a patch per
// thread) and then shared by the 64 rows and the 3 kz taps.  Per wave and chunk of 2 channels that is 36 MFMAs
// against ~30 LDS fragment reads and ~40 transform instructions (the first version of this kernel transformed
// the patch in the lane that fed it to the MFMA: 7 non-MFMA instructions per MFMA, 55 % pipe utilisation).
//
// Pipeline per chunk k (ONE barrier): U(k+1) and the raw input rows of chunk k+3 are in flight as LDS-DMA
// (buffer loads with hardware range checks: zero padding, ragged blocks and channels >= K cost no instruction),
// the V transform of chunk k+1 is scheduled between the MFMAs of chunk k, fragments are double-buffered per kz.
// Epilogue: the 16 transform points of an output live in 8 waves; each wave reduces its pair along x, the partial
// sums go through LDS (conflict-free swizzle) and are added in a fixed order (deterministic), then bias /
// activation / gate / unshuffle / virtual-concat stores as before, 128 contiguous bytes per 16 lanes.
#include "sr3d_common.h"

#include <limits.h>
#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int WKC = 2;                      // input channels per chunk
constexpr int WPL = 3;                      // output planes per workgroup
constexpr int WVP = WPL + 2;                // input planes per workgroup
// Voxel tile of one plane = 32 Winograd tiles: 4 x 32 voxels (TS = 0) or 8 x 16 voxels (TS = 1, for the 80- and
// 40-wide grids of the deeper U-Net levels, where 32-wide tiles would be 20 % / 60 % padding)
template <int TS>
struct WinoGeo {
  static constexpr int TR = TS ? 4 : 2, TC = TS ? 8 : 16;   // tile rows x tile columns
  static constexpr int VY = 2 * TR, VX = 2 * TC;            // voxels
  static constexpr int HY = VY + 2, HX = VX + 2;            // halo rows x columns of one plane
  static constexpr int RE = WVP * HY * HX;                  // raw floats per channel (1020 / 900)
};
constexpr int WRC = 1056;                   // raw channel pitch (== 32 mod 64: the two channels use disjoint banks)
constexpr int WRB = WKC * WRC;              // raw buffer
constexpr int WVB = WVP * 16 * WKC * 32;    // V buffer [plane][xi][c][tile]
constexpr int WUS = 3 * 16 * 2 * WKC * 32;  // U buffer [kz][xi][row tile][c][32 rows] = one packed image piece
constexpr int WNT = 512;
constexpr int WXB = 8 * 2 * 1024;           // epilogue exchange buffer [wave][value][32 rows x 32 tiles]
constexpr int WNU = 3;                      // U buffers: the packed weights of chunk k+2 are in flight during chunk k
constexpr int kWinoLdsFloats = (2 * WVB + WNU * WUS + 3 * WRB) > 2 * WXB ? (2 * WVB + WNU * WUS + 3 * WRB) : 2 * WXB;
constexpr int WCT = SR3D_WINO_MAX_K;        // channel-pointer table entries (8 bytes each) behind the buffers
constexpr size_t kWinoLds = (size_t)kWinoLdsFloats * 4 + (size_t)WCT * 8;
static_assert(kWinoLds <= 160 * 1024, "LDS budget");

typedef const __attribute__((address_space(1))) float* gfloat_p;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_p;

__device__ __forceinline__ float wact(float v, int act) {
  if (act == SR3D_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == SR3D_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
  return v;
}

template <int NRT, int TS>   // NRT: 32-row tiles per workgroup (2, or 1 for a last block with <= 32 rows); TS: tile shape
__global__ __launch_bounds__(512, 2) void wino_kernel(const SrWinoParams p) {
  using G = WinoGeo<TS>;
  constexpr int WHY = G::HY, WHX = G::HX, WRE = G::RE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Vs = lds;                 // 2 buffers
  float* Us = lds + 2 * WVB;       // WNU buffers
  float* Rs = Us + WNU * WUS;      // 3 buffers

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // workgroup -> (row block, spatial block); row blocks of one spatial block adjacent on one XCD
  int v;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // 64-row block of the layer (index into the packed image) and, for the one-tile variant on small grids, which
  // of its two 32-row tiles this workgroup takes (rt_split: the launch has two workgroups per 64-row block)
  const int rblk = v % p.nblk;
  const int nblk = p.nb_off + (NRT == 1 && p.rt_split ? rblk >> 1 : rblk);
  const int rt0 = (NRT == 1 && p.rt_split) ? (rblk & 1) : 0;
  int blk = v / p.nblk;
  const int tix = blk % p.ntx;
  blk /= p.ntx;
  const int tiy = blk % p.nty;
  const int tiz = blk / p.nty;
  const int b = blockIdx.y;
  const int z0 = tiz * WPL, y0 = tiy * G::VY, x0 = tix * G::VX;
  const long long ZYX = (long long)p.Z * p.Y * p.X;
  const int chan_bytes = (int)(ZYX * 4);

  // ---- raw rows: LDS-DMA, lane i of a wave instruction lands at dst + 4 i.  Wave w covers elements
  // [64 (w + 8 i), +64) of a channel's [plane][row][col] halo, i = 0, 1; byte offsets are fixed for the block.
  unsigned roff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int e = (wave + 8 * i) * 64 + lane;
    const int pl = e / (WHY * WHX), r2 = e - pl * (WHY * WHX);
    const int hy = r2 / WHX, hx = r2 - hy * WHX;
    const int gz = z0 - 1 + pl, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = e < WRE && (unsigned)gz < (unsigned)p.Z && (unsigned)gy < (unsigned)p.Y && (unsigned)gx < (unsigned)p.X;
    roff[i] = ok ? (unsigned)((gz * p.Y + gy) * p.X + gx) * 4u : 0xffffffffu;
  }
  // Base pointer of every input channel (virtual concat, this sample), worked out once into an LDS table: the
  // per-chunk descriptor is then one broadcast LDS read + two readfirstlanes instead of ~60 scalar instructions
  // of slice lookup and 64-bit multiplies, which sat in every wave's instruction stream next to the MFMAs.
  unsigned long long* ctab = reinterpret_cast<unsigned long long*>(lds + kWinoLdsFloats);
  for (int c = tid; c < p.K; c += WNT) {
    const int si = cat_find(p.in, c);
    ctab[c] = reinterpret_cast<unsigned long long>(cat_ptr(p.in, si) + ((long long)b * cat_bstride(p.in, si) +
                                                                        (long long)(c - cat_cbeg(p.in, si)) * ZYX));
  }
  __syncthreads();
  auto dma_raw = [&](const int chunk, float* R) {
#pragma unroll
    for (int c = 0; c < WKC; c++) {
      const int gc = chunk * WKC + c;   // wave-uniform
      const int gcc = gc < p.K ? gc : p.K - 1;   // branch-free: channels past the end get an empty descriptor
      const unsigned long long a = ctab[gcc];
      const unsigned long long base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)a);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, gc < p.K ? chan_bytes : 0, 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; i++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_p)(R + c * WRC + (wave + 8 * i) * 64), 4, roff[i], 0, 0, 0);
    }
  };
  auto dma_u = [&](const int chunk, float* U) {
    const int ch = chunk < p.nchunks ? chunk : p.nchunks - 1;   // past the end: re-read the last piece (never used)
    const float* gw = p.up + (size_t)(nblk * p.nchunks + ch) * WUS;
#pragma unroll
    for (int i = 0; i < 3; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw + (wave + 8 * i) * 256 + lane * 4),
                                       (lds_p)(U + (wave + 8 * i) * 256), 16, 0, 0);
  };

  // ---- V transform, half a patch per thread: item q = (plane, half) is wave-uniform, lanes = (tile row, channel,
  // tile column).  half 0 produces xi_x = 0, 1 from patch columns 0, 1, 2; half 1 xi_x = 2, 3 from columns 2, 3, 1.
  const int ttx = lane & (G::TC - 1), tch = (lane / G::TC) & 1, tty = lane / (2 * G::TC);
  const int t_rd = tch * WRC + (2 * tty) * WHX + 2 * ttx;
  const int t_wr = tch * 32 + tty * G::TC + ttx;
  float vt[4][2];
  auto tv_read = [&](const float* R, const int q) {
    const int pl = q >> 1, h = q & 1;
    const float sgn = h ? -1.f : 1.f;
    const float* rp = R + t_rd + pl * (WHY * WHX);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const f32x2 pq = *reinterpret_cast<const f32x2*>(rp + i * WHX + (h ? 2 : 0));
      const float r = rp[i * WHX + (h ? 1 : 2)];
      vt[i][0] = pq.x - r;
      vt[i][1] = fmaf(pq.y, sgn, r);
    }
  };
  auto tv_write = [&](float* V, const int q) {
    const int pl = q >> 1, h = q & 1;
    float* o = V + ((pl * 16 + 2 * h) * WKC) * 32 + t_wr;
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      o[(0 * 4 + jj) * (WKC * 32)] = vt[0][jj] - vt[2][jj];
      o[(1 * 4 + jj) * (WKC * 32)] = vt[1][jj] + vt[2][jj];
      o[(2 * 4 + jj) * (WKC * 32)] = vt[2][jj] - vt[1][jj];
      o[(3 * 4 + jj) * (WKC * 32)] = vt[1][jj] - vt[3][jj];
    }
  };

  f32x16 acc[6 * NRT];   // [(rt * 3 + plane) * 2 + xl]
#pragma unroll
  for (int i = 0; i < 6 * NRT; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;

  // ---- prologue: raw rows of chunks 0..2, U(0), U(1); V(0)
  dma_u(0, Us);
  dma_u(1, Us + WUS);
  dma_raw(0, Rs);
  dma_raw(1, Rs + WRB);
  dma_raw(2, Rs + 2 * WRB);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  tv_read(Rs, wave);
  tv_write(Vs, wave);
  if (wave < 2) {
    tv_read(Rs, wave + 8);
    tv_write(Vs, wave + 8);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int fu = 2 * wave * (2 * WKC * 32) + rt0 * (WKC * 32) + lane;   // xi = 2 * wave: A fragments  [kz][xi][rt][c][32]
  const int fv = 2 * wave * (WKC * 32) + lane;       //                 B fragments  [plane][xi][c][32]
  float a[3][2][NRT], bq[3][3][2];                     // fragment sets (one per kz): [set][xl][rt], [set][plane][xl]
  auto frags = [&](const float* U, const float* V, const int kz, const int set) {
#pragma unroll
    for (int xl = 0; xl < 2; xl++) {
#pragma unroll
      for (int rt = 0; rt < NRT; rt++) a[set][xl][rt] = U[((kz * 16 + xl) * 2 + rt) * (WKC * 32)];
#pragma unroll
      for (int pl = 0; pl < 3; pl++) bq[set][pl][xl] = V[((pl + kz) * 16 + xl) * (WKC * 32)];
    }
  };
  auto mfmas = [&](const int set) {
#pragma unroll
    for (int rt = 0; rt < NRT; rt++)
#pragma unroll
      for (int pl = 0; pl < 3; pl++)
#pragma unroll
        for (int xl = 0; xl < 2; xl++)
          acc[(rt * 3 + pl) * 2 + xl] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][xl][rt], bq[set][pl][xl],
                                                                             acc[(rt * 3 + pl) * 2 + xl], 0, 0, 0);
  };
  int rb = 0;   // raw buffer of chunk k
  int ub = 0;   // U buffer of chunk k
  // One chunk per iteration.  Its kz = 0 fragments are already in set 0 (prefetched).  The single barrier sits
  // BEFORE the last MFMA group: by then every wave has read chunk k's fragments into registers and written its
  // share of V(k+1), so the last 12 MFMAs run while the next chunk's first fragments are fetched and the loop
  // turns around.
  frags(Us + fu, Vs + fv, 0, 0);
  for (int k = 0; k < p.nchunks; k++) {
    const int cur = k & 1;
    const int rb1 = rb == 2 ? 0 : rb + 1;
    const int ub1 = ub == WNU - 1 ? 0 : ub + 1, ub2 = ub1 == WNU - 1 ? 0 : ub1 + 1;
    const float* U = Us + ub * WUS + fu;
    const float* V = Vs + cur * WVB + fv;
    const float* Rn = Rs + rb1 * WRB;       // raw rows of chunk k+1
    float* Vn = Vs + (cur ^ 1) * WVB;
    frags(U, V, 1, 1);
    tv_read(Rn, wave);
    dma_u(k + 2, Us + ub2 * WUS);
    dma_raw(k + 3, Rs + rb * WRB);          // chunk k's raw buffer is free (its V was made during chunk k-1)
    mfmas(0);
    tv_write(Vn, wave);
    __builtin_amdgcn_sched_barrier(0);
    frags(U, V, 2, 2);
    mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
    if (wave < 2) {
      tv_read(Rn, wave + 8);
      tv_write(Vn, wave + 8);
    }
    // U(k+1) (issued one chunk ago) and everything older has landed; this chunk's 3 U(k+2) pieces and the 4 raw-row
    // loads of chunk k+3 may stay in flight
    asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    frags(Us + ub1 * WUS + fu, Vn + fv, 0, 0);   // chunk k+1, kz = 0
    mfmas(2);
    __builtin_amdgcn_sched_barrier(0);
    rb = rb1;
    ub = ub1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stray prefetches must not land in the exchange buffers
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: Y = A^T M A with A^T = [1 1 1 0; 0 1 -1 -1].  Wave w holds xi_y = w >> 1 and the xi_x pair
  // (w & 1): it reduces its pair along x (2 values per element), the 8 waves' values meet in LDS.
  const int xh = wave & 1;
  const int er = tid >> 5, et = tid & 31;              // reader: rows er, er + 16 of the 32-row tile, tile et
  const int ety = et / G::TC, etx = et & (G::TC - 1);
  const int oy = y0 + 2 * ety, ox = x0 + 2 * etx;
  const long long TZYX = (long long)p.TZ_ * p.TY_ * p.TX_;
  const bool pair_ok = (p.TX_ & 1) == 0 && p.pair_aligned;   // (x, x+1) pairs are 8-byte aligned in every destination
  // Destinations and biases of this thread's rows are worked out ONCE (64-bit address arithmetic and the bias
  // loads would otherwise be repeated, latency exposed, in each of the 6 rounds).
  float ebias[NRT][2];  // [rt][row half]; gated: [rt][0] = feature bias, [rt][1] = gate bias
  float* eptr[NRT][2];  // voxel (z0, oy, ox) of the row's destination; nullptr = nothing to store
  const bool unsh = p.epi == SR3D_EPI_UNSHUFFLE;
  const long long pstride = (long long)p.TY_ * p.TX_ * (unsh ? 2 : 1);   // one output plane further
  const int ystride = unsh ? 2 * p.TX_ : p.TX_, xstride = unsh ? 2 : 1;
#pragma unroll
  for (int rt = 0; rt < NRT; rt++) {
    const int rbase = p.n_off + nblk * 64 + (rt0 + rt) * 32;
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
      const int n = rbase + er + 16 * hh;
      ebias[rt][hh] = 0.f, eptr[rt][hh] = nullptr;
      if (p.epi == SR3D_EPI_GATED) {
        const int co = rbase / 2 + er;   // 16 channels per 32-row tile: rows 0..15 features, 16..31 gates
        if (co < p.Cg) {
          const float* bsrc = hh ? p.bias2 : p.bias;
          ebias[rt][hh] = bsrc ? bsrc[co] : 0.f;
          eptr[rt][hh] = p.y + (((long long)b * p.Cg + co) * TZYX + ((long long)z0 * p.TY_ + oy) * p.TX_ + ox);
        }
      } else if (n < p.N) {
        ebias[rt][hh] = p.bias ? p.bias[n] : 0.f;
        if (unsh) {
          const int f = n / p.unsh_C, c = n - f * p.unsh_C;
          eptr[rt][hh] = p.y + (((long long)b * p.unsh_C + c) * TZYX +
                                ((long long)(2 * z0 + (f >> 2)) * p.TY_ + 2 * oy + ((f >> 1) & 1)) * p.TX_ + 2 * ox + (f & 1));
        } else {
          const int si = cat_find(p.out, n);
          float* base = cat_ptr(p.out, si);
          if (base != nullptr)
            eptr[rt][hh] = base + ((long long)b * cat_bstride(p.out, si) + (long long)(n - cat_cbeg(p.out, si)) * TZYX +
                                   ((long long)z0 * p.TY_ + oy) * p.TX_ + ox);
        }
      }
    }
  }
  const long long sf_off = p.save_f ? p.save_f - p.y : 0, ss_off = p.save_s ? p.save_s - p.y : 0;
  int round = 0;
#pragma unroll
  for (int rt = 0; rt < NRT; rt++)
#pragma unroll
    for (int pl = 0; pl < 3; pl++, round++) {
      float* X = lds + (round & 1) * WXB;
      {
        float* xw = X + wave * 2048;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const float m0 = acc[(rt * 3 + pl) * 2][e], m1 = acc[(rt * 3 + pl) * 2 + 1][e];
          const float s0 = xh ? m0 : m0 + m1;
          const float s1 = xh ? -m0 - m1 : m1;
          const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          const int o = (row * 32 + (lane & 31)) ^ (((row >> 2) & 1) << 5);
          xw[o] = s0, xw[1024 + o] = s1;
        }
      }
      __syncthreads();
      float yv[2][2][2];   // [row half][yo][xo]
#pragma unroll
      for (int hh = 0; hh < 2; hh++) {
        const int row = er + 16 * hh;
        const int o = (row * 32 + et) ^ (((row >> 2) & 1) << 5);
#pragma unroll
        for (int xo = 0; xo < 2; xo++) {
          float s[4];
#pragma unroll
          for (int xy = 0; xy < 4; xy++) s[xy] = X[(2 * xy) * 2048 + xo * 1024 + o] + X[(2 * xy + 1) * 2048 + xo * 1024 + o];
          yv[hh][0][xo] = (s[0] + s[1]) + s[2];
          yv[hh][1][xo] = (s[1] - s[2]) - s[3];
        }
      }
      if (z0 + pl >= p.Z) continue;   // (wave-uniform; the barrier above has been passed by everyone)
      if (p.epi == SR3D_EPI_GATED) {
        if (eptr[rt][0] != nullptr) {
#pragma unroll
          for (int yo = 0; yo < 2; yo++) {
            if (oy + yo >= p.Y) continue;
            float f[2], sg[2];
#pragma unroll
            for (int xo = 0; xo < 2; xo++) {
              f[xo] = wact(yv[0][yo][xo] + ebias[rt][0], p.act);
              sg[xo] = 1.f / (1.f + expf(-(yv[1][yo][xo] + ebias[rt][1])));
            }
            float* o = eptr[rt][0] + (pl * pstride + yo * ystride);
            if (pair_ok && ox + 1 < p.X) {
              *reinterpret_cast<f32x2*>(o) = f32x2{sg[0] * f[0], sg[1] * f[1]};
              if (p.save_f) *reinterpret_cast<f32x2*>(o + sf_off) = f32x2{f[0], f[1]};
              if (p.save_s) *reinterpret_cast<f32x2*>(o + ss_off) = f32x2{sg[0], sg[1]};
            } else {
#pragma unroll
              for (int xo = 0; xo < 2; xo++) {
                if (ox + xo >= p.X) continue;
                o[xo] = sg[xo] * f[xo];
                if (p.save_f) o[sf_off + xo] = f[xo];
                if (p.save_s) o[ss_off + xo] = sg[xo];
              }
            }
          }
        }
      } else {
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
          if (eptr[rt][hh] == nullptr) continue;
          const float bv = ebias[rt][hh];
#pragma unroll
          for (int yo = 0; yo < 2; yo++) {
            if (oy + yo >= p.Y) continue;
            float* o = eptr[rt][hh] + (pl * pstride + yo * ystride);
            const float r0 = wact(yv[hh][yo][0] + bv, p.act), r1 = wact(yv[hh][yo][1] + bv, p.act);
            if (!unsh && pair_ok && ox + 1 < p.X) {
              *reinterpret_cast<f32x2*>(o) = f32x2{r0, r1};
            } else {
              if (ox < p.X) o[0] = r0;
              if (ox + 1 < p.X) o[xstride] = r1;
            }
          }
        }
      }
    }
}

// ---- weight transform + packing: image [row block of 64][chunk][kz][xi][row tile][c][32 rows]
struct WinoPackParams {
  const float* w1;
  const float* w2;
  float* up;
  int Cout, Cin, kind, K, N, nchunks, nblk;
  int rbeg[SR3D_MAX_SRC + 1];
  int cbeg[SR3D_MAX_SRC];
};

// One thread per (row, channel, kz): it reads the 9 filter values of that plane once and writes all 16 transform
// points (the first version had one thread per image element, i.e. 16 threads re-reading the same 9 values).
__global__ __launch_bounds__(256) void wino_pack_kernel(const WinoPackParams p) {
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  const long long total = (long long)p.nblk * p.nchunks * (WUS / 16);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    long long r = e;
    const int rr = r % 32;
    r /= 32;
    const int kc = r % WKC;
    r /= WKC;
    const int rt = r % 2;
    r /= 2;
    const int kz = r % 3;
    r /= 3;
    const int chunk = r % p.nchunks;
    const int nb = r / p.nchunks;
    const int n = nb * 64 + rt * 32 + rr, k = chunk * WKC + kc;
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 9; i++) g[i / 3][i % 3] = 0.f;
    if (n < p.N && k < p.K) {
      const float* w = nullptr;  // -> w[co][ci][0][0][0]
      bool flip = false;
      if (p.kind == SR3D_PACK_FWD) {
        w = p.w1 + ((long long)n * p.Cin + k) * 27;
      } else if (p.kind == SR3D_PACK_FWD_GATED) {
        const int co = (n >> 5) * 16 + (n & 15);  // rows 0..15 of a 32-row tile: features, 16..31: gates
        if (co < p.Cout) w = ((n & 16) ? p.w2 : p.w1) + ((long long)co * p.Cin + k) * 27;
      } else {  // input gradient: rows = input channels that need a gradient, K = output channels, taps mirrored
        const int si = (n >= p.rbeg[1]) + (n >= p.rbeg[2]) + (n >= p.rbeg[3]);
        const int ci = p.cbeg[si] + (n - p.rbeg[si]);
        w = (k < p.Cout ? p.w1 + (long long)k * p.Cin * 27 : p.w2 + (long long)(k - p.Cout) * p.Cin * 27) + ci * 27;
        flip = true;
      }
      if (w != nullptr) {
        const int kzz = flip ? 2 - kz : kz;
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
          for (int kx = 0; kx < 3; kx++) g[ky][kx] = w[(kzz * 3 + (flip ? 2 - ky : ky)) * 3 + (flip ? 2 - kx : kx)];
      }
    }
    // U = G g G^T
    float t[4][3];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int kx = 0; kx < 3; kx++) t[a][kx] = G[a][0] * g[0][kx] + G[a][1] * g[1][kx] + G[a][2] * g[2][kx];
    float* dst = p.up + ((((((long long)nb * p.nchunks + chunk) * 3 + kz) * 16) * 2 + rt) * WKC + kc) * 32 + rr;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b2 = 0; b2 < 4; b2++)
        dst[(long long)(a * 4 + b2) * (2 * WKC * 32)] = t[a][0] * G[b2][0] + t[a][1] * G[b2][1] + t[a][2] * G[b2][2];
  }
}

}  // namespace

bool sr3d_wino_enabled() {
  static const bool on = getenv("SR3D_WINOGRAD") ? atoi(getenv("SR3D_WINOGRAD")) != 0 : true;
  return on;
}

size_t sr3d_wino_image_floats(int rows, int K) { return (size_t)ceil_div(rows, 64) * ceil_div(K, WKC) * WUS; }

int sr3d_wino_pack(int kind, int Cout, int Cin, int rows, int K, const float* w1, const float* w2, const int* rbeg,
                   const int* cbeg, float* image, hipStream_t st) {
  WinoPackParams p{};
  p.w1 = w1, p.w2 = w2, p.up = image;
  p.Cout = Cout, p.Cin = Cin, p.kind = kind, p.K = K, p.N = rows;
  p.nchunks = ceil_div(K, WKC), p.nblk = ceil_div(rows, 64);
  for (int i = 0; i <= SR3D_MAX_SRC; i++) p.rbeg[i] = rbeg ? rbeg[i] : INT_MAX;
  for (int i = 0; i < SR3D_MAX_SRC; i++) p.cbeg[i] = cbeg ? cbeg[i] : 0;
  const long long total = (long long)p.nblk * p.nchunks * (WUS / 16);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  SrProfScope prof(SR3D_PROF_PACK, 4.0 * ((double)rows * K * 27 + (double)p.nblk * p.nchunks * WUS), st);
  hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, st, p);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

namespace {
template <int NRT>
void wino_launch_ts(int ts, dim3 grid, hipStream_t st, const SrWinoParams& p) {
  if (ts)
    hipLaunchKernelGGL((wino_kernel<NRT, 1>), grid, dim3(WNT), kWinoLds, st, p);
  else
    hipLaunchKernelGGL((wino_kernel<NRT, 0>), grid, dim3(WNT), kWinoLds, st, p);
}
}  // namespace

int sr3d_wino_launch(SrWinoParams& p, int B, hipStream_t st) {
  // tile shape: 4 x 32 voxels, or 8 x 16 where that pads the (y, x) plane less (the 80- and 40-wide grids)
  const long long pad0 = (long long)ceil_div(p.Y, 4) * 4 * ceil_div(p.X, 32) * 32;
  const long long pad1 = (long long)ceil_div(p.Y, 8) * 8 * ceil_div(p.X, 16) * 16;
  const int ts = pad1 < pad0 ? 1 : 0;
  p.ntz = ceil_div(p.Z, WPL);
  p.nty = ceil_div(p.Y, ts ? 8 : 4), p.ntx = ceil_div(p.X, ts ? 16 : 32);
  p.nchunks = ceil_div(p.K, WKC);
  {
    uintptr_t bits = reinterpret_cast<uintptr_t>(p.y) | reinterpret_cast<uintptr_t>(p.save_f) | reinterpret_cast<uintptr_t>(p.save_s);
    for (int i = 0; i < p.out.n; i++) bits |= reinterpret_cast<uintptr_t>(p.out.ptr[i]);
    p.pair_aligned = (bits & 7) == 0;
  }
  // 64-row blocks run with two 32-row tiles per workgroup; a last block of <= 32 rows runs with one (half the MFMAs)
  const int nfull = p.N / 64, rem = p.N - nfull * 64;
  const int n2 = nfull + (rem > 32 ? 1 : 0), n1 = (rem > 0 && rem <= 32) ? 1 : 0;
  const long long nsp = (long long)p.ntz * p.nty * p.ntx;
  SR3D_CHECK(nsp * (n2 + n1) < (1ll << 31) && B <= 65535, SR3D_E_ARG, "winograd conv: grid too large");
  SR3D_CHECK((long long)p.Z * p.Y * p.X < (1ll << 29), SR3D_E_ARG, "winograd conv: more than 2^29 voxels per channel");
  SR3D_CHECK(p.K <= WCT, SR3D_E_ARG, "winograd conv: more than %d input channels", WCT);
  static SrPerDevice setup;   // (the attribute is per device, not per thread)
  if (int rc = setup.once([&]() -> int {
        SR3D_HIP(hipFuncSetAttribute((const void*)wino_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLds));
        SR3D_HIP(hipFuncSetAttribute((const void*)wino_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLds));
        SR3D_HIP(hipFuncSetAttribute((const void*)wino_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLds));
        SR3D_HIP(hipFuncSetAttribute((const void*)wino_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLds));
        return SR3D_OK;
      }))
    return rc;
  void* tok = nullptr;
  if (sr3d_prof_active()) {
    const double rows = p.epi == SR3D_EPI_GATED ? 2.0 * p.Cg : (double)p.N;
    sr3d_prof_begin(SR3D_PROF_IGEMM_S1, 2.0 * 27 * p.K * rows * (double)p.Z * p.Y * p.X * B, st, &tok);
  }
  p.nimg = n2 + n1;
  // Small grids (levels 3-4 of the U-Net) give fewer 64-row workgroups than the chip has CUs: run every 32-row tile
  // as its own workgroup instead (same image, twice the workgroups, each with half the MFMAs).
  const int ntile32 = ceil_div(p.N, 32);
  if (nsp * (n2 + n1) < 160 && ntile32 > n2 + n1) {
    p.nblk = ntile32, p.nb_off = 0, p.rt_split = 1;
    wino_launch_ts<1>(ts, dim3((unsigned)(nsp * ntile32), B), st, p);
  } else {
    p.rt_split = 0;
    if (n2 > 0) {
      p.nblk = n2, p.nb_off = 0;
      wino_launch_ts<2>(ts, dim3((unsigned)(nsp * n2), B), st, p);
    }
    if (n1 > 0) {
      p.nblk = 1, p.nb_off = n2;
      wino_launch_ts<1>(ts, dim3((unsigned)nsp, B), st, p);
    }
  }
  sr3d_prof_end(tok, st);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}
