// Weight gradient of a stride-1 3x3x3 convolution with FEW input channels (Cin <= 5: `conv0`, 5 -> 2 x 64) -- or, with the roles of
// x and dY exchanged, FEW output rows (`last`, 69 -> 4; sr3d_hwgrad_fc(..., swapped)) -- in fp32 or bf16 storage, on the split-f16
// scheme of sr3d_hwgrad.hip (fp32 operands as two fp16 halves, three v_mfma_f32_16x16x32_f16 per product group):
//
//   dW[n][c][kz,ky,kx] = sum_{b,z,y,x'} dY[n][z][y][x'] * X[c][z + kz - 1][y + ky - 1][x' + kx - 1]
//
// sr3d_hwgrad.hip tiles the channels in blocks of 32 (16-channel MFMA tiles): 5 channels would fill a sixth of it, and the
// layer ran on the fp32-MFMA kernel (3.8 ms for 0.28 TFLOP, at full resolution).  Here the 16 columns of an MFMA tile are
// the (channel, kx) PAIRS -- 15 of 16 used -- so the x shift of the tap moves into the operand with the few channels:
// X rows are staged three times, shifted by -1, 0, +1 (one split, the copies made with v_alignbit), dY rows once, and
// every fragment is one aligned ds_read_b128.  One MFMA reduces over the 32 voxels of a row segment.
//
// Workgroup = 4 waves: 64 rows n x 32 voxels of x x a range of (b, z, y) rows (split-K); wave w owns the 16-row tile w and
// all 9 (kz, ky) taps: 27 MFMAs per step.  It marches along y like sr3d_hwgrad.hip (3 planes x 4 y slots of X rows, dY
// rows double-buffered, one barrier per step).  The kernel is bound by the dY stream (512 B per row and step against 27
// MFMAs per wave): the loads run NS = 3 steps ahead (a ring of pieces in registers) and two workgroups share a CU, ~96 KB
// in flight per CU.  Scales, sign alternation and the deterministic slab reduction are those of sr3d_hwgrad.hip.
#include "sr3d_split_f16.h"

#include <limits.h>
#include <stdlib.h>

#include <utility>

namespace {

constexpr int FPITCH = 96;                     // bytes per 32-voxel fp16 row in LDS (sr3d_hwgrad.hip: conflict-free fragment reads)
// BF = true (bf16 storage, round 4): one part, no scales, one v_mfma_f32_16x16x32_bf16 per tile; a dY piece goes to LDS as it
// was loaded, the three shifted X columns are made with v_alignbit on the packed pairs (as in sr3d_hwgrad.hip)
// NW waves = 16 NW rows n per workgroup: 4 (64 rows), or 5 (80 rows: the SWAPPED form of `last`, whose 69 input channels
// play the rows -- see sr3d_hwgrad_fc)
template <int NW, bool BF = false>
struct FcGeo {
  static constexpr int NP = BF ? 1 : 2;                    // operand parts
  static constexpr int NT = 64 * NW;
  static constexpr int NB = 16 * NW;                       // rows n per workgroup
  static constexpr int XROW = NP * 16 * FPITCH;            // one X row: [part][column (c, kx)][PITCH]
  static constexpr int XBYTES = 12 * XROW;                 // 3 planes x 4 y slots
  static constexpr int DROW = NP * NB * FPITCH;            // one dY row: [part][n][PITCH]
  static constexpr size_t LDS = XBYTES + 2 * (size_t)DROW;
};
#ifndef HWGRAD_FC_NS
#define HWGRAD_FC_NS 3
#endif
constexpr int FNS = HWGRAD_FC_NS;                         // steps the loads run ahead
static_assert(2 * FcGeo<5, false>::LDS <= 160 * 1024, "two workgroups per CU");

template <class F, int... I>
__device__ __forceinline__ void fc_static_for(F& f, const int t0, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}, t0), ...);
}

__host__ __device__ inline int fc_scale_exp_of(float amax) {
  const int s = split_scale_exp(amax);
  return s == kSplitScaleNone ? 0 : s;
}

struct FcParams {
  ChanCat x, dy;
  int C, N;                  // channels (<= 5), rows
  int B, Z, Y, X;
  int nnb, nseg, S;          // row blocks, x segments, splits
  long long rows_per_split;  // (b, z, y) rows per split
  int Npad, Cpad;
  float* slab;               // [S * nseg][27][Npad][Cpad]
  const float* amax;         // [0..3] = max|x slice i|, [4..7] = max|dy slice i|
};

template <int NW, bool BF>
__global__ __launch_bounds__(64 * NW, 2) void hwgrad_fc_kernel(const FcParams p) {
  using G = FcGeo<NW, BF>;
  constexpr int FNT = G::NT, FNB = G::NB, FDROW = G::DROW, FXROW = G::XROW, FXBYTES = G::XBYTES;
  constexpr int ESZ = BF ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Xs = lds;
  unsigned char* Ds = lds + FXBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < FNT / 64);

  int v = blockIdx.x;
  const int nb = v % p.nnb;
  v /= p.nnb;
  const int seg = v % p.nseg;
  const int split = v / p.nseg;
  const int x0 = seg * 32;
  const long long YX = (long long)p.Y * p.X, ZYX = YX * p.Z;

  // ---- staging roles: every thread one dY item (row tid / 4, 8-voxel piece tid & 3); the first 12 C threads (wave 0) also
  // one X item (plane dz, channel c, piece q)
  const int d_n = tid >> 2, d_q = tid & 3;
  const unsigned char* d_src = nullptr;   // (bytes; element size ESZ)
  long long d_b = 0;
  float md = 1.f;
  bool d_on = false;
  {
    const int n = nb * FNB + d_n;
    if (n < p.N) {
      const int si = cat_find(p.dy, n);
      d_src = reinterpret_cast<const unsigned char*>(cat_ptr(p.dy, si)) + (long long)(n - cat_cbeg(p.dy, si)) * ZYX * ESZ;
      d_b = cat_bstride(p.dy, si);
      if constexpr (!BF) md = ldexpf(1.f, fc_scale_exp_of(p.amax[4 + si]));
      d_on = x0 + 8 * d_q < p.X;
    }
  }
  const bool is_x = tid < 12 * p.C;
  int x_dz = 0, x_c = 0, x_q = 0;
  const unsigned char* x_src = nullptr;
  long long x_b = 0;
  float mx = 1.f;
  bool x_on = false;
  if (is_x) {
    x_dz = tid / (4 * p.C), x_c = (tid >> 2) % p.C, x_q = tid & 3;
    const int si = cat_find(p.x, x_c);
    x_src = reinterpret_cast<const unsigned char*>(cat_ptr(p.x, si)) + (long long)(x_c - cat_cbeg(p.x, si)) * ZYX * ESZ;
    x_b = cat_bstride(p.x, si);
    if constexpr (!BF) mx = ldexpf(1.f, fc_scale_exp_of(p.amax[si]));
    x_on = x0 + 8 * x_q < p.X;
  }
  const int dxq = x0 + 8 * d_q, xxq = x0 + 8 * x_q;

  // pieces in flight, kept AS LOADED (two quads; X: and the two neighbours apart): an array with the elements at other positions
  // (px[10] = elements -1 .. 8, the quads at 1 .. 8) made hipcc load into temporaries and move them -- behind an s_waitcnt vmcnt(0) right after the
  // loads were issued: wave 0 (the X stager) sat out the full memory latency in EVERY step, and the other waves at the barrier
  // with it (2.9 us per step: conv0's weight gradient ran at 1.4 TB/s of its dY stream)
  float pd[FNS][8];                    // dY: elements 0..7
  float pxv[FNS][10];                  // X: elements 0..7 as loaded, then element -1 ([8]) and element 8 ([9])
#pragma unroll
  for (int u = 0; u < FNS; u++) {
#pragma unroll
    for (int j = 0; j < 8; j++) pd[u][j] = pxv[u][j] = 0.f;
    pxv[u][8] = pxv[u][9] = 0.f;
  }
  // (bf16: pd[u][0..3] / pxv[u][0..3] hold the 8 elements as 4 packed dwords, pxv[u][8] / [9] the neighbours' 16 bits)
  auto load_dy = [&](const int u, const int b, const int z, const int y) {
    const bool ok = d_on && (unsigned)z < (unsigned)p.Z && (unsigned)y < (unsigned)p.Y;
    if (ok) {
      const long long off = (long long)b * d_b + (long long)z * YX + (long long)y * p.X + dxq;
      if constexpr (BF) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(d_src + off * 2);
        pd[u][0] = a.x, pd[u][1] = a.y, pd[u][2] = a.z, pd[u][3] = a.w;
      } else {
        const float* r = reinterpret_cast<const float*>(d_src) + off;
        const f32x4 a = *reinterpret_cast<const f32x4*>(r), c4 = *reinterpret_cast<const f32x4*>(r + 4);
        pd[u][0] = a.x, pd[u][1] = a.y, pd[u][2] = a.z, pd[u][3] = a.w, pd[u][4] = c4.x, pd[u][5] = c4.y, pd[u][6] = c4.z, pd[u][7] = c4.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) pd[u][j] = 0.f;
    }
  };
  auto load_x = [&](const int u, const int b, const int z, const int y) {
    const bool ok = x_on && (unsigned)z < (unsigned)p.Z && (unsigned)y < (unsigned)p.Y;
    if (ok) {
      const long long off = (long long)b * x_b + (long long)z * YX + (long long)y * p.X + xxq;
      if constexpr (BF) {
        const unsigned short* r = reinterpret_cast<const unsigned short*>(x_src) + off;
        const f32x4 a = *reinterpret_cast<const f32x4*>(r);
        pxv[u][0] = a.x, pxv[u][1] = a.y, pxv[u][2] = a.z, pxv[u][3] = a.w;
        pxv[u][8] = __builtin_bit_cast(float, xxq > 0 ? (unsigned)r[-1] : 0u);
        pxv[u][9] = __builtin_bit_cast(float, xxq + 8 < p.X ? (unsigned)r[8] : 0u);
      } else {
        const float* r = reinterpret_cast<const float*>(x_src) + off;
        const f32x4 a = *reinterpret_cast<const f32x4*>(r), c4 = *reinterpret_cast<const f32x4*>(r + 4);
        pxv[u][0] = a.x, pxv[u][1] = a.y, pxv[u][2] = a.z, pxv[u][3] = a.w, pxv[u][4] = c4.x, pxv[u][5] = c4.y, pxv[u][6] = c4.z, pxv[u][7] = c4.w;
        pxv[u][8] = xxq > 0 ? r[-1] : 0.f;
        pxv[u][9] = xxq + 8 < p.X ? r[8] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 10; j++) pxv[u][j] = 0.f;
    }
  };
  // dY row -> buffer dbuf; X row -> y slot `xslot` of its plane, columns (c, kx = 0, 1, 2)
  auto write_dy = [&](const int u, const int dbuf, const float dsign) {
    unsigned char* d = Ds + dbuf * FDROW + d_n * FPITCH + d_q * 16;
    if constexpr (BF) {
      const unsigned sm = dsign < 0.f ? 0x80008000u : 0u;   // sign flip of both bf16 halves
      *reinterpret_cast<u32x4*>(d) = u32x4{__builtin_bit_cast(unsigned, pd[u][0]) ^ sm, __builtin_bit_cast(unsigned, pd[u][1]) ^ sm,
                                           __builtin_bit_cast(unsigned, pd[u][2]) ^ sm, __builtin_bit_cast(unsigned, pd[u][3]) ^ sm};
      return;
    }
    unsigned h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; k++) split_pair(pd[u][2 * k], pd[u][2 * k + 1], md * dsign, h[k], l[k]);
    *reinterpret_cast<u32x4*>(d) = u32x4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(d + FNB * FPITCH) = u32x4{l[0], l[1], l[2], l[3]};
  };
  auto write_x = [&](const int u, const int xslot) {
    unsigned char* d = Xs + (x_dz * 4 + xslot) * FXROW + (x_c * 3) * FPITCH + x_q * 16;
    if constexpr (BF) {   // column kx holds X[x' + kx - 1]: kx = 0: elements -1 .. 6; kx = 1: 0 .. 7; kx = 2: 1 .. 8
      const unsigned d0 = __builtin_bit_cast(unsigned, pxv[u][0]), d1 = __builtin_bit_cast(unsigned, pxv[u][1]);
      const unsigned d2 = __builtin_bit_cast(unsigned, pxv[u][2]), d3 = __builtin_bit_cast(unsigned, pxv[u][3]);
      const unsigned pl = __builtin_bit_cast(unsigned, pxv[u][8]) << 16, nh = __builtin_bit_cast(unsigned, pxv[u][9]);
      *reinterpret_cast<u32x4*>(d) = u32x4{__builtin_amdgcn_alignbit(d0, pl, 16), __builtin_amdgcn_alignbit(d1, d0, 16),
                                           __builtin_amdgcn_alignbit(d2, d1, 16), __builtin_amdgcn_alignbit(d3, d2, 16)};
      *reinterpret_cast<u32x4*>(d + FPITCH) = u32x4{d0, d1, d2, d3};
      *reinterpret_cast<u32x4*>(d + 2 * FPITCH) = u32x4{__builtin_amdgcn_alignbit(d1, d0, 16), __builtin_amdgcn_alignbit(d2, d1, 16),
                                                        __builtin_amdgcn_alignbit(d3, d2, 16), __builtin_amdgcn_alignbit(nh, d3, 16)};
      return;
    }
    // the 10 elements -1 .. 8 are split once, as the pairs (-1, 0), (1, 2), ..., (7, 8); column kx holds X[x' + kx - 1]:
    //   kx = 0: elements -1 .. 6 = pairs 0 .. 3;  kx = 2: elements 1 .. 8 = pairs 1 .. 4;
    //   kx = 1: elements 0 .. 7 = the high half of pair k with the low half of pair k + 1 (v_alignbit)
    unsigned ph[5], pl[5];
    split_pair(pxv[u][8], pxv[u][0], mx, ph[0], pl[0]);
#pragma unroll
    for (int k = 1; k < 4; k++) split_pair(pxv[u][2 * k - 1], pxv[u][2 * k], mx, ph[k], pl[k]);
    split_pair(pxv[u][7], pxv[u][9], mx, ph[4], pl[4]);
    *reinterpret_cast<u32x4*>(d) = u32x4{ph[0], ph[1], ph[2], ph[3]};
    *reinterpret_cast<u32x4*>(d + 16 * FPITCH) = u32x4{pl[0], pl[1], pl[2], pl[3]};
    u32x4 mh, ml;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      mh[k] = __builtin_amdgcn_alignbit(ph[k + 1], ph[k], 16);
      ml[k] = __builtin_amdgcn_alignbit(pl[k + 1], pl[k], 16);
    }
    *reinterpret_cast<u32x4*>(d + FPITCH) = mh;
    *reinterpret_cast<u32x4*>(d + FPITCH + 16 * FPITCH) = ml;
    *reinterpret_cast<u32x4*>(d + 2 * FPITCH) = u32x4{ph[1], ph[2], ph[3], ph[4]};
    *reinterpret_cast<u32x4*>(d + 2 * FPITCH + 16 * FPITCH) = u32x4{pl[1], pl[2], pl[3], pl[4]};
  };

  // columns 3 C .. 15 of every X row are never written: zero them once (they are multiplied, their results dropped)
  for (int i = tid; i < FXBYTES / 16; i += FNT) reinterpret_cast<u32x4*>(Xs)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  f32x4 acc[3][3];   // [kz][ky]: 16 rows x 16 columns (c, kx)
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int k = 0; k < 3; k++) acc[a][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float acc_sign = 1.f;
  const int fr = (lane & 15) * FPITCH + (lane >> 4) * 16;   // fragment: row / column lane & 15, voxels 8 (lane >> 4) .. +8
  const int fa = wave * 16 * FPITCH + fr;

  const long long rows_total = (long long)p.B * p.Z * p.Y;
  long long r0 = (long long)split * p.rows_per_split, r1 = r0 + p.rows_per_split;
  if (r1 > rows_total) r1 = rows_total;
  while (r0 < r1) {
    const long long plane = r0 / p.Y;                 // (b, z)
    const int b = (int)(plane / p.Z), z = (int)(plane - (long long)b * p.Z);
    const int ya = (int)(r0 - plane * p.Y);
    const long long pend = (plane + 1) * p.Y;
    const int yb = (int)((r1 < pend ? r1 : pend) - plane * p.Y);
    // step t: write what was loaded in step t - NS (X rows t + 2, dY row t + 1) from ring entry u, refill that entry with
    // the rows of step t + NS, multiply row t
    // (the ring entry u as a compile-time constant of a generic lambda: with a loop variable -- `#pragma unroll` is only a
    //  request -- the small per-entry arrays went to scratch memory)
    auto step = [&](auto U, const int t0) {
      {
        constexpr int u = decltype(U)::value;
        const int t = t0 + u;
        if (t >= yb) return;
        if (t > ya - 4) {
          const long long rr = plane * p.Y + (t + 1);
          write_dy(u, (t + 1) & 1, ((rr >> 5) & 1) ? -1.f : 1.f);
          if (is_x) write_x(u, (t + 2) & 3);
        }
        load_dy(u, b, z, (t + 1 + FNS >= ya && t + 1 + FNS < yb) ? t + 1 + FNS : -1);
        if (is_x) load_x(u, b, z + x_dz - 1, t + 2 + FNS);
        if (t >= ya) {
          const long long rr = plane * p.Y + t;
          const float sgn = ((rr >> 5) & 1) ? -1.f : 1.f;
          if (sgn != acc_sign) {   // (wave-uniform) sign alternation, see sr3d_hwgrad.hip
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
              for (int k = 0; k < 3; k++) acc[a][k] = -acc[a][k];
            acc_sign = sgn;
          }
          const unsigned char* da = Ds + (t & 1) * FDROW + fa;
          const h8 ah = *reinterpret_cast<const h8*>(da);
          h8 al = ah;
          if constexpr (!BF) al = *reinterpret_cast<const h8*>(da + FNB * FPITCH);
#pragma unroll
          for (int a = 0; a < 3; a++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
              const unsigned char* xb = Xs + (a * 4 + ((t + k - 1) & 3)) * FXROW + fr;
              const h8 bh = *reinterpret_cast<const h8*>(xb);
              if constexpr (BF) {
                acc[a][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, ah), __builtin_bit_cast(bf8, bh), acc[a][k], 0, 0, 0);
              } else {
                const h8 bl = *reinterpret_cast<const h8*>(xb + 16 * FPITCH);
                acc[a][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[a][k], 0, 0, 0);
                acc[a][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[a][k], 0, 0, 0);
                acc[a][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[a][k], 0, 0, 0);
              }
            }
        }
        // (not __syncthreads(): that would drain vmcnt and expose the latency of the loads issued above in every step)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    };
    for (int t0 = ya - 3 - FNS; t0 < yb; t0 += FNS) fc_static_for(step, t0, std::make_integer_sequence<int, FNS>{});
    r0 = plane * p.Y + yb;
  }

  // ---- partial block -> slab[split, segment][tap][n][c]; 16x16 tile: column = lane & 15 = 3 c + kx, row = 4 (lane >> 4) + register
  const int col = lane & 15, c = col / 3, kx = col - 3 * c;
  if (c < p.C) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int tap = (a * 3 + k) * 3 + kx;
        float* out = p.slab + ((long long)(split * p.nseg + seg) * 27 + tap) * p.Npad * p.Cpad;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int n = nb * FNB + wave * 16 + 4 * (lane >> 4) + r;
          out[(long long)n * p.Cpad + c] = acc[a][k][r] * acc_sign;
        }
      }
  }
}

__global__ __launch_bounds__(64) void fc_gather_amax_kernel(const unsigned* xs, int nx, const unsigned* ds, int nd, unsigned* amax) {
  const int lane = threadIdx.x;
  for (int i = 0; i < nx + nd; i++) {
    const unsigned* src = i < nx ? (xs ? xs + i * 64 : nullptr) : (ds ? ds + (i - nx) * 64 : nullptr);
    if (src == nullptr) continue;
    float m = __uint_as_float(src[lane]);   // (bits of non-negative floats order like unsigned integers)
    m = split_wave_max(m);
    if (lane == 0) amax[i < nx ? i : 4 + (i - nx)] = __float_as_uint(m);
  }
}

struct FcSliceMap {
  int xcb[SR3D_MAX_SRC], dcb[SR3D_MAX_SRC];   // first channel / row of every slice (INT_MAX: unused)
};

// dW[n][c][tap] = 2^-(sx(c)+sd(n)) * sum_s slab[s][tap][n][c]  (fixed order: deterministic)
// swapped: the kernel's rows n were the layer's INPUT channels and its few channels c the layer's output rows, with the taps
// mirrored (sr3d_hwgrad_fc): dW[c][n][26 - tap]
__global__ __launch_bounds__(256) void hwgrad_fc_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int N, int C,
                                                               int ldc, int Npad, int Cpad, const float* amax, const FcSliceMap sm,
                                                               int swapped) {
  const long long plane = (long long)Npad * Cpad;
  const long long total = (long long)N * C * 27;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const long long r = e / C;
    const int n = (int)(r % N), tap = (int)(r / N);
    const int xi = (c >= sm.xcb[1]) + (c >= sm.xcb[2]) + (c >= sm.xcb[3]), di = (n >= sm.dcb[1]) + (n >= sm.dcb[2]) + (n >= sm.dcb[3]);
    const float mult = amax ? ldexpf(1.f, -(fc_scale_exp_of(amax[xi]) + fc_scale_exp_of(amax[4 + di]))) : 1.f;   // (bf16: unscaled)
    const float* s0 = slab + (long long)tap * plane + (long long)n * Cpad + c;
    float s = 0.f;
    int k = 0;
    for (; k + 4 <= S; k += 4) {
      const float v0 = s0[(long long)k * 27 * plane], v1 = s0[(long long)(k + 1) * 27 * plane];
      const float v2 = s0[(long long)(k + 2) * 27 * plane], v3 = s0[(long long)(k + 3) * 27 * plane];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; k < S; k++) s += s0[(long long)k * 27 * plane];
    if (swapped)
      dw[((long long)c * ldc + n) * 27 + (26 - tap)] = s * mult;
    else
      dw[((long long)n * ldc + c) * 27 + tap] = s * mult;
  }
}

struct FcPlan {
  int nnb, nseg, S, Npad, Cpad;
  long long rows_per_split;
};

// n_many: rows of the kernel (the layer's rows, or its input channels in the swapped form); nw: waves per workgroup
FcPlan fc_plan(const sr3d_conv_desc_t* d, int n_many, int nw) {
  FcPlan g;
  g.nnb = ceil_div(n_many, 16 * nw), g.nseg = ceil_div(d->X, 32);
  g.Npad = g.nnb * 16 * nw, g.Cpad = 8;
  const long long rows = (long long)d->B * d->Z * d->Y;
  const long long cols = (long long)g.nnb * g.nseg;
  // ~4 rounds over the 512 workgroup slots of the chip, at least 24 rows per split (warm-up steps per plane segment)
  long long S = ceil_div(2048, cols);
  const long long smax = rows / 24 > 0 ? rows / 24 : 1;
  S = S < 1 ? 1 : (S > smax ? smax : S);
  g.rows_per_split = (rows + S - 1) / S;
  g.S = (int)((rows + g.rows_per_split - 1) / g.rows_per_split);
  return g;
}

}  // namespace

// swapped form (few OUTPUT rows, `last`: 69 -> 4): the roles of x and dY exchanged, see sr3d_hwgrad_fc
inline int fc_swapped_nw(int cin) { return cin > 64 && cin <= 80 ? 5 : 4; }

size_t sr3d_hwgrad_fc_ws_bytes(const sr3d_conv_desc_t* d, int n_total, bool swapped) {
  const FcPlan g = swapped ? fc_plan(d, d->Cin, fc_swapped_nw(d->Cin)) : fc_plan(d, n_total, 4);
  return 256 + (size_t)g.S * g.nseg * 27 * g.Npad * g.Cpad * 4;
}

bool sr3d_hwgrad_fc_ok(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy, int n_total, bool swapped) {
  if (d->stride != 1 || d->X % 8 != 0) return false;
  if (swapped ? n_total > 5 : d->Cin > 5) return false;
  for (int i = 0; i < x.n; i++)
    if (reinterpret_cast<uintptr_t>(x.ptr[i]) & 15) return false;
  for (int i = 0; i < dy.n; i++)
    if (reinterpret_cast<uintptr_t>(dy.ptr[i]) & 15) return false;
  return true;
}

// dW[n][c][tap] = sum_v dY[n][v] X[c][v + tap - 1].  swapped = false: few input channels (conv0), x plays the kernel's X.
// swapped = true (round 4): few OUTPUT rows (`last`, 69 -> 4: the split weight-gradient kernel used 4 of the 32 rows of its
// blocks, 3.6 ms).  With v' = v + tap - 1 the same sum is  sum_v' X[c][v'] dY[n][v' + (2 - tap) - 1]  -- zero padding on either
// side gives the same terms --: the kernel runs with the layer's input channels as its ROWS and dY as its few channels, and the
// reduce kernel writes dW[c'][n'][26 - tap'].
int sr3d_hwgrad_fc(const sr3d_conv_desc_t* d, const ChanCat& x_real, const ChanCat& dy_real, int n_total, float* dw, float* ws, hipStream_t st,
                   const unsigned* x_absmax_real, const unsigned* dy_absmax_real, bool swapped) {
  const ChanCat& x = swapped ? dy_real : x_real;     // the kernel's few-channel operand
  const ChanCat& dy = swapped ? x_real : dy_real;    // the kernel's row operand
  const unsigned* x_absmax = swapped ? dy_absmax_real : x_absmax_real;
  const unsigned* dy_absmax = swapped ? x_absmax_real : dy_absmax_real;
  const int C = swapped ? n_total : d->Cin, N = swapped ? d->Cin : n_total;
  const int nw = swapped ? fc_swapped_nw(d->Cin) : 4;
  const FcPlan g = fc_plan(d, N, nw);
  const bool bf = d->dtype == SR3D_DTYPE_BF16;
  unsigned* amax = (unsigned*)ws;
  if (!bf) {
    if (int rc = sr3d_zero_words(amax, 64, st)) return rc;
    SrProfScope prof(SR3D_PROF_DATA, 0.0, st);
    if (x_absmax == nullptr)
      for (int i = 0; i < x.n; i++)
        if (int rc = sr3d_absmax_launch(x.ptr[i], (long long)d->B * x.bstride[i], amax + i, st)) return rc;
    if (dy_absmax == nullptr)
      for (int i = 0; i < dy.n; i++)
        if (int rc = sr3d_absmax_launch(dy.ptr[i], (long long)d->B * dy.bstride[i], amax + 4 + i, st)) return rc;
    if (x_absmax != nullptr || dy_absmax != nullptr) {
      hipLaunchKernelGGL(fc_gather_amax_kernel, dim3(1), dim3(64), 0, st, x_absmax, x.n, dy_absmax, dy.n, amax);
      SR3D_HIP(hipGetLastError());
    }
  }
  static SrPerDevice setup;   // (the attribute is per device, not per thread)
  if (int rc = setup.once([&]() -> int {
        SR3D_HIP(hipFuncSetAttribute((const void*)hwgrad_fc_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FcGeo<4, false>::LDS));
        SR3D_HIP(hipFuncSetAttribute((const void*)hwgrad_fc_kernel<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FcGeo<5, false>::LDS));
        SR3D_HIP(hipFuncSetAttribute((const void*)hwgrad_fc_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FcGeo<4, true>::LDS));
        SR3D_HIP(hipFuncSetAttribute((const void*)hwgrad_fc_kernel<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FcGeo<5, true>::LDS));
        return SR3D_OK;
      }))
    return rc;
  FcParams p{};
  p.x = x, p.dy = dy, p.C = C, p.N = N;
  p.B = d->B, p.Z = d->Z, p.Y = d->Y, p.X = d->X;
  p.nnb = g.nnb, p.nseg = g.nseg, p.S = g.S, p.rows_per_split = g.rows_per_split;
  p.Npad = g.Npad, p.Cpad = g.Cpad;
  p.slab = ws + 64, p.amax = bf ? nullptr : (const float*)amax;
  const long long nwg = (long long)g.nnb * g.nseg * g.S;
  SR3D_CHECK(nwg < (1ll << 31), SR3D_E_ARG, "few-channel weight gradient: grid too large");
  {
    SrProfScope prof(SR3D_PROF_WGRAD, 2.0 * 27 * d->Cin * (double)n_total * (double)d->Z * d->Y * d->X * d->B, st);
    constexpr size_t l4 = FcGeo<4, false>::LDS, l5 = FcGeo<5, false>::LDS, l4b = FcGeo<4, true>::LDS, l5b = FcGeo<5, true>::LDS;
    if (bf) {
      if (nw == 5)
        hipLaunchKernelGGL((hwgrad_fc_kernel<5, true>), dim3((unsigned)nwg), dim3(320), l5b, st, p);
      else
        hipLaunchKernelGGL((hwgrad_fc_kernel<4, true>), dim3((unsigned)nwg), dim3(256), l4b, st, p);
    } else if (nw == 5) {
      hipLaunchKernelGGL((hwgrad_fc_kernel<5, false>), dim3((unsigned)nwg), dim3(320), l5, st, p);
    } else {
      hipLaunchKernelGGL((hwgrad_fc_kernel<4, false>), dim3((unsigned)nwg), dim3(256), l4, st, p);
    }
    SR3D_HIP(hipGetLastError());
  }
  SrProfScope prof(SR3D_PROF_PACK, 4.0 * ((double)g.S * g.nseg + 1) * 27 * g.Npad * g.Cpad, st);
  const long long total = (long long)N * C * 27;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  FcSliceMap sm;
  for (int i = 0; i < SR3D_MAX_SRC; i++) sm.xcb[i] = x.cbeg[i], sm.dcb[i] = dy.cbeg[i];
  hipLaunchKernelGGL(hwgrad_fc_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)p.slab, dw, g.S * g.nseg, N, C,
                     d->Cin, g.Npad, g.Cpad, bf ? (const float*)nullptr : (const float*)amax, sm, swapped ? 1 : 0);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}
