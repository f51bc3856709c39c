// Weight gradient of the STRIDE-2 3x3x3 convolutions (DownBlock's first layer, pytorch/model/unet.py:27-36) on the f16 /
// bf16 MFMA, the scheme of sr3d_hwgrad.hip (fp32 operands as two fp16 halves, three v_mfma_f32_16x16x32_f16 per tile; or
// bf16 storage, one v_mfma_f32_16x16x32_bf16):
//
//   dW[n][c][kz,ky,kx] = sum_{b,oz,oy,ox} dY[n][oz][oy][ox] * X[c][2 oz + kz - 1][2 oy + ky - 1][2 ox + kx - 1]
//
// One MFMA reduces over the K = 32 consecutive ox of a row segment: A = a dY row segment (16 rows n), B = an X row (16
// channels c) sampled at every second voxel.  An X row of 64 fine voxels is therefore staged DE-INTERLEAVED, as its even
// and its odd sub-row (Xe[q] = X[2q], Xo[q] = X[2q + 1]):
//   kx = 1:  X[2 ox]     = Xe[ox]        B = even sub-row,  A = dY
//   kx = 2:  X[2 ox + 1] = Xo[ox]        B = odd sub-row,   A = dY
//   kx = 0:  X[2 ox - 1] = Xo[ox - 1]    sum_ox dY[ox] Xo[ox - 1] = sum_q dY[q + 1] Xo[q]:  B = odd sub-row, A = dY SHIFTED by one
// so every fragment stays one aligned ds_read_b128 (the shifted operand is the small one, dY, staged twice).
// The (kz, ky) taps pick WHICH X row: output row (oz, oy) reads the planes 2 oz - 1 .. 2 oz + 1 and the rows 2 oy - 1 ..
// 2 oy + 1.  A workgroup marches along oy: per step TWO new X rows of each of the 3 planes (row 2 oy + 1 is shared with the
// next step; 5 row slots per plane in LDS) and one dY row (2 copies, double-buffered).
//
// Workgroup = 12 waves: (64 or 32 rows n) x (16 channels c) x (32 coarse voxels) x (a range of (b, oz, oy) rows: split-K);
// waves 0..8 own the taps (kz, ky) = (w / 3, w % 3) x kx 0, 1; waves 9..11 kx = 2 of three (kz, ky) groups each -- as in
// sr3d_hwgrad.hip.  Staging: waves 0-5 one X item each (plane, new row, channel, 16-fine-voxel piece), waves 6-9 the dY
// items.  Scales (split form): one power of two per tensor slice from max |.| (exported by the producing kernels or swept);
// sign flip of the dY rows and the accumulators every 32 rows against the MFMA's truncation bias; partial blocks to a slab,
// summed in a fixed order (bit-reproducible).
// It replaces the fp32-MFMA direct kernel (sr3d_wgrad.hip: 97-100 TFLOP/s; 76 reading bf16) for the stride-2 layers whose
// coarse row length is a multiple of 8; other shapes keep the direct kernel.
#include "sr3d_split_f16.h"

#include <limits.h>
#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int WNT = 768;
constexpr int PITCH = 96;    // bytes per 32-voxel row in LDS: conflict-free ds_read_b128 (see sr3d_hwgrad.hip)
constexpr int CB = 16;       // channels per workgroup

template <int RT, bool BF>
struct SGeo {
  static constexpr int NP = BF ? 1 : 2;                  // operand parts
  static constexpr int XSUB = CB * PITCH;                // one parity sub-row of one part
  static constexpr int XROW = NP * 2 * XSUB;             // an X row: [part][parity][c][PITCH]
  static constexpr int XBYTES = 15 * XROW;               // 3 planes x 5 row slots
  static constexpr int DCOPY = NP * 32 * RT * PITCH;     // one copy of a dY row: [part][n][PITCH]
  static constexpr int DROW = 2 * DCOPY;                 // copy 0 = dY[q + 1], copy 1 = dY[q]
  static constexpr size_t LDS = XBYTES + 2 * (size_t)DROW;
  static constexpr int NDY = 32 * RT * 4;
};
static_assert(SGeo<2, false>::LDS <= 160 * 1024, "LDS budget");

__host__ __device__ inline int scale_exp_of(float amax) {
  const int s = split_scale_exp(amax);
  return s == kSplitScaleNone ? 0 : s;
}

struct Hw2Params {
  ChanCat x, dy;
  int cu, N;
  int B, IZ, IY, IX, OZ, OY, OX;
  int nnb, ncb, nseg, S;
  long long rows_per_split;
  int Npad, Cpad;
  float* slab;               // [S * nseg][27][Npad][Cpad]
  const float* amax;         // [0..3] = max|x slice i|, [4..7] = max|dy slice i|
};

template <int RT, bool BF>
__global__ __launch_bounds__(WNT) void hwgrad_s2_kernel(const Hw2Params p) {
  using G = SGeo<RT, BF>;
  constexpr int NP = G::NP, XSUB = G::XSUB, XROW = G::XROW;
  constexpr int ESZ = BF ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Xs = lds;
  unsigned char* Ds = lds + G::XBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < WNT / 64);

  // the nnb x ncb workgroups of one (segment, split) read the SAME X and dY rows: consecutive virtual ids on ONE XCD (its L2), as in
  // sr3d_hwgrad.hip (6.6 GB fetched per launch for 3.1 GB of operands with the plain order, profiles/r04_pmc_hbm_traffic.json)
  int v;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nb = v % p.nnb;
  v /= p.nnb;
  const int cb = v % p.ncb;
  v /= p.ncb;
  const int seg = v % p.nseg;
  const int split = v / p.nseg;
  const int x0 = seg * 32;                      // coarse
  const long long IYX = (long long)p.IY * p.IX, IZYX = IYX * p.IZ;
  const long long OYX = (long long)p.OY * p.OX, OZYX = OYX * p.OZ;

  // ---- staging role (fixed): waves 0..5 one X item (plane dz, new row r, channel c, piece q of 16 fine voxels);
  // waves 6..9 one dY item (row n, piece q of 8 coarse voxels)
  const bool is_x = wave < 6;
  int it_c = 0, it_q = 0, it_dz = 0, it_r = 0, it_n = 0;
  const unsigned char* src = nullptr;
  long long src_b = 0;
  bool it_on = false;
  float mx = 1.f, md = 1.f;
  if (is_x) {
    it_dz = tid / 128, it_r = (tid % 128) / 64, it_c = (tid % 64) / 4, it_q = tid & 3;
    const int c = cb * CB + it_c;
    if (c < p.cu) {
      const int si = cat_find(p.x, c);
      src = reinterpret_cast<const unsigned char*>(cat_ptr(p.x, si)) + (long long)(c - cat_cbeg(p.x, si)) * IZYX * ESZ;
      src_b = cat_bstride(p.x, si);
      if constexpr (!BF) mx = ldexpf(1.f, scale_exp_of(p.amax[si]));
      it_on = 2 * x0 + 16 * it_q < p.IX;
    }
  } else {
    const int i = tid - 384;
    if (i < G::NDY) {
      it_n = i / 4, it_q = i & 3;
      const int n = nb * (32 * RT) + it_n;
      if (n < p.N) {
        const int si = cat_find(p.dy, n);
        src = reinterpret_cast<const unsigned char*>(cat_ptr(p.dy, si)) + (long long)(n - cat_cbeg(p.dy, si)) * OZYX * ESZ;
        src_b = cat_bstride(p.dy, si);
        if constexpr (!BF) md = ldexpf(1.f, scale_exp_of(p.amax[4 + si]));
        it_on = x0 + 8 * it_q < p.OX;
      }
    }
  }
  const bool stager = is_x || (tid - 384) < G::NDY;
  const int xf = 2 * x0 + 16 * it_q;            // first fine voxel of an X item
  const int xq = x0 + 8 * it_q;                 // first coarse voxel of a dY item
  float pv[16];                                 // fp32: X item 16 fine voxels / dY item elements 0..8
  u32x4 pa = {0u, 0u, 0u, 0u}, pb = {0u, 0u, 0u, 0u};   // bf16: X item 16 fine voxels (pa, pb) / dY item 8 (pa) + next (pb[0])
#pragma unroll
  for (int j = 0; j < 16; j++) pv[j] = 0.f;

  // X item: row (b, z, y) of the FINE grid; dY item: row (b, oz, oy) of the coarse grid (y < 0: nothing)
  auto load_piece = [&](const int b, const int z, const int y) {
    if (is_x) {
      const bool ok = it_on && (unsigned)z < (unsigned)p.IZ && (unsigned)y < (unsigned)p.IY;
      if constexpr (BF) {
        pa = pb = u32x4{0u, 0u, 0u, 0u};
        if (ok) {
          const unsigned short* r = reinterpret_cast<const unsigned short*>(src) + (long long)b * src_b + (long long)z * IYX + (long long)y * p.IX + xf;
          pa = *reinterpret_cast<const u32x4*>(r);
          if (xf + 8 < p.IX) pb = *reinterpret_cast<const u32x4*>(r + 8);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; j++) pv[j] = 0.f;
        if (ok) {
          const float* r = reinterpret_cast<const float*>(src) + (long long)b * src_b + (long long)z * IYX + (long long)y * p.IX + xf;
#pragma unroll
          for (int g4 = 0; g4 < 4; g4++)
            if (xf + 4 * g4 < p.IX) {
              const f32x4 a = *reinterpret_cast<const f32x4*>(r + 4 * g4);
              pv[4 * g4] = a.x, pv[4 * g4 + 1] = a.y, pv[4 * g4 + 2] = a.z, pv[4 * g4 + 3] = a.w;
            }
        }
      }
    } else {
      const bool ok = it_on && y >= 0;
      if constexpr (BF) {
        pa = pb = u32x4{0u, 0u, 0u, 0u};
        if (ok) {
          const unsigned short* r = reinterpret_cast<const unsigned short*>(src) + (long long)b * src_b + (long long)z * OYX + (long long)y * p.OX + xq;
          pa = *reinterpret_cast<const u32x4*>(r);
          pb[0] = xq + 8 < p.OX ? (unsigned)r[8] : 0u;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; j++) pv[j] = 0.f;
        if (ok) {
          const float* r = reinterpret_cast<const float*>(src) + (long long)b * src_b + (long long)z * OYX + (long long)y * p.OX + xq;
          const f32x4 a = *reinterpret_cast<const f32x4*>(r), c4 = *reinterpret_cast<const f32x4*>(r + 4);
          pv[0] = a.x, pv[1] = a.y, pv[2] = a.z, pv[3] = a.w, pv[4] = c4.x, pv[5] = c4.y, pv[6] = c4.z, pv[7] = c4.w;
          pv[8] = xq + 8 < p.OX ? r[8] : 0.f;
        }
      }
    }
  };
  // 8 values pv[off], pv[off + st], ... scaled and split
  auto split8 = [&](const int off, const int st, const float mult, h8& hi, h8& lo) {
    split_piece(&pv[off], st, mult, hi, lo);
  };
  // X item -> row slot `xslot` of its plane, both parities; dY item -> buffer dbuf, both copies
  auto write_piece = [&](const int xslot, const int dbuf, const float dsign) {
    if (!stager) return;
    if (is_x) {
      unsigned char* d = Xs + (it_dz * 5 + xslot) * XROW + it_c * PITCH + it_q * 16;
      if constexpr (BF) {
        // even elements = low halves, odd = high halves of the 8 dwords
        u32x4 ev, od;
#pragma unroll
        for (int k = 0; k < 2; k++) {
          ev[k] = __builtin_amdgcn_perm(pa[2 * k + 1], pa[2 * k], 0x05040100u);
          od[k] = __builtin_amdgcn_perm(pa[2 * k + 1], pa[2 * k], 0x07060302u);
          ev[2 + k] = __builtin_amdgcn_perm(pb[2 * k + 1], pb[2 * k], 0x05040100u);
          od[2 + k] = __builtin_amdgcn_perm(pb[2 * k + 1], pb[2 * k], 0x07060302u);
        }
        *reinterpret_cast<u32x4*>(d) = ev;
        *reinterpret_cast<u32x4*>(d + XSUB) = od;
      } else {
#pragma unroll
        for (int par = 0; par < 2; par++) {
          h8 hi, lo;
          split8(par, 2, mx, hi, lo);
          *reinterpret_cast<h8*>(d + par * XSUB) = hi;
          *reinterpret_cast<h8*>(d + 2 * XSUB + par * XSUB) = lo;
        }
      }
    } else {
      unsigned char* d = Ds + dbuf * G::DROW + it_n * PITCH + it_q * 16;
      if constexpr (BF) {
        const unsigned sm = dsign < 0.f ? 0x80008000u : 0u;
        const unsigned d0 = pa[0] ^ sm, d1 = pa[1] ^ sm, d2 = pa[2] ^ sm, d3 = pa[3] ^ sm, nh = pb[0] ^ (sm & 0xffffu);
        const u32x4 c0 = {__builtin_amdgcn_alignbit(d1, d0, 16), __builtin_amdgcn_alignbit(d2, d1, 16),
                          __builtin_amdgcn_alignbit(d3, d2, 16), __builtin_amdgcn_alignbit(nh, d3, 16)};   // dY[q + 1]
        *reinterpret_cast<u32x4*>(d) = c0;
        *reinterpret_cast<u32x4*>(d + G::DCOPY) = u32x4{d0, d1, d2, d3};
      } else {
#pragma unroll
        for (int cp = 0; cp < 2; cp++) {   // copy 0 = dY[q + 1], copy 1 = dY[q]
          h8 hi, lo;
          split8(1 - cp, 1, md * dsign, hi, lo);
          *reinterpret_cast<h8*>(d + cp * G::DCOPY) = hi;
          *reinterpret_cast<h8*>(d + cp * G::DCOPY + 32 * RT * PITCH) = lo;
        }
      }
    }
  };

  auto run = [&](auto few_tag) {
    constexpr bool FEW = decltype(few_tag)::value;
    constexpr int NT = 2 * RT;                  // 16-row tiles of the n block; ONE 16-channel tile
    constexpr int NK = FEW ? 3 : 2;
    f32x4 acc[NK][NT];
#pragma unroll
    for (int k = 0; k < NK; k++)
#pragma unroll
      for (int i = 0; i < NT; i++) acc[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float acc_sign = 1.f;
    // waves 0..8: (kz, ky) group `wave`, slots kx = 0, 1;  waves 9..11: kx = 2 of the groups 3 (w - 9) .. + 2
    const int g0 = FEW ? 3 * (wave - 9) : wave;
    const int fr = (lane & 15) * PITCH + (lane >> 4) * 16;
    // X row of tap group g at step t: plane g / 3, fine row 2 t - 1 + g % 3 -> slot (row + 1) mod 5 = (2 t + g % 3) mod 5
    auto xrow = [&](const int g, const int t) { return Xs + ((g / 3) * 5 + (2 * t + (g % 3) + 10) % 5) * XROW + fr; };

    const long long rows_total = (long long)p.B * p.OZ * p.OY;
    long long r0 = (long long)split * p.rows_per_split, r1 = r0 + p.rows_per_split;
    if (r1 > rows_total) r1 = rows_total;
    while (r0 < r1) {
      const long long plane = r0 / p.OY;                 // (b, oz)
      const int b = (int)(plane / p.OZ), oz = (int)(plane - (long long)b * p.OZ);
      const int ya = (int)(r0 - plane * p.OY);
      const long long pend = (plane + 1) * p.OY;
      const int yb = (int)((r1 < pend ? r1 : pend) - plane * p.OY);
      // step t: write what was loaded in step t - 1 (X rows 2t+2+r, dY row t+1), load (X rows 2t+4+r, dY row t+2),
      // multiply output row t (X rows 2t-1 .. 2t+1)
      for (int t = ya - 3; t < yb; t++) {
        if (t > ya - 3) {
          const long long rr = plane * p.OY + (t + 1);
          write_piece((2 * t + 3 + it_r + 10) % 5, (t + 1) & 1, ((rr >> 5) & 1) ? -1.f : 1.f);
        }
        if (is_x)
          load_piece(b, 2 * oz - 1 + it_dz, 2 * t + 4 + it_r);
        else
          load_piece(b, oz, (t + 2 >= ya && t + 2 < yb) ? t + 2 : -1);
        if (t >= ya) {
          const long long rr = plane * p.OY + t;
          const float sgn = ((rr >> 5) & 1) ? -1.f : 1.f;
          if (sgn != acc_sign) {
#pragma unroll
            for (int k = 0; k < NK; k++)
#pragma unroll
              for (int i = 0; i < NT; i++) acc[k][i] = -acc[k][i];
            acc_sign = sgn;
          }
          const unsigned char* db = Ds + (t & 1) * G::DROW + fr;
#pragma unroll
          for (int k = 0; k < NK; k++) {
            const int kx = FEW ? 2 : k;
            // kx = 1: even sub-row, dY; kx = 0: odd sub-row, dY shifted by one; kx = 2: odd sub-row, dY
            const unsigned char* xb = xrow(FEW ? g0 + k : g0, t) + (kx == 1 ? 0 : XSUB);
            const unsigned char* da = db + (kx == 0 ? 0 : G::DCOPY);
            const h8 bh = *reinterpret_cast<const h8*>(xb);
            h8 bl = bh;
            if constexpr (!BF) bl = *reinterpret_cast<const h8*>(xb + 2 * XSUB);
#pragma unroll
            for (int i = 0; i < NT; i++) {
              const h8 ah = *reinterpret_cast<const h8*>(da + i * 16 * PITCH);
              if constexpr (BF) {
                acc[k][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, ah), __builtin_bit_cast(bf8, bh), acc[k][i], 0, 0, 0);
              } else {
                const h8 al = *reinterpret_cast<const h8*>(da + 32 * RT * PITCH + i * 16 * PITCH);
                acc[k][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[k][i], 0, 0, 0);
                acc[k][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[k][i], 0, 0, 0);
                acc[k][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[k][i], 0, 0, 0);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      r0 = plane * p.OY + yb;
    }

    // ---- partial block -> slab[split, segment][tap][n][c]; 16x16 tile: column = lane & 15, row = 4 (lane >> 4) + register
#pragma unroll
    for (int k = 0; k < NK; k++) {
      const int tap = FEW ? (g0 + k) * 3 + 2 : g0 * 3 + k;
      float* out = p.slab + ((long long)(split * p.nseg + seg) * 27 + tap) * p.Npad * p.Cpad;
#pragma unroll
      for (int i = 0; i < NT; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int n = nb * (32 * RT) + i * 16 + 4 * (lane >> 4) + r;
          const int c = cb * CB + (lane & 15);
          out[(long long)n * p.Cpad + c] = acc[k][i][r] * acc_sign;
        }
    }
  };
  if (wave >= 9)
    run(std::true_type{});
  else
    run(std::false_type{});
}

struct Hw2SliceMap {
  int xcb[SR3D_MAX_SRC], dcb[SR3D_MAX_SRC];
};

// dW[n][c][tap] = 2^-(sx(c)+sd(n)) * sum_s slab[s][tap][n][c]  (fixed order: deterministic)
__global__ __launch_bounds__(256) void hwgrad_s2_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int N,
                                                               int cu, int ldc, int Npad, int Cpad, const float* amax,
                                                               const Hw2SliceMap sm) {
  const long long plane = (long long)Npad * Cpad;
  const long long total = (long long)N * cu * 27;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % cu);
    const long long r = e / cu;
    const int n = (int)(r % N), tap = (int)(r / N);
    const int xi = (c >= sm.xcb[1]) + (c >= sm.xcb[2]) + (c >= sm.xcb[3]), di = (n >= sm.dcb[1]) + (n >= sm.dcb[2]) + (n >= sm.dcb[3]);
    const float mult = amax ? ldexpf(1.f, -(scale_exp_of(amax[xi]) + scale_exp_of(amax[4 + di]))) : 1.f;
    const float* s0 = slab + (long long)tap * plane + (long long)n * Cpad + c;
    float s = 0.f;
    for (int k = 0; k < S; k++) s += s0[(long long)k * 27 * plane];
    dw[((long long)n * ldc + c) * 27 + tap] = s * mult;
  }
}

struct Hw2Plan {
  int rt, nnb, ncb, nseg, S, Npad, Cpad, OZ, OY, OX;
  long long rows_per_split;
};

Hw2Plan hw2_plan(const sr3d_conv_desc_t* d, int n_total) {
  Hw2Plan g;
  g.OZ = (d->Z - 1) / 2 + 1, g.OY = (d->Y - 1) / 2 + 1, g.OX = (d->X - 1) / 2 + 1;
  g.rt = n_total > 32 ? 2 : 1;
  g.nnb = ceil_div(n_total, 32 * g.rt), g.ncb = ceil_div(d->Cin, CB), g.nseg = ceil_div(g.OX, 32);
  g.Npad = g.nnb * 32 * g.rt, g.Cpad = g.ncb * CB;
  const long long rows = (long long)d->B * g.OZ * g.OY;
  const long long cols = (long long)g.nnb * g.ncb * g.nseg;
  // a whole number of rounds over the 256 CUs (one workgroup per CU), at least 16 rows per split (3 warm-up steps)
  long long S = 1;
  double best = -1.0;
  const long long smax = rows / 16 > 0 ? rows / 16 : 1;
  const long long s_hi = (1536 + cols - 1) / cols < 96 ? (1536 + cols - 1) / cols : 96;
  const long long s_lo = (512 + cols - 1) / cols < s_hi ? (512 + cols - 1) / cols : s_hi;
  for (long long s = s_lo; s <= s_hi; s++) {
    const long long sc = s < 1 ? 1 : (s > smax ? smax : s);
    const double wgs = (double)cols * sc, fill = wgs / (256.0 * (double)((long long)(wgs + 255) / 256));
    if (fill > best + 1e-3) best = fill, S = sc;
  }
  g.rows_per_split = (rows + S - 1) / S;
  g.S = (int)((rows + g.rows_per_split - 1) / g.rows_per_split);
  return g;
}

template <bool BF>
int hw2_launch(int rt, long long nwg, const Hw2Params& p, hipStream_t st) {
  constexpr size_t l2 = SGeo<2, BF>::LDS, l1 = SGeo<1, BF>::LDS;
  static SrPerDevice setup;   // (the attribute is per device, not per thread)
  if (int rc = setup.once([&]() -> int {
        SR3D_HIP(hipFuncSetAttribute((const void*)hwgrad_s2_kernel<2, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2));
        SR3D_HIP(hipFuncSetAttribute((const void*)hwgrad_s2_kernel<1, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1));
        return SR3D_OK;
      }))
    return rc;
  if (rt == 2)
    hipLaunchKernelGGL((hwgrad_s2_kernel<2, BF>), dim3((unsigned)nwg), dim3(WNT), l2, st, p);
  else
    hipLaunchKernelGGL((hwgrad_s2_kernel<1, BF>), dim3((unsigned)nwg), dim3(WNT), l1, st, p);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

}  // namespace

size_t sr3d_hwgrad_s2_ws_bytes(const sr3d_conv_desc_t* d, int n_total) {
  const Hw2Plan g = hw2_plan(d, n_total);
  return 256 + (size_t)g.S * g.nseg * 27 * g.Npad * g.Cpad * 4;
}

// 16-byte pieces: the coarse row length must be a multiple of 8, the fine one exactly twice it, tensors 16-byte aligned
bool sr3d_hwgrad_s2_ok(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy) {
  if (d->stride != 2 || d->X % 16 != 0) return false;
  for (int i = 0; i < x.n; i++)
    if (reinterpret_cast<uintptr_t>(x.ptr[i]) & 15) return false;
  for (int i = 0; i < dy.n; i++)
    if (reinterpret_cast<uintptr_t>(dy.ptr[i]) & 15) return false;
  return true;
}

int sr3d_hwgrad_s2(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy, int n_total, float* dw, float* ws, hipStream_t st,
                   const unsigned* x_absmax, const unsigned* dy_absmax) {
  const Hw2Plan g = hw2_plan(d, n_total);
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
    if (x_absmax != nullptr || dy_absmax != nullptr)
      if (int rc = sr3d_gather_absmax(x_absmax, x.n, dy_absmax, dy.n, amax, st)) return rc;
    SR3D_HIP(hipGetLastError());
  }
  Hw2Params p{};
  p.x = x, p.dy = dy, p.cu = d->Cin, p.N = n_total;
  p.B = d->B, p.IZ = d->Z, p.IY = d->Y, p.IX = d->X, p.OZ = g.OZ, p.OY = g.OY, p.OX = g.OX;
  p.nnb = g.nnb, p.ncb = g.ncb, p.nseg = g.nseg, p.S = g.S, p.rows_per_split = g.rows_per_split;
  p.Npad = g.Npad, p.Cpad = g.Cpad;
  p.slab = ws + 64, p.amax = (const float*)amax;
  const long long nwg = (long long)g.nnb * g.ncb * g.nseg * g.S;
  SR3D_CHECK(nwg < (1ll << 31), SR3D_E_ARG, "stride-2 weight gradient: grid too large");
  {
    SrProfScope prof(SR3D_PROF_WGRAD, 2.0 * 27 * d->Cin * (double)n_total * (double)g.OZ * g.OY * g.OX * d->B, st);
    if (int rc = bf ? hw2_launch<true>(g.rt, nwg, p, st) : hw2_launch<false>(g.rt, nwg, p, st)) return rc;
  }
  SrProfScope prof(SR3D_PROF_PACK, 4.0 * ((double)g.S * g.nseg + 1) * 27 * g.Npad * g.Cpad, st);
  const long long total = (long long)n_total * d->Cin * 27;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  Hw2SliceMap sm;
  for (int i = 0; i < SR3D_MAX_SRC; i++) sm.xcb[i] = x.cbeg[i], sm.dcb[i] = dy.cbeg[i];
  hipLaunchKernelGGL(hwgrad_s2_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)p.slab, dw, g.S * g.nseg, n_total, d->Cin,
                     d->Cin, g.Npad, g.Cpad, bf ? (const float*)nullptr : (const float*)amax, sm);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}
