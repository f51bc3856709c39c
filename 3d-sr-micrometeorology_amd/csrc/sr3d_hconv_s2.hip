// Stride-2 3x3x3 convolution on the split-f16 scheme of sr3d_hconv.hip (fp32 operands as two fp16 halves, three
// v_mfma_f32_32x32x16_f16 per product group, fp32 accumulate, block-floating-point scaling, sign-alternating
// accumulation): the forward conv of the DownBlocks and its input gradient.
//
// A stride-2 conv reads every input voxel for 27/8 outputs on average, so a direct halo tile is 8x larger per output
// than at stride 1 (5 x 9 x 65 voxels for 2 x 4 x 32 outputs: 187 KB as split fp16).  Both directions are therefore
// written as stride-1 convolutions over PARITY CLASSES with 1, 2, 4 or 8 taps (27 in total):
//   forward   y[o] = sum_k w[k] x[2o + k - 1]:  x splits into 8 sub-volumes x_p[q] = x[2q + p]; per dimension an even
//             class contributes the tap k = 1 at q = o, an odd class the taps k = 0 at q = o - 1 and k = 2 at q = o.
//             MODE 1: the chunk loop runs over (class, 16 channels); a chunk has (1+pz)(1+py) phases of (1+px) taps.
//   gradient  dx[2i + q] = sum over the taps that reach parity q: even: k = 1 from dy[i]; odd: k = 0 from dy[i + 1]
//             and k = 2 from dy[i].  MODE 2: one launch, blockIdx.z = output class; chunks run over 16 dy channels.
// The halo of a 2 x 4 x 32 tile is then 3 x 5 x 33 voxels per class, stored with exactly those pitches (32 KB as split fp16),
// 40 % of which the range-checked buffer loads skip without traffic.
//
// Per chunk only 12 .. 96 MFMAs per wave stand against the staging of 16 x 495 values, so the pipeline is simpler
// than at stride 1: the halo refill is not overlapped inside a workgroup; the second workgroup of the CU covers it.
#include "sr3d_split_f16.h"

#include <limits.h>
#include <stdlib.h>

#ifndef S2F_TPP_BF
#define S2F_TPP_BF 3
#endif
#ifndef S2F_WGS_BF
#define S2F_WGS_BF 3
#endif
#ifndef S2B_GPP_BF
#define S2B_GPP_BF 2
#endif
#ifndef S2B_GPP_F32
#define S2B_GPP_F32 2
#endif

namespace {


constexpr int HKC = 16;
constexpr int S2FLIP_SH = 2;                   // the packed weights and the accumulators change sign every 4 virtual chunks
// LDS pitches = the used region, 3 x 5 x 33 voxels per plane (the first version kept the 4 x 6 x 34 pitches of sr3d_hconv.hip:
// 56 KB of halo for 32 KB of data and two workgroups per CU; with tight planes a workgroup takes 48 KB and THREE share
// a CU -- this kernel lives on the overlap between workgroups, see the header -- and a chunk is staged in 4 rounds instead of 5)
constexpr int HHY = 5, HHX = 33;
constexpr int UZ = 3, UY = 5, UX = 33;
constexpr int HNR = 4;                         // staging rounds: voxel indices < 3 * 5 * 33 = 495 <= 8 * 64
constexpr int HVOX = 495, HVP = 512;           // voxels of a plane, padded (the wave maxima are exchanged in the padding of plane 0)
constexpr int HPLANE = HVP * 16;
constexpr int HBYTES = 4 * HPLANE;
constexpr int HNT = 256;
template <int RT, bool BF = false>
struct SGeo {
  static constexpr int NP = BF ? 1 : 2;            // operand parts: [hi | lo], or the bf16 value itself (sr3d_hconv.hip)
  static constexpr int WBUF = 2 * NP * RT * 1024;  // one phase: up to 2 taps x NP parts x RT fragments
  static constexpr int HB = NP * 2 * HPLANE;
  static constexpr size_t LDS = HB + 2 * (size_t)WBUF;
};
static_assert(3 * SGeo<2>::LDS <= 160 * 1024, "LDS budget: three workgroups per CU");

// halo coordinate of local tap i in a dimension of parity `par` (host and device)
//   MODE 1 (forward):  even: k = 1 at h = 1;  odd: i = 0 -> k = 0 at h = 0, i = 1 -> k = 2 at h = 1   (halo origin o0 - 1)
//   MODE 2 (gradient): even: k = 1 at h = 0;  odd: i = 0 -> k = 0 at h = 1, i = 1 -> k = 2 at h = 0   (halo origin i0)
__host__ __device__ inline int tap_h(int mode, int par, int i) { return mode == 1 ? (par ? i : 1) : (par ? 1 - i : 0); }
__host__ __device__ inline int tap_k(int par, int i) { return par ? 2 * i : 1; }

// accumulators -> destination: gated (forward) or plain (forward / the class `ocls` of the fine grid, input gradient)
template <int RT, int MODE, bool BF>
__device__ __forceinline__ void s2_epilogue(const SrHconvS2Params& p, f32x16 (&acc)[RT][2], const float out_mult, const int wave, const int lane,
                                            const int b, const int nblk, const int ocls, const int z0, const int y0, const int x0,
                                            const int TZ, const int TY, const int TX) {
  const int ox = x0 + (lane & 31);
  const int rblock = p.n_off + (p.nb_off + nblk) * 64;
  const long long TZYX = (long long)p.TZ_ * p.TY_ * p.TX_;
  if (ox >= TX) return;
  if (MODE == 1 && p.epi == SR3D_EPI_GATED) {
    if constexpr (RT == 2) {
      const int cbase = rblock / 2 + 4 * (lane >> 5);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int vt = 2 * wave + j;
        const int oz = z0 + (vt >> 2), oy = y0 + (vt & 3);
        if (oz >= TZ || oy >= TY) continue;
        const long long sp = ((long long)oz * p.TY_ + oy) * p.TX_ + ox;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int co = cbase + (r & 3) + 8 * (r >> 2);
          if (co < p.Cg) {
            float f = acc[0][j][r] * out_mult;
            if (p.bias) f += p.bias[co];
            const float g = acc[1][j][r] * out_mult + (p.bias2 ? p.bias2[co] : 0.f);
            const float s = 1.f / (1.f + expf(-g));
            f = split_act(f, p.act);
            const long long o = ((long long)b * p.Cg + co) * TZYX + sp;
            st_act<BF>(p.y, o, s * f);
            if (p.save_f) st_act<BF>(p.save_f, o, f);
            if (p.save_s) st_act<BF>(p.save_s, o, s);
          }
        }
      }
    }
  } else {
    const int so = MODE == 2 ? 2 : 1;
    const int qz = MODE == 2 ? (ocls >> 2) & 1 : 0, qy = MODE == 2 ? (ocls >> 1) & 1 : 0, qx = MODE == 2 ? ocls & 1 : 0;
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int n = rblock + i * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
        if (n >= p.N) continue;
        const int si = cat_find(p.out, n);
        float* base = cat_ptr(p.out, si);
        if (base == nullptr) continue;
        const long long boff = (long long)b * cat_bstride(p.out, si) + (long long)(n - cat_cbeg(p.out, si)) * TZYX;   // elements
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int vt = 2 * wave + j;
          const int oz = z0 + (vt >> 2), oy = y0 + (vt & 3);
          if (oz < TZ && oy < TY)
            st_act<BF>(base, boff + ((long long)(oz * so + qz) * p.TY_ + (oy * so + qy)) * p.TX_ + (ox * so + qx),
                       split_act(acc[i][j][r] * out_mult + bv, p.act));
        }
      }
  }
}

// QD (input gradient, fp32, IX % 4 == 0, 16-byte aligned tensors): the dY halo rows are contiguous in x and are fetched as
// 16-byte QUADS: wave w stages channels 4w .. 4w + 3 of the chunk, a lane takes quad q of halo row (hz, hy) -- 15 rows x 9
// quads = 135 tasks in 3 rounds -- 12 loads per lane and chunk instead of 40.  The kernel was bound by the NUMBER of
// vector-memory instructions (~22 cycles of the texture path per wave instruction: 320 per chunk pair of a CU against
// 2600 cycles of MFMAs); a lane then holds 4 voxels x 4 channels and writes 8-byte half pieces.
template <int RT, int MODE, bool BF, bool QD = false>
__global__ __launch_bounds__(HNT, (BF && MODE == 2 && RT == 2 && !QD) ? 2 : 3) void hconv_s2_kernel(const SrHconvS2Params p) {   // (that one form needs 219 VGPRs)
  static_assert(!QD || MODE == 2, "quad loads: the input gradient (fp32: 16-byte quads; bf16, round 4: 8-byte quads)");
  static_assert(HNR == 4, "the counted wait behind phase 0 assumes 8 * HNR = 32 raw-row loads");
  using G = SGeo<RT, BF>;
  constexpr int NP = G::NP;
  constexpr int ESZ = BF ? 2 : 4;              // bytes per activation element
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Hs = lds;
  unsigned char* Ws = lds + G::HB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < HNT / 64);

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
  const int ocls = MODE == 2 ? (int)blockIdx.z : 0;                 // output parity class of this launch slice
  const int TZ = MODE == 2 ? p.cZ[ocls] : p.Z, TY = MODE == 2 ? p.cY[ocls] : p.Y, TX = MODE == 2 ? p.cX[ocls] : p.X;
  const int z0 = tiz * 2, y0 = tiy * 4, x0 = tix * 32;
  if (z0 >= TZ || y0 >= TY || x0 >= TX) return;                     // (the grid is sized for the largest class)
  const long long IZYX = (long long)p.IZ * p.IY * p.IX;
  const int chan_bytes = (int)(IZYX * ESZ);

  int sw = BF ? 0 : split_scale_exp(*p.absmax_w);
  if (sw == kSplitScaleNone) sw = 0;
  float* xmax = reinterpret_cast<float*>(Hs + HVOX * 16);

  // ---- staging geometry (class-independent part): halo voxel of this lane in round r
  // Byte offset of this lane's halo voxel inside a channel volume, worked out ONCE: sbase = offset for K-side class
  // (0, 0, 0), smask bit c = the voxel of class c lies inside the grid.  A load is then one shift, one add, one select.
  // (The first version recomputed coordinates and range checks for every one of the 8 * HNR loads of every chunk: 12 vector
  // and 14 scalar instructions per MFMA, matrix pipe 18 % busy -- profiles/r03c_pmc_instruction_mix_down1_0.json.)
  const int sh = wave & 1;
  unsigned sbase[HNR], smask[HNR];
  int swr[HNR];
#pragma unroll
  for (int r = 0; r < HNR; r++) {
    const int e = (r * 2 + (wave >> 1)) * 64 + lane;
    const int hz = e / (HHY * HHX), r2 = e - hz * (HHY * HHX);
    const int hy = r2 / HHX, hx = r2 - hy * HHX;
    const bool in_halo = hz < UZ && hy < UY && hx < UX;
    swr[r] = e * 16;
    unsigned m = 0u;
    if (MODE == 1) {   // sub-volume index q = o0 - 1 + h, input coordinate 2q + class parity
      const int gz = 2 * (z0 - 1 + hz), gy = 2 * (y0 - 1 + hy), gx = 2 * (x0 - 1 + hx);
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const bool ok = in_halo && (unsigned)(gz + ((c >> 2) & 1)) < (unsigned)p.IZ && (unsigned)(gy + ((c >> 1) & 1)) < (unsigned)p.IY &&
                        (unsigned)(gx + (c & 1)) < (unsigned)p.IX;
        m |= ok ? (1u << c) : 0u;
      }
      sbase[r] = (unsigned)((gz * p.IY + gy) * p.IX + gx) * (unsigned)ESZ;   // (wraps for border voxels; used only where a bit is set)
    } else {           // dy coordinate i0 + h
      const int gz = z0 + hz, gy = y0 + hy, gx = x0 + hx;
      m = (in_halo && (unsigned)gz < (unsigned)p.IZ && (unsigned)gy < (unsigned)p.IY && (unsigned)gx < (unsigned)p.IX) ? 1u : 0u;
      sbase[r] = (unsigned)((gz * p.IY + gy) * p.IX + gx) * (unsigned)ESZ;
    }
    smask[r] = m;
  }
  constexpr int QNR = 3, QPR = 9;     // rounds; quads per halo row (hx = 4 q .. 4 q + 3, the last one holds column 32 only)
  unsigned qoff[QNR];                 // byte offset of the quad inside a channel volume (0xffffffff: zero padding)
  int qwr[QNR], qmask[QNR];           // LDS byte offset of the quad's first voxel; its voxels inside the halo
  if constexpr (QD) {
#pragma unroll
    for (int r = 0; r < QNR; r++) {
      const int t = r * 64 + lane;
      const int row = t / QPR, q = t - row * QPR;
      const int hz = row / UY, hy = row - hz * UY;
      const int gz = z0 + hz, gy = y0 + hy, gx = x0 + 4 * q;
      const bool task = t < UZ * UY * QPR;
      const bool ok = task && (unsigned)gz < (unsigned)p.IZ && (unsigned)gy < (unsigned)p.IY && (unsigned)gx < (unsigned)p.IX;   // IX % 4 == 0: all four or none
      qoff[r] = ok ? (unsigned)((gz * p.IY + gy) * p.IX + gx) * (unsigned)ESZ : 0xffffffffu;
      qwr[r] = ((hz * HHY + hy) * HHX + 4 * q) * 16;
      qmask[r] = !task ? 0 : (q == QPR - 1 ? 1 : 15);
    }
  }

  // per-slice base pointers in scalar registers, mask arithmetic (see sr3d_hconv.hip)
#define SR3D_SLICE_BASE(i) (reinterpret_cast<unsigned long long>(p.in.ptr[i]) + (unsigned long long)((long long)b * p.in.bstride[i]) * ESZ)
  unsigned long long sb0 = SR3D_SLICE_BASE(0), sb1 = SR3D_SLICE_BASE(1), sb2 = SR3D_SLICE_BASE(2), sb3 = SR3D_SLICE_BASE(3);
#undef SR3D_SLICE_BASE
  int cb0 = p.in.cbeg[0], cb1 = p.in.cbeg[1], cb2 = p.in.cbeg[2], cb3 = p.in.cbeg[3];
  split_pin_scalar(sb0), split_pin_scalar(sb1), split_pin_scalar(sb2), split_pin_scalar(sb3);
  split_pin_scalar(cb0), split_pin_scalar(cb1), split_pin_scalar(cb2), split_pin_scalar(cb3);
  unsigned long long dsb1 = sb1 - sb0, dsb2 = sb2 - sb1, dsb3 = sb3 - sb2;
  int dcb1 = cb1 - cb0, dcb2 = cb2 - cb1, dcb3 = cb3 - cb2;
  split_pin_scalar(dsb1), split_pin_scalar(dsb2), split_pin_scalar(dsb3), split_pin_scalar(dcb1), split_pin_scalar(dcb2), split_pin_scalar(dcb3);
  auto chan_base = [&](const int gc) {
    const long long m1 = -(long long)(gc >= cb1), m2 = -(long long)(gc >= cb2), m3 = -(long long)(gc >= cb3);
    const unsigned long long base = sb0 + (dsb1 & (unsigned long long)m1) + (dsb2 & (unsigned long long)m2) + (dsb3 & (unsigned long long)m3);
    const int c0 = cb0 + (dcb1 & (int)m1) + (dcb2 & (int)m2) + (dcb3 & (int)m3);
    return base + (unsigned long long)(unsigned)(gc - c0) * (unsigned long long)(unsigned)chan_bytes;
  };

  // ---- chunk iteration space.  MODE 1: virtual chunk vc = class * cpc + cc (class = K-side parity).  MODE 2: vc = cc,
  // the tap structure comes from the launch's output class.
  const int cpc = p.nchunks;
  const int NV = MODE == 1 ? 8 * cpc : cpc;
  // (class, chunk-in-class) of a virtual chunk are kept as running counters: no divisions in the loop

  float raw[HNR][8];
  float rawq[QNR][16];   // QD: [round][channel 0..3][voxel of the quad]
  auto load_raw = [&](const int vc, const int kc_, const int cc_) {   // kc_: K-side class (mode 1), cc_: 16-channel chunk
    if constexpr (QD) {
      const bool live = vc < NV;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int gc = cc_ * HKC + wave * 4 + c;   // wave-uniform
        const unsigned long long base = chan_base(gc < p.K ? gc : p.K - 1);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, live && gc < p.K ? chan_bytes : 0, 0x00020000);
#pragma unroll
        for (int r = 0; r < QNR; r++) {
          if constexpr (BF) {   // 4 bf16 x-neighbours: two dwords (voxels 0|1, 2|3)
            const auto t = __builtin_amdgcn_raw_buffer_load_b64(rs, qoff[r], 0, 0);
            rawq[r][c * 4 + 0] = __builtin_bit_cast(float, (unsigned)t[0]);
            rawq[r][c * 4 + 1] = __builtin_bit_cast(float, (unsigned)t[1]);
          } else {
            const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, qoff[r], 0, 0);
#pragma unroll
            for (int v = 0; v < 4; v++) rawq[r][c * 4 + v] = __builtin_bit_cast(float, (unsigned)t[v]);
          }
        }
      }
      return;
    }
    const int kc = MODE == 1 ? kc_ : 0;                            // (mode 2: dy is not subsampled)
    const int cc = cc_;
    const bool live = vc < NV;
    const int gc0 = cc * HKC + sh * 8;
    const unsigned cdelta = (unsigned)((((kc >> 2) & 1) * p.IY + ((kc >> 1) & 1)) * p.IX + (kc & 1)) * (unsigned)ESZ;   // wave-uniform
    unsigned so_r[HNR];
#pragma unroll
    for (int r = 0; r < HNR; r++) so_r[r] = ((smask[r] >> kc) & 1u) ? sbase[r] + cdelta : 0xffffffffu;
    // channel bases: the usual case is 8 consecutive channels of one tensor
    const int first = gc0 < p.K ? gc0 : p.K - 1, last = gc0 + 7 < p.K ? gc0 + 7 : p.K - 1;
    const bool one_slice = ((first >= cb1) + (first >= cb2) + (first >= cb3)) == ((last >= cb1) + (last >= cb2) + (last >= cb3));
    const unsigned long long b0 = chan_base(first);
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int gc = gc0 + c;
      const unsigned long long base = one_slice ? b0 + (unsigned long long)c * (unsigned)chan_bytes : chan_base(gc < p.K ? gc : p.K - 1);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, live && gc < p.K ? chan_bytes : 0, 0x00020000);
#pragma unroll
      for (int r = 0; r < HNR; r++) {
        const unsigned so = so_r[r];
        if constexpr (BF)   // the 16 bits of the bf16 element, zero-extended
          raw[r][c] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, so, 0, 0));
        else
          raw[r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, so, 0, 0));
      }
    }
  };
  // per-slice maxima of |x| over this workgroup's halo tiles (forward, first row block): exported at the end for the
  // split-f16 weight gradient of the same layer, as sr3d_hconv.hip does (the 8 parity classes together see every element)
  const bool export_max = MODE == 1 && !BF && !QD && p.amax_out != nullptr && nblk == 0;
  float rmax0 = 0.f, rmax1 = 0.f, rmax2 = 0.f, rmax3 = 0.f;
  auto publish_max = [&](const int parity, const int cc_) {
    if constexpr (BF) return;   // no scaling: bf16 has fp32's exponent range
    float m = 0.f;
    if constexpr (QD) {   // (the last quad of a row brings three columns from beyond the halo: the tile scale is only more cautious)
#pragma unroll
      for (int r = 0; r < QNR; r++)
#pragma unroll
        for (int c = 0; c < 16; c += 2) m = fmaxf(fmaxf(m, fabsf(rawq[r][c])), fabsf(rawq[r][c + 1]));
    } else {
#pragma unroll
      for (int r = 0; r < HNR; r++)
#pragma unroll
        for (int c = 0; c < 8; c += 2) m = fmaxf(fmaxf(m, fabsf(raw[r][c])), fabsf(raw[r][c + 1]));
    }
    m = split_wave_max(m);
    if (lane == 0) xmax[parity * 4 + wave] = m;
    if constexpr (MODE == 1 && !BF && !QD) {
      if (export_max) {
        const int gc0 = cc_ * HKC + sh * 8;
        auto slice_of = [&](const int gc) { return (gc >= cb1) + (gc >= cb2) + (gc >= cb3); };
        const int sa = slice_of(gc0 < p.K ? gc0 : p.K - 1), sb = slice_of(gc0 + 7 < p.K ? gc0 + 7 : p.K - 1);
        auto credit = [&](const int sl, const float mv) {
          rmax0 = sl == 0 ? fmaxf(rmax0, mv) : rmax0;
          rmax1 = sl == 1 ? fmaxf(rmax1, mv) : rmax1;
          rmax2 = sl == 2 ? fmaxf(rmax2, mv) : rmax2;
          rmax3 = sl == 3 ? fmaxf(rmax3, mv) : rmax3;
        };
        if (sa == sb) {
          credit(sa, m);
        } else {   // the 8 channels straddle a slice boundary: one maximum per channel
#pragma unroll
          for (int c = 0; c < 8; c++) {
            float mc = 0.f;
#pragma unroll
            for (int r = 0; r < HNR; r++) mc = fmaxf(mc, fabsf(raw[r][c]));
            mc = split_wave_max(mc);
            credit(slice_of(gc0 + c < p.K ? gc0 + c : p.K - 1), mc);
          }
        }
      }
    }
  };
  auto next_scale = [&](const int parity, const int s_run) {
    if constexpr (BF) return 0;
    const float m = fmaxf(fmaxf(xmax[parity * 4 + 0], xmax[parity * 4 + 1]), fmaxf(xmax[parity * 4 + 2], xmax[parity * 4 + 3]));
    const int s_c = __builtin_amdgcn_readfirstlane(split_scale_exp(m));
    return s_c < s_run ? s_c : s_run;
  };
  auto split_and_write = [&](const float in_mult) {
    if constexpr (QD && BF) {   // the same half pieces, packed instead of split: (ch 0 | ch 1), (ch 2 | ch 3) of voxel v
      unsigned char* H0 = Hs + (wave >> 1) * HPLANE + (wave & 1) * 8;
#pragma unroll
      for (int r = 0; r < QNR; r++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const unsigned sel = (v & 1) ? 0x07060302u : 0x05040100u;     // high / low halves of the two dwords
          const unsigned a0 = __builtin_bit_cast(unsigned, rawq[r][0 * 4 + (v >> 1)]), a1 = __builtin_bit_cast(unsigned, rawq[r][1 * 4 + (v >> 1)]);
          const unsigned a2 = __builtin_bit_cast(unsigned, rawq[r][2 * 4 + (v >> 1)]), a3 = __builtin_bit_cast(unsigned, rawq[r][3 * 4 + (v >> 1)]);
          if ((qmask[r] >> v) & 1) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2*>(H0 + qwr[r] + v * 16) = u32x2{__builtin_amdgcn_perm(a1, a0, sel), __builtin_amdgcn_perm(a3, a2, sel)};
          }
        }
      return;
    }
    if constexpr (QD) {   // channels 4 w .. 4 w + 3 = bytes 8 (w & 1) .. + 7 of the 16-byte piece of channel half w >> 1
      unsigned char* H0 = Hs + (wave >> 1) * HPLANE + (wave & 1) * 8;
#pragma unroll
      for (int r = 0; r < QNR; r++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
          unsigned h0, l0, h1, l1;
          split_pair(rawq[r][0 * 4 + v], rawq[r][1 * 4 + v], in_mult, h0, l0);
          split_pair(rawq[r][2 * 4 + v], rawq[r][3 * 4 + v], in_mult, h1, l1);
          if ((qmask[r] >> v) & 1) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2*>(H0 + qwr[r] + v * 16) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(H0 + 2 * HPLANE + qwr[r] + v * 16) = u32x2{l0, l1};
          }
        }
      return;
    }
#pragma unroll
    for (int r = 0; r < HNR; r++) {
      if constexpr (BF) {   // the 8 channels of a voxel, packed: the MFMA operand as it is
        u32x4 pk;
#pragma unroll
        for (int c = 0; c < 4; c++)
          pk[c] = __builtin_bit_cast(unsigned, raw[r][2 * c]) | (__builtin_bit_cast(unsigned, raw[r][2 * c + 1]) << 16);
        if (swr[r] < HVOX * 16) *reinterpret_cast<u32x4*>(Hs + sh * HPLANE + swr[r]) = pk;
        continue;
      }
      h8 hi, lo;
      split_piece(&raw[r][0], 1, in_mult, hi, lo);
      if (swr[r] < HVOX * 16) {   // (the padding behind the 495 voxels holds the exchange slots)
        *reinterpret_cast<h8*>(Hs + (0 * 2 + sh) * HPLANE + swr[r]) = hi;
        *reinterpret_cast<h8*>(Hs + (1 * 2 + sh) * HPLANE + swr[r]) = lo;
      }
    }
  };

  // ---- weights: the phases of this workgroup's row block are contiguous in execution order
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.wimg) + (MODE == 2 ? p.cls_off[ocls] : 0) +
                               (size_t)(p.nb_off + nblk) * (MODE == 2 ? p.cls_blk[ocls] : p.blk_stride);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (int)(MODE == 2 ? p.cls_blk[ocls] : p.blk_stride), 0x00020000);
  auto dma_w = [&](const int woff, const int npieces, unsigned char* W) {
#pragma unroll
    for (int ii = 0; ii < 2; ii++) {
      const int i = wave + 4 * ii;
      if (i < npieces) split_lds_dma16(wrs, (lds_p)(W + i * 1024), woff + i * 1024 + lane * 16);
    }
  };

  f32x16 acc[RT][2];
#pragma unroll
  for (int i = 0; i < RT; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  int bbase[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int vt = 2 * wave + j;
    bbase[j] = (lane >> 5) * HPLANE + (((vt >> 2) * HHY + (vt & 3)) * HHX + (lane & 31)) * 16;
  }
  const int abase = lane * 16;

  // ---- prologue: weights of the first phase, first chunk staged
  int cls = MODE == 1 ? 0 : ocls;
  int cc_cur = 0;                               // chunk inside the class (mode 1) / chunk (mode 2)
  int nxp = 1 + (cls & 1);                      // taps of a phase in this chunk's class
  int woff = 0, gph = 0;
  dma_w(0, nxp * NP * RT, Ws);
  load_raw(0, 0, 0);
  publish_max(0, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int s_run = next_scale(0, kSplitScaleNone);
  split_and_write(ldexpf(1.f, s_run));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int vc = 0; vc < NV; vc++) {
    const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
    const int ny = 1 + py, nph = (1 + pz) * ny;
    nxp = 1 + px;
    // the next virtual chunk
    int cc_n = cc_cur + 1, kcls_n = cls;
    if (MODE == 1 && cc_n == cpc) cc_n = 0, kcls_n = cls + 1;
    const int cls_n = vc + 1 < NV ? (MODE == 1 ? kcls_n : ocls) : cls;
    for (int ph = 0; ph < nph; ph++, gph++) {
      const unsigned char* W = Ws + (gph & 1) * G::WBUF + abase;
      const int wsize = nxp * NP * RT * 1024;
      const bool last = ph + 1 == nph;
      if (!(last && vc + 1 == NV)) dma_w(woff + wsize, (last ? 1 + (cls_n & 1) : nxp) * NP * RT, Ws + ((gph + 1) & 1) * G::WBUF);
      __builtin_amdgcn_sched_barrier(0);   // (the wait below counts on the DMA being older than the raw rows)
      if (ph == 0) load_raw(vc + 1, kcls_n, cc_n);
      const int iz = ph >> py, iy = ph & py;     // ph = iz * ny + iy with ny = 1 + py
      const unsigned char* Hk = Hs + ((tap_h(MODE, pz, iz) * HHY + tap_h(MODE, py, iy)) * HHX) * 16;
      // one x tap at a time (fragments of both taps in registers at once were 64 VGPRs: 194 in all, two waves per SIMD; with 32
      // the kernel fits three)
#pragma unroll
      for (int ix = 0; ix < 2; ix++) {
        if (ix < nxp) {
          h8 fa[NP][RT], fb[NP][2];   // [part][row tile], [part][voxel row]
          const int hx = tap_h(MODE, px, ix);
#pragma unroll
          for (int part = 0; part < NP; part++) {
#pragma unroll
            for (int i = 0; i < RT; i++) fa[part][i] = *reinterpret_cast<const h8*>(W + ((ix * NP + part) * RT + i) * 1024);
#pragma unroll
            for (int j = 0; j < 2; j++) fb[part][j] = *reinterpret_cast<const h8*>(Hk + part * (2 * HPLANE) + bbase[j] + hx * 16);
          }
#pragma unroll
          for (int i = 0; i < RT; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
              if constexpr (BF) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[0][i]), __builtin_bit_cast(bf8, fb[0][j]),
                                                                   acc[i][j], 0, 0, 0);
              } else {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[NP - 1][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[NP - 1][i], fb[0][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
              }
            }
        }
      }
      // The chunk's LAST phase also publishes the maxima of the next chunk's rows (landed: they are waited for here), so
      // that its barrier serves the exchange too -- a chunk has 1 .. 4 phases of 12 .. 24 MFMAs per wave, and a separate
      // publish-and-barrier round after them was a third of the synchronisation of the kernel.
      if (last && vc + 1 < NV) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish_max((vc + 1) & 1, cc_n);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else if (ph == 0) {   // next phase's weights landed (older than the 8 * HNR (QD: 4 * 3) raw-row loads issued in phase 0)
        if constexpr (QD)
          asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      woff += wsize;
    }
    if (vc + 1 < NV) {
      const int s_next = next_scale((vc + 1) & 1, s_run);
      // sign alternation + rescale, see sr3d_hconv.hip: the sign turns every 2^S2FLIP_SH chunks (a chunk here is short: 12 .. 96
      // MFMAs per wave against 32 packed multiplies for the 64 accumulator registers), and the multiply is skipped when
      // there is neither a turn nor a new scale
      const bool turn = (((vc + 1) >> S2FLIP_SH) ^ (vc >> S2FLIP_SH)) & 1;
      if (!BF && (turn || s_next != s_run)) {
        const float flip = ldexpf(turn ? -1.f : 1.f, s_next - s_run);
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
          for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] *= flip;
      }
      s_run = s_next;
      split_and_write(ldexpf(1.f, s_run));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    cc_cur = cc_n;
    if (MODE == 1) cls = kcls_n;
  }

  if (export_max && lane == 0) {   // (bits of a non-negative float order like unsigned integers)
    unsigned* slot = p.amax_out + ((v / p.nblk) & 63);
    if (rmax0 > 0.f) atomicMax(slot, __float_as_uint(rmax0));
    if (rmax1 > 0.f) atomicMax(slot + 64, __float_as_uint(rmax1));
    if (rmax2 > 0.f) atomicMax(slot + 128, __float_as_uint(rmax2));
    if (rmax3 > 0.f) atomicMax(slot + 192, __float_as_uint(rmax3));
  }
  // ------------------------------------------------------------------ epilogue
  const float out_mult = ldexpf((!BF && (((NV - 1) >> S2FLIP_SH) & 1)) ? -1.f : 1.f, -((s_run == kSplitScaleNone ? 0 : s_run) + sw));
  s2_epilogue<RT, MODE, BF>(p, acc, out_mult, wave, lane, b, nblk, ocls, z0, y0, x0, TZ, TY, TX);
}

// ---- FORWARD WITH BOTH X PARITIES PER LOAD (round 4; X % 4 == 0).  The class form above fetches every second element of a
// row (4-byte loads at an 8-byte pitch: 32 wave instructions per 16-channel chunk for 12 .. 96 MFMAs, each cache line visited
// by the even-x and by the odd-x class, cpc chunks apart) and was bound by the NUMBER of vector-memory instructions like the
// input gradient before its quad loads -- in bf16 storage, with a third of the MFMAs, at 234 TFLOP/s against ~1000 at stride
// 1.  Here a virtual chunk is ((pz, py), 16 channels) and holds BOTH x classes: the 3 x 5 halo rows of the pair are fetched
// as 16-byte quads x = 2 x0 - 4 + 4 q .. + 3, q = 0 .. 16 (255 tasks = 2 rounds of 2 waves; wave w stages channel half w & 1
// as in sr3d_hconv.hip: 16 loads per lane for what took 64), voxels 0 / 2 of a quad go to the even plane and 1 / 3 to the odd
// plane (halo columns 2 q - 1 and 2 q), and the three x taps of a (kz, ky) -- k = 1 from the even plane, k = 0 and 2 from
// the odd one -- follow each other as a flat tap list in phases of two taps (14 phases per 16 channels instead of 18).
// LDS: two planes sets, 64 + 16 KB (split form, two workgroups per CU) or 32 + 8 KB (bf16, three).
constexpr int FNR = 2, FQX = 17;
template <int RT, bool BF>
struct FGeo {
  static constexpr int NP = BF ? 1 : 2;
  // taps per phase (= per barrier): the split form has 12 MFMAs per tap and wave and LDS for two taps per buffer; bf16 has 4
  // MFMAs per tap -- two taps were a barrier per 8 MFMAs --: three, the x taps of one (kz, ky) (down1.0 forward 2.35 ms with
  // two, 2.20 with four, 2.15 with three and three workgroups per CU: profiles/r04ad_s2_fwd_variants.log)
  static constexpr int TPP = BF ? S2F_TPP_BF : 2;
  static constexpr int WBUF = TPP * NP * RT * 1024;
  static constexpr int PXB = NP * 2 * HPLANE;       // one x class: [part][channel half] planes
  static constexpr int HB = 2 * PXB;
  static constexpr size_t LDS = HB + 2 * (size_t)WBUF;
};
static_assert(2 * FGeo<2, false>::LDS <= 160 * 1024 && 3 * FGeo<2, true>::LDS <= 160 * 1024, "LDS budget of the paired forward");
__host__ __device__ inline int pair_taps(int pp) { return 3 * (1 + (pp >> 1)) * (1 + (pp & 1)); }   // pp = 2 pz + py
__host__ __device__ inline int pair_taps_before(int pp) { return pp == 0 ? 0 : pp == 1 ? 3 : pp == 2 ? 9 : 15; }

template <int RT, bool BF>
__global__ __launch_bounds__(HNT, BF ? S2F_WGS_BF : 2) void hconv_s2_fwd_kernel(const SrHconvS2Params p) {
  using G = FGeo<RT, BF>;
  constexpr int NP = G::NP;
  constexpr int ESZ = BF ? 2 : 4;
  constexpr int NRAW = 8 * FNR;                // raw-row loads per lane and chunk
  constexpr int TPP = G::TPP;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Hs = lds;
  unsigned char* Ws = lds + G::HB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < HNT / 64);
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
  const int TZ = p.Z, TY = p.Y, TX = p.X;
  const int z0 = tiz * 2, y0 = tiy * 4, x0 = tix * 32;
  if (z0 >= TZ || y0 >= TY || x0 >= TX) return;
  const long long IZYX = (long long)p.IZ * p.IY * p.IX;
  const int chan_bytes = (int)(IZYX * ESZ);

  int sw = BF ? 0 : split_scale_exp(*p.absmax_w);
  if (sw == kSplitScaleNone) sw = 0;
  float* xmax = reinterpret_cast<float*>(Hs + HVOX * 16);

  // ---- staging geometry: quad q of halo row (hz, hy); fbase = byte offset for (pz, py) = (0, 0), bit pp of fmask = the
  // row of pair pp lies inside the grid (IX % 4 == 0: a quad is inside or outside as a whole)
  const int sh = wave & 1;
  unsigned fbase[FNR], fmask[FNR];
  int fwr[FNR], fvox[FNR];
#pragma unroll
  for (int r = 0; r < FNR; r++) {
    const int t = (r * 2 + (wave >> 1)) * 64 + lane;
    const int row = t / FQX, q = t - row * FQX;
    const int hz = row / UY, hy = row - hz * UY;
    const bool task = t < UZ * UY * FQX;
    const int gz = 2 * (z0 - 1 + hz), gy = 2 * (y0 - 1 + hy), gx = 2 * x0 - 4 + 4 * q;
    const bool okx = task && (unsigned)gx < (unsigned)p.IX;
    unsigned m = 0u;
#pragma unroll
    for (int pp = 0; pp < 4; pp++)
      m |= (okx && (unsigned)(gz + (pp >> 1)) < (unsigned)p.IZ && (unsigned)(gy + (pp & 1)) < (unsigned)p.IY) ? (1u << pp) : 0u;
    fmask[r] = m;
    fbase[r] = (unsigned)((gz * p.IY + gy) * p.IX + gx) * (unsigned)ESZ;   // (wraps at the borders; used only where a bit is set)
    fwr[r] = ((hz * HHY + hy) * HHX + 2 * q - 1) * 16;                       // halo column 2 q - 1 (voxels 0, 1); voxels 2, 3: + 16
    fvox[r] = !task ? 0 : (q == 0 ? 8 : 15);                                 // q = 0: only x = 2 x0 - 1 (odd plane, column 0)
  }

#define SR3D_SLICE_BASE(i) (reinterpret_cast<unsigned long long>(p.in.ptr[i]) + (unsigned long long)((long long)b * p.in.bstride[i]) * ESZ)
  unsigned long long sb0 = SR3D_SLICE_BASE(0), sb1 = SR3D_SLICE_BASE(1), sb2 = SR3D_SLICE_BASE(2), sb3 = SR3D_SLICE_BASE(3);
#undef SR3D_SLICE_BASE
  int cb0 = p.in.cbeg[0], cb1 = p.in.cbeg[1], cb2 = p.in.cbeg[2], cb3 = p.in.cbeg[3];
  split_pin_scalar(sb0), split_pin_scalar(sb1), split_pin_scalar(sb2), split_pin_scalar(sb3);
  split_pin_scalar(cb0), split_pin_scalar(cb1), split_pin_scalar(cb2), split_pin_scalar(cb3);
  unsigned long long dsb1 = sb1 - sb0, dsb2 = sb2 - sb1, dsb3 = sb3 - sb2;
  int dcb1 = cb1 - cb0, dcb2 = cb2 - cb1, dcb3 = cb3 - cb2;
  split_pin_scalar(dsb1), split_pin_scalar(dsb2), split_pin_scalar(dsb3), split_pin_scalar(dcb1), split_pin_scalar(dcb2), split_pin_scalar(dcb3);
  auto slice_of = [&](const int gc) { return (gc >= cb1) + (gc >= cb2) + (gc >= cb3); };
  auto chan_base = [&](const int gc) {
    const long long m1 = -(long long)(gc >= cb1), m2 = -(long long)(gc >= cb2), m3 = -(long long)(gc >= cb3);
    const unsigned long long base = sb0 + (dsb1 & (unsigned long long)m1) + (dsb2 & (unsigned long long)m2) + (dsb3 & (unsigned long long)m3);
    const int c0 = cb0 + (dcb1 & (int)m1) + (dcb2 & (int)m2) + (dcb3 & (int)m3);
    return base + (unsigned long long)(unsigned)(gc - c0) * (unsigned long long)(unsigned)chan_bytes;
  };

  const int cpc = p.nchunks;
  const int NV = 4 * cpc;                     // virtual chunk = pair * cpc + 16-channel chunk
  constexpr int RW = BF ? 16 : 32;            // fp32: [channel][voxel of the quad]; bf16: [channel][dword: voxels (0, 1) | (2, 3)]
  float raw[FNR][RW];
  auto load_raw = [&](const bool live, const int pp, const int cc) {
    const unsigned cdelta = (unsigned)(((pp >> 1) * p.IY + (pp & 1)) * p.IX) * (unsigned)ESZ;   // wave-uniform
    unsigned so[FNR];
#pragma unroll
    for (int r = 0; r < FNR; r++) so[r] = ((fmask[r] >> pp) & 1u) ? fbase[r] + cdelta : 0xffffffffu;
    // (the usual case -- the wave's 8 channels lie in one tensor -- takes ONE slice lookup, behind a real branch: these kernels issue
    //  11 scalar instructions per MFMA (profiles/r04bf_pmc_instruction_mix_down1_fp32.json) on the CU's one scalar unit, and the
    //  per-channel lookups were a third of them)
    const int g0 = cc * HKC + sh * 8;   // wave-uniform
    auto loads = [&](const int c, const unsigned long long base) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, live && g0 + c < p.K ? chan_bytes : 0, 0x00020000);
#pragma unroll
      for (int r = 0; r < FNR; r++) {
        if constexpr (BF) {
          const auto t = __builtin_amdgcn_raw_buffer_load_b64(rs, so[r], 0, 0);
          raw[r][2 * c] = __builtin_bit_cast(float, (unsigned)t[0]);
          raw[r][2 * c + 1] = __builtin_bit_cast(float, (unsigned)t[1]);
        } else {
          const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, so[r], 0, 0);
#pragma unroll
          for (int vx = 0; vx < 4; vx++) raw[r][c * 4 + vx] = __builtin_bit_cast(float, (unsigned)t[vx]);
        }
      }
    };
    const int first = g0 < p.K ? g0 : p.K - 1, last = g0 + 7 < p.K ? g0 + 7 : p.K - 1;
    if (__builtin_expect(slice_of(first) == slice_of(last), 1)) {
      unsigned long long base = chan_base(first);
      split_pin_scalar(base);
#pragma unroll
      for (int c = 0; c < 8; c++) loads(c, base + (unsigned long long)c * (unsigned)chan_bytes);
    } else {
#pragma unroll
      for (int c = 0; c < 8; c++) loads(c, chan_base(g0 + c < p.K ? g0 + c : p.K - 1));
    }
  };
  // maxima: as in the class form (the pairs together see every element; a quad's columns beyond the halo belong to the same
  // tensor, so the tile scale is only more cautious and the exported maxima stay exact)
  const bool export_max = !BF && p.amax_out != nullptr && nblk == 0;
  float rmax0 = 0.f, rmax1 = 0.f, rmax2 = 0.f, rmax3 = 0.f;
  auto publish_max = [&](const int parity, const int cc) {
    if constexpr (BF) return;
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < FNR; r++)
#pragma unroll
      for (int c = 0; c < RW; c += 2) m = fmaxf(fmaxf(m, fabsf(raw[r][c])), fabsf(raw[r][c + 1]));
    m = split_wave_max(m);
    if (lane == 0) xmax[parity * 4 + wave] = m;
    if (export_max) {
      const int gc0 = cc * HKC + sh * 8;
      const int sa = slice_of(gc0 < p.K ? gc0 : p.K - 1), sb = slice_of(gc0 + 7 < p.K ? gc0 + 7 : p.K - 1);
      auto credit = [&](const int sl, const float mv) {
        rmax0 = sl == 0 ? fmaxf(rmax0, mv) : rmax0;
        rmax1 = sl == 1 ? fmaxf(rmax1, mv) : rmax1;
        rmax2 = sl == 2 ? fmaxf(rmax2, mv) : rmax2;
        rmax3 = sl == 3 ? fmaxf(rmax3, mv) : rmax3;
      };
      if (sa == sb) {
        credit(sa, m);
      } else {
#pragma unroll
        for (int c = 0; c < 8; c++) {
          float mc = 0.f;
#pragma unroll
          for (int r = 0; r < FNR; r++)
#pragma unroll
            for (int vx = 0; vx < 4; vx++) mc = fmaxf(mc, fabsf(raw[r][c * 4 + vx]));
          mc = split_wave_max(mc);
          credit(slice_of(gc0 + c < p.K ? gc0 + c : p.K - 1), mc);
        }
      }
    }
  };
  auto next_scale = [&](const int parity, const int s_run) {
    if constexpr (BF) return 0;
    const float m = fmaxf(fmaxf(xmax[parity * 4 + 0], xmax[parity * 4 + 1]), fmaxf(xmax[parity * 4 + 2], xmax[parity * 4 + 3]));
    const int s_c = __builtin_amdgcn_readfirstlane(split_scale_exp(m));
    return s_c < s_run ? s_c : s_run;
  };
  // voxel vx of a quad: x parity vx & 1 -> plane set, halo column 2 q - 1 + (vx >> 1); a 16-byte piece = this wave's 8 channels
  auto split_and_write = [&](const float in_mult) {
#pragma unroll
    for (int r = 0; r < FNR; r++)
#pragma unroll
      for (int vx = 0; vx < 4; vx++) {
        unsigned char* dst = Hs + (vx & 1) * G::PXB + sh * HPLANE + fwr[r] + (vx >> 1) * 16;
        if constexpr (BF) {
          u32x4 pc;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const unsigned a = __builtin_bit_cast(unsigned, raw[r][2 * (2 * k) + (vx >> 1)]);
            const unsigned bq = __builtin_bit_cast(unsigned, raw[r][2 * (2 * k + 1) + (vx >> 1)]);
            pc[k] = __builtin_amdgcn_perm(bq, a, (vx & 1) ? 0x07060302u : 0x05040100u);   // channels 2k, 2k + 1 at voxel vx
          }
          if ((fvox[r] >> vx) & 1) *reinterpret_cast<u32x4*>(dst) = pc;
        } else {
          h8 hi, lo;
          split_piece(&raw[r][vx], 4, in_mult, hi, lo);
          if ((fvox[r] >> vx) & 1) {
            *reinterpret_cast<h8*>(dst) = hi;
            *reinterpret_cast<h8*>(dst + 2 * HPLANE) = lo;
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // one voxel at a time: its pieces die before the next one's are built
      }
  };

  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.wimg) + (size_t)(p.nb_off + nblk) * p.blk_stride;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (int)p.blk_stride, 0x00020000);
  auto dma_w = [&](const int woff, const int npieces, unsigned char* W) {
#pragma unroll
    for (int ii = 0; ii < (TPP * NP * RT + 3) / 4; ii++) {
      const int i = wave + 4 * ii;
      if (i < npieces) split_lds_dma16(wrs, (lds_p)(W + i * 1024), woff + i * 1024 + lane * 16);
    }
  };

  f32x16 acc[RT][2];
#pragma unroll
  for (int i = 0; i < RT; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  int bbase[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int vt = 2 * wave + j;
    bbase[j] = (lane >> 5) * HPLANE + (((vt >> 2) * HHY + (vt & 3)) * HHX + (lane & 31)) * 16;
  }
  const int abase = lane * 16;

  // ---- prologue
  // IM2COL TAIL (sr3d_hconv.hip): when the last 16-channel chunk holds 1 - 2 channels (every stride-2 layer of the model: K = 64 / 128 /
  // 256 features + the mask) the K = 16 of an MFMA are taps of ONE of them -- the T <= 12 taps of the pair in list order -- instead
  // of 16 channels at one tap: nt groups per (pair, tail chunk) instead of T, 4 nt instead of 27 per tail chunk (K = 65: 5 chunks
  // of work became 4.15).  The B operand is gathered with 2-byte LDS reads at the lane's 8 tap offsets.
  const bool itail = p.itail != 0;
  const int nt = p.K & 15;
  int pp = 0, cc_cur = 0;
  auto slots_of = [&](const int pp_, const int cc_) { return (itail && cc_ + 1 == cpc) ? nt : pair_taps(pp_); };   // MFMA groups of a chunk
  int cbase = 0, gph = 0;                        // byte offset of the current chunk's first piece (a chunk always OWNS T pieces)
  constexpr int PIECE = NP * RT * 1024;
  {
    const int s0 = slots_of(0, 0);
    dma_w(0, (s0 < TPP ? s0 : TPP) * NP * RT, Ws);
  }
  load_raw(true, 0, 0);
  publish_max(0, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int s_run = next_scale(0, kSplitScaleNone);
  split_and_write(ldexpf(1.f, s_run));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int vc = 0; vc < NV; vc++) {
    const int pz = pp >> 1, py = pp & 1;
    const int T = pair_taps(pp);
    const bool tailc = itail && cc_cur + 1 == cpc;
    const int Tc = tailc ? nt : T;               // MFMA groups of this chunk
    const int nph = (Tc + TPP - 1) / TPP;
    int cc_n = cc_cur + 1, pp_n = pp;
    if (cc_n == cpc) cc_n = 0, pp_n = pp + 1;
    const int Tc_n = slots_of(pp_n > 3 ? 3 : pp_n, cc_n);
    // lane t of `hktab`: halo byte offset of tap t of this pair's list (plane set of its x class, (hz, hy, hx)); a tap of the
    // phase loop then costs ONE v_readlane instead of ~20 scalar instructions of decoding (t / 3, parities, tap_h ...)
    int hktab;
    {
      const int kk = lane < T ? lane : 0;
      const int g = kk / 3, xt = kk - 3 * g;
      const int iz = g >> py, iy = g & py;
      hktab = (xt != 0 ? G::PXB : 0) + ((tap_h(1, pz, iz) * HHY + tap_h(1, py, iy)) * HHX + (xt == 1 ? 0 : 1)) * 16;
    }
    int goff[8];                                 // tail: halo offsets of the lane's 8 taps (K half lane >> 5) in this pair's list
    if (tailc) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        int kk = 8 * (lane >> 5) + i;
        kk = kk < T ? kk : 0;                    // (beyond the list: any valid voxel, the weights are zero)
        const int g = kk / 3, xt = kk - 3 * g;
        const int iz = g >> py, iy = g & py;
        goff[i] = (xt != 0 ? G::PXB : 0) + ((tap_h(1, pz, iz) * HHY + tap_h(1, py, iy)) * HHX + (xt == 1 ? 0 : 1)) * 16;
      }
    }
    for (int ph = 0; ph < nph; ph++, gph++) {
      const unsigned char* W = Ws + (gph & 1) * G::WBUF + abase;
      const int ntap = Tc - TPP * ph < TPP ? Tc - TPP * ph : TPP;
      const bool last = ph + 1 == nph;
      const int ntap_n = last ? (Tc_n < TPP ? Tc_n : TPP) : (Tc - TPP * (ph + 1) < TPP ? Tc - TPP * (ph + 1) : TPP);
      const int woff_n = last ? cbase + T * PIECE : cbase + TPP * (ph + 1) * PIECE;
      if (!(last && vc + 1 == NV)) dma_w(woff_n, ntap_n * NP * RT, Ws + ((gph + 1) & 1) * G::WBUF);
      __builtin_amdgcn_sched_barrier(0);   // (the wait below counts on the DMA being older than the raw rows)
      if (ph == 0) load_raw(vc + 1 < NV, pp_n, cc_n);
#pragma unroll
      for (int ix = 0; ix < TPP; ix++) {
        if (ix < ntap) {
          const int t = TPP * ph + ix;     // tap of the pair's list (tail chunk: tail channel)
          h8 fa[NP][RT], fb[NP][2];
#pragma unroll
          for (int part = 0; part < NP; part++)
#pragma unroll
            for (int i = 0; i < RT; i++) fa[part][i] = *reinterpret_cast<const h8*>(W + ((ix * NP + part) * RT + i) * 1024);
          if (tailc) {
#pragma unroll
            for (int part = 0; part < NP; part++)
#pragma unroll
              for (int j = 0; j < 2; j++) {
                const unsigned char* hb = Hs + part * (2 * HPLANE) + (bbase[j] - (lane >> 5) * HPLANE) + t * 2;   // channel half 0, channel t
                u32x4 pk;
#pragma unroll
                for (int k = 0; k < 4; k++)
                  pk[k] = (unsigned)*reinterpret_cast<const unsigned short*>(hb + goff[2 * k]) |
                          ((unsigned)*reinterpret_cast<const unsigned short*>(hb + goff[2 * k + 1]) << 16);
                fb[part][j] = __builtin_bit_cast(h8, pk);
              }
          } else {
            const unsigned char* Hk = Hs + __builtin_amdgcn_readlane(hktab, t);   // (t is wave-uniform)
#pragma unroll
            for (int part = 0; part < NP; part++)
#pragma unroll
              for (int j = 0; j < 2; j++) fb[part][j] = *reinterpret_cast<const h8*>(Hk + part * (2 * HPLANE) + bbase[j]);
          }
#pragma unroll
          for (int i = 0; i < RT; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
              if constexpr (BF) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[0][i]), __builtin_bit_cast(bf8, fb[0][j]),
                                                                   acc[i][j], 0, 0, 0);
              } else {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[NP - 1][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[NP - 1][i], fb[0][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
              }
            }
        }
      }
      // (as in the class form: the chunk's last phase publishes the maxima of the next chunk's rows)
      if (last && vc + 1 < NV) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish_max((vc + 1) & 1, cc_n);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else if (ph == 0) {   // next phase's weights landed (older than the NRAW raw-row loads issued behind them)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NRAW) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    cbase += T * PIECE;
    if (vc + 1 < NV) {
      const int s_next = next_scale((vc + 1) & 1, s_run);
      const bool turn = (((vc + 1) >> S2FLIP_SH) ^ (vc >> S2FLIP_SH)) & 1;
      if (!BF && (turn || s_next != s_run)) {
        const float flip = ldexpf(turn ? -1.f : 1.f, s_next - s_run);
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
          for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] *= flip;
      }
      s_run = s_next;
      split_and_write(ldexpf(1.f, s_run));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    cc_cur = cc_n;
    pp = pp_n;
  }

  if (export_max && lane == 0) {
    unsigned* slot = p.amax_out + ((v / p.nblk) & 63);
    if (rmax0 > 0.f) atomicMax(slot, __float_as_uint(rmax0));
    if (rmax1 > 0.f) atomicMax(slot + 64, __float_as_uint(rmax1));
    if (rmax2 > 0.f) atomicMax(slot + 128, __float_as_uint(rmax2));
    if (rmax3 > 0.f) atomicMax(slot + 192, __float_as_uint(rmax3));
  }
  const float out_mult = ldexpf((!BF && (((NV - 1) >> S2FLIP_SH) & 1)) ? -1.f : 1.f, -((s_run == kSplitScaleNone ? 0 : s_run) + sw));
  s2_epilogue<RT, 1, BF>(p, acc, out_mult, wave, lane, b, nblk, 0, z0, y0, x0, TZ, TY, TX);
}

// ---- INPUT GRADIENT WITH BOTH X CLASSES PER WORKGROUP (round 4; quad-load conditions of the class form).  The eight output
// classes of the class form each stage the SAME dY halo (8 launches slices x 12 quad loads per lane and 16 channels, for 1 .. 8
// taps); here blockIdx.z = (qz, qy) and a workgroup computes dx at x = 2 i AND 2 i + 1 from one halo: per (kz, ky) of the pair
// three x taps -- k = 1 -> even x from dy[i]; k = 0 -> odd x from dy[i + 1]; k = 2 -> odd x from dy[i] (the B fragment of the
// first) -- into two accumulator sets: half the vector-memory instructions and half the L2 reads per MFMA, one phase (barrier,
// weight DMA) per three taps, and the two classes leave as ONE 8-byte (bf16: 4-byte) store per lane instead of two 4-byte
// stores at an 8-byte pitch from different workgroups.  128 accumulator registers: two workgroups per CU.
template <int RT, bool BF>
struct PGeo {
  static constexpr int NP = BF ? 1 : 2;
  // (kz, ky) groups of three x taps per phase (= per barrier, weight DMA, wait): two -- with one group per phase the waves stood
  // parked at barriers a third (bf16: half) of the time (PMC, profiles/r04bf_*.json); down1.0 3.88 -> 3.76 ms fp32, 2.31 -> 2.26 ms
  // bf16 (profiles/r04bm_ab_s2_bwd_groups_per_phase.log).  Split form: 32 + 48 KB of LDS, still two workgroups per CU.
  static constexpr int GPP = BF ? S2B_GPP_BF : S2B_GPP_F32;
  static constexpr int WBUF = GPP * 3 * NP * RT * 1024;
  static constexpr int HB = NP * 2 * HPLANE;
  static constexpr size_t LDS = HB + 2 * (size_t)WBUF;
};
static_assert(2 * PGeo<2, false>::LDS <= 160 * 1024, "LDS budget of the paired input gradient");

template <int RT, bool BF>
__global__ __launch_bounds__(HNT, 2) void hconv_s2_bwd_pair_kernel(const SrHconvS2Params p) {
  using G = PGeo<RT, BF>;
  constexpr int NP = G::NP;
  constexpr int ESZ = BF ? 2 : 4;
  constexpr int QNR = 3, QPR = 9;
  constexpr int NRAW = 4 * QNR;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Hs = lds;
  unsigned char* Ws = lds + G::HB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < HNT / 64);
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
  const int pq = (int)blockIdx.z;                 // 2 qz + qy
  const int qz = pq >> 1, qy = pq & 1;
  const int c0 = 4 * qz + 2 * qy;                 // the even-x class of the pair
  const int TZ = p.cZ[c0], TY = p.cY[c0], TX0 = p.cX[c0], TX1 = p.cX[c0 + 1];
  const int z0 = tiz * 2, y0 = tiy * 4, x0 = tix * 32;
  if (z0 >= TZ || y0 >= TY || x0 >= TX0) return;  // (the grid is sized for the largest class)
  const long long IZYX = (long long)p.IZ * p.IY * p.IX;
  const int chan_bytes = (int)(IZYX * ESZ);

  int sw = BF ? 0 : split_scale_exp(*p.absmax_w);
  if (sw == kSplitScaleNone) sw = 0;
  float* xmax = reinterpret_cast<float*>(Hs + HVOX * 16);

  unsigned qoff[QNR];
  int qwr[QNR], qmask[QNR];
#pragma unroll
  for (int r = 0; r < QNR; r++) {
    const int t = r * 64 + lane;
    const int row = t / QPR, q = t - row * QPR;
    const int hz = row / UY, hy = row - hz * UY;
    const int gz = z0 + hz, gy = y0 + hy, gx = x0 + 4 * q;
    const bool task = t < UZ * UY * QPR;
    const bool ok = task && (unsigned)gz < (unsigned)p.IZ && (unsigned)gy < (unsigned)p.IY && (unsigned)gx < (unsigned)p.IX;
    qoff[r] = ok ? (unsigned)((gz * p.IY + gy) * p.IX + gx) * (unsigned)ESZ : 0xffffffffu;
    qwr[r] = ((hz * HHY + hy) * HHX + 4 * q) * 16;
    qmask[r] = !task ? 0 : (q == QPR - 1 ? 1 : 15);
  }

#define SR3D_SLICE_BASE(i) (reinterpret_cast<unsigned long long>(p.in.ptr[i]) + (unsigned long long)((long long)b * p.in.bstride[i]) * ESZ)
  unsigned long long sb0 = SR3D_SLICE_BASE(0), sb1 = SR3D_SLICE_BASE(1), sb2 = SR3D_SLICE_BASE(2), sb3 = SR3D_SLICE_BASE(3);
#undef SR3D_SLICE_BASE
  int cb0 = p.in.cbeg[0], cb1 = p.in.cbeg[1], cb2 = p.in.cbeg[2], cb3 = p.in.cbeg[3];
  split_pin_scalar(sb0), split_pin_scalar(sb1), split_pin_scalar(sb2), split_pin_scalar(sb3);
  split_pin_scalar(cb0), split_pin_scalar(cb1), split_pin_scalar(cb2), split_pin_scalar(cb3);
  unsigned long long dsb1 = sb1 - sb0, dsb2 = sb2 - sb1, dsb3 = sb3 - sb2;
  int dcb1 = cb1 - cb0, dcb2 = cb2 - cb1, dcb3 = cb3 - cb2;
  split_pin_scalar(dsb1), split_pin_scalar(dsb2), split_pin_scalar(dsb3), split_pin_scalar(dcb1), split_pin_scalar(dcb2), split_pin_scalar(dcb3);
  auto chan_base = [&](const int gc) {
    const long long m1 = -(long long)(gc >= cb1), m2 = -(long long)(gc >= cb2), m3 = -(long long)(gc >= cb3);
    const unsigned long long base = sb0 + (dsb1 & (unsigned long long)m1) + (dsb2 & (unsigned long long)m2) + (dsb3 & (unsigned long long)m3);
    const int cc0 = cb0 + (dcb1 & (int)m1) + (dcb2 & (int)m2) + (dcb3 & (int)m3);
    return base + (unsigned long long)(unsigned)(gc - cc0) * (unsigned long long)(unsigned)chan_bytes;
  };

  const int NV = p.nchunks;
  constexpr int RW = BF ? 8 : 16;    // fp32: [channel 0..3][voxel]; bf16: [channel][dword: voxels (0, 1) | (2, 3)]
  float rawq[QNR][RW];
  auto load_raw = [&](const bool live, const int cc) {
    // (one slice lookup for the wave's 4 channels behind a branch, as in the paired forward, measured slightly SLOWER here:
    //  profiles/r04bg_ab_s2_one_slice.log)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int gc = cc * HKC + wave * 4 + c;   // wave-uniform
      const unsigned long long base = chan_base(gc < p.K ? gc : p.K - 1);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, live && gc < p.K ? chan_bytes : 0, 0x00020000);
#pragma unroll
      for (int r = 0; r < QNR; r++) {
        if constexpr (BF) {
          const auto t = __builtin_amdgcn_raw_buffer_load_b64(rs, qoff[r], 0, 0);
          rawq[r][c * 2 + 0] = __builtin_bit_cast(float, (unsigned)t[0]);
          rawq[r][c * 2 + 1] = __builtin_bit_cast(float, (unsigned)t[1]);
        } else {
          const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, qoff[r], 0, 0);
#pragma unroll
          for (int vx = 0; vx < 4; vx++) rawq[r][c * 4 + vx] = __builtin_bit_cast(float, (unsigned)t[vx]);
        }
      }
    }
  };
  auto publish_max = [&](const int parity) {
    if constexpr (BF) return;
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < QNR; r++)
#pragma unroll
      for (int c = 0; c < RW; c += 2) m = fmaxf(fmaxf(m, fabsf(rawq[r][c])), fabsf(rawq[r][c + 1]));
    m = split_wave_max(m);
    if (lane == 0) xmax[parity * 4 + wave] = m;
  };
  auto next_scale = [&](const int parity, const int s_run) {
    if constexpr (BF) return 0;
    const float m = fmaxf(fmaxf(xmax[parity * 4 + 0], xmax[parity * 4 + 1]), fmaxf(xmax[parity * 4 + 2], xmax[parity * 4 + 3]));
    const int s_c = __builtin_amdgcn_readfirstlane(split_scale_exp(m));
    return s_c < s_run ? s_c : s_run;
  };
  // channels 4 w .. 4 w + 3 = bytes 8 (w & 1) .. + 7 of the 16-byte piece of channel half w >> 1
  auto split_and_write = [&](const float in_mult) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    unsigned char* H0 = Hs + (wave >> 1) * HPLANE + (wave & 1) * 8;
#pragma unroll
    for (int r = 0; r < QNR; r++)
#pragma unroll
      for (int vx = 0; vx < 4; vx++) {
        if constexpr (BF) {
          const unsigned sel = (vx & 1) ? 0x07060302u : 0x05040100u;
          const unsigned a0 = __builtin_bit_cast(unsigned, rawq[r][0 * 2 + (vx >> 1)]), a1 = __builtin_bit_cast(unsigned, rawq[r][1 * 2 + (vx >> 1)]);
          const unsigned a2 = __builtin_bit_cast(unsigned, rawq[r][2 * 2 + (vx >> 1)]), a3 = __builtin_bit_cast(unsigned, rawq[r][3 * 2 + (vx >> 1)]);
          if ((qmask[r] >> vx) & 1)
            *reinterpret_cast<u32x2*>(H0 + qwr[r] + vx * 16) = u32x2{__builtin_amdgcn_perm(a1, a0, sel), __builtin_amdgcn_perm(a3, a2, sel)};
        } else {
          unsigned h0, l0, h1, l1;
          split_pair(rawq[r][0 * 4 + vx], rawq[r][1 * 4 + vx], in_mult, h0, l0);
          split_pair(rawq[r][2 * 4 + vx], rawq[r][3 * 4 + vx], in_mult, h1, l1);
          if ((qmask[r] >> vx) & 1) {
            *reinterpret_cast<u32x2*>(H0 + qwr[r] + vx * 16) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(H0 + 2 * HPLANE + qwr[r] + vx * 16) = u32x2{l0, l1};
          }
        }
      }
  };

  constexpr int GPP = G::GPP;
  const int T = pair_taps(pq), ngr = T / 3, nph = (ngr + GPP - 1) / GPP;   // (kz, ky) groups of the pair; phases of a chunk
  const long long blk_bytes = (long long)p.nchunks * T * NP * RT * 1024;
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.wimg) + p.cls_off[pq] + (size_t)(p.nb_off + nblk) * blk_bytes;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (int)blk_bytes, 0x00020000);
  auto dma_w = [&](const int woff, const int ngroups, unsigned char* W) {
#pragma unroll
    for (int ii = 0; ii < (GPP * 3 * NP * RT + 3) / 4; ii++) {
      const int i = wave + 4 * ii;
      if (i < ngroups * 3 * NP * RT) split_lds_dma16(wrs, (lds_p)(W + i * 1024), woff + i * 1024 + lane * 16);
    }
  };

  f32x16 acc[2][RT][2];   // [x class][row tile][voxel row]
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[c][i][j][r] = 0.f;
  int bbase[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int vt = 2 * wave + j;
    bbase[j] = (lane >> 5) * HPLANE + (((vt >> 2) * HHY + (vt & 3)) * HHX + (lane & 31)) * 16;
  }
  const int abase = lane * 16;
  constexpr int GSIZE = 3 * NP * RT * 1024;   // bytes of one group's weights

  int woff = 0, gph = 0;
  dma_w(0, ngr < GPP ? ngr : GPP, Ws);
  load_raw(true, 0);
  publish_max(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int s_run = next_scale(0, kSplitScaleNone);
  split_and_write(ldexpf(1.f, s_run));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int vc = 0; vc < NV; vc++) {
    for (int ph = 0; ph < nph; ph++, gph++) {
      const unsigned char* W = Ws + (gph & 1) * G::WBUF + abase;
      const bool last = ph + 1 == nph;
      const int ng = ngr - GPP * ph < GPP ? ngr - GPP * ph : GPP;                       // groups of this phase
      const int ng_n = last ? (ngr < GPP ? ngr : GPP) : (ngr - GPP * (ph + 1) < GPP ? ngr - GPP * (ph + 1) : GPP);
      if (!(last && vc + 1 == NV)) dma_w(woff + ng * GSIZE, ng_n, Ws + ((gph + 1) & 1) * G::WBUF);
      __builtin_amdgcn_sched_barrier(0);   // (the wait below counts on the DMA being older than the raw rows)
      if (ph == 0) load_raw(vc + 1 < NV, vc + 1);
#pragma unroll
      for (int gs = 0; gs < GPP; gs++) {
      if (gs >= ng) break;
      const int gi = ph * GPP + gs;            // group = iz * (1 + qy) + iy
      const int iz = gi >> qy, iy = gi & qy;
      const unsigned char* Hk = Hs + ((tap_h(2, qz, iz) * HHY + tap_h(2, qy, iy)) * HHX) * 16;
      const unsigned char* Wg = W + gs * GSIZE;
      // x taps: 0: k = 1, even x, dy[i]; 1: k = 0, odd x, dy[i + 1]; 2: k = 2, odd x, dy[i]
#pragma unroll
      for (int hx = 0; hx < 2; hx++) {
        h8 fb[NP][2];
#pragma unroll
        for (int part = 0; part < NP; part++)
#pragma unroll
          for (int j = 0; j < 2; j++) fb[part][j] = *reinterpret_cast<const h8*>(Hk + part * (2 * HPLANE) + bbase[j] + hx * 16);
#pragma unroll
        for (int xt = 0; xt < 3; xt++) {
          if ((xt == 1) != (hx == 1)) continue;
          h8 fa[NP][RT];
#pragma unroll
          for (int part = 0; part < NP; part++)
#pragma unroll
            for (int i = 0; i < RT; i++) fa[part][i] = *reinterpret_cast<const h8*>(Wg + ((xt * NP + part) * RT + i) * 1024);
          const int cx = xt == 0 ? 0 : 1;
#pragma unroll
          for (int i = 0; i < RT; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
              if constexpr (BF) {
                acc[cx][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[0][i]), __builtin_bit_cast(bf8, fb[0][j]),
                                                                       acc[cx][i][j], 0, 0, 0);
              } else {
                acc[cx][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[NP - 1][j], acc[cx][i][j], 0, 0, 0);
                acc[cx][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[NP - 1][i], fb[0][j], acc[cx][i][j], 0, 0, 0);
                acc[cx][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0][j], acc[cx][i][j], 0, 0, 0);
              }
            }
        }
      }
      }   // gs
      if (last && vc + 1 < NV) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish_max((vc + 1) & 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else if (ph == 0) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NRAW) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      woff += ng * GSIZE;
    }
    if (vc + 1 < NV) {
      const int s_next = next_scale((vc + 1) & 1, s_run);
      const bool turn = (((vc + 1) >> S2FLIP_SH) ^ (vc >> S2FLIP_SH)) & 1;
      if (!BF && (turn || s_next != s_run)) {
        const float flip = ldexpf(turn ? -1.f : 1.f, s_next - s_run);
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int i = 0; i < RT; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
              for (int r = 0; r < 16; r++) acc[c][i][j][r] *= flip;
      }
      s_run = s_next;
      split_and_write(ldexpf(1.f, s_run));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

  // ---- epilogue: dx[2 oz + qz][2 oy + qy][2 ox], [.. + 1] of GEMM row n
  const float out_mult = ldexpf((!BF && (((NV - 1) >> S2FLIP_SH) & 1)) ? -1.f : 1.f, -((s_run == kSplitScaleNone ? 0 : s_run) + sw));
  const int ox = x0 + (lane & 31);
  if (ox >= TX0) return;
  const bool has1 = ox < TX1;
  const int rblock = p.n_off + (p.nb_off + nblk) * 64;
  const long long TZYX = (long long)p.TZ_ * p.TY_ * p.TX_;
  const bool row_even = (p.TX_ & 1) == 0;          // then every (2 ox) element of a row is 8-byte (bf16: 4-byte) aligned with its tensor
#pragma unroll
  for (int i = 0; i < RT; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int n = rblock + i * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
      if (n >= p.N) continue;
      const int si = cat_find(p.out, n);
      float* base = cat_ptr(p.out, si);
      if (base == nullptr) continue;
      const long long boff = (long long)b * cat_bstride(p.out, si) + (long long)(n - cat_cbeg(p.out, si)) * TZYX;   // elements
      const float bv = p.bias ? p.bias[n] : 0.f;
      const bool vec = has1 && row_even && ((reinterpret_cast<uintptr_t>(base) & (BF ? 3 : 7)) == 0) && ((boff & 1) == 0);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int vt = 2 * wave + j;
        const int oz = z0 + (vt >> 2), oy = y0 + (vt & 3);
        if (oz >= TZ || oy >= TY) continue;
        const long long o = boff + ((long long)(oz * 2 + qz) * p.TY_ + (oy * 2 + qy)) * p.TX_ + 2 * ox;
        const float v0 = split_act(acc[0][i][j][r] * out_mult + bv, p.act), v1 = split_act(acc[1][i][j][r] * out_mult + bv, p.act);
        if (vec) {
          if constexpr (BF) {
            const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v0) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v1) << 16);
            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(base) + o) = pk;
          } else {
            *reinterpret_cast<float2*>(base + o) = float2{v0, v1};
          }
        } else {
          st_act<BF>(base, o, v0);
          if (has1) st_act<BF>(base, o + 1, v1);
        }
      }
    }
}

// ---- weight split + packing.  Image of one row block (MODE 1) / of one (class, row block) (MODE 2), in execution
// order: [class][chunk][phase (iz, iy)][ix][part][row tile][channel half][32 rows][8 ch] fp16
struct S2PackParams {
  const float* w1;
  const float* w2;
  const float* absmax_w;
  _Float16* img;
  int Cout, Cin, kind, K, N, cpc, nblk, RT, n_off, mode;
  int bf;   // 1: one bf16 part per weight, unscaled
  int itail;   // mode 3: the last chunk's 1 - 2 channels in im2col form (sr3d_hconv_s2_fwd_itail)
  int rbeg[SR3D_MAX_SRC + 1];
  int cbeg[SR3D_MAX_SRC];
};

__host__ __device__ inline int cls_taps(int cls) { return (1 + ((cls >> 2) & 1)) * (1 + ((cls >> 1) & 1)) * (1 + (cls & 1)); }
// taps of the classes before `cls` (classes in order 0..7)
__host__ __device__ inline int cls_taps_before(int cls) {
  int s = 0;
  for (int c = 0; c < cls; c++) s += cls_taps(c);
  return s;
}

__global__ __launch_bounds__(256) void hconv_s2_pack_kernel(const S2PackParams p) {
  const int sw = p.bf ? 0 : split_scale_exp(*p.absmax_w);
  const float w_mult = ldexpf(1.f, sw == kSplitScaleNone ? 0 : sw);
  // items: (row block, class, chunk, local tap, row tile, channel half, row); 27 taps over the 8 classes
  const long long total = (long long)p.nblk * p.cpc * 27 * p.RT * 64;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    long long r = e;
    const int tg = r % 27;     // global tap slot: class by cumulative tap count
    r /= 27;
    const int row = r % 32;
    r /= 32;
    const int h = r % 2;
    r /= 2;
    const int rt = r % p.RT;
    r /= p.RT;
    const int cc = r % p.cpc;
    const int nb = r / p.cpc;
    int cls = 0, tl, tap, vc;
    bool itc = false;   // im2col tail chunk of the paired forward: the item is (tail channel tl, K half h), element j = tap 8 h + j of the pair's list
    if (p.mode >= 3) {   // paired forward (3) / paired input gradient (4): `cls` = pair 2 pz + py, local tap = 3 (iz * ny + iy) + x tap (k = 1 | 0 | 2)
      while (cls < 3 && tg >= pair_taps_before(cls + 1)) cls++;
      tl = tg - pair_taps_before(cls);
      itc = p.mode == 3 && p.itail != 0 && cc + 1 == p.cpc;
      if (itc && tl >= (p.K & 15)) continue;
      const int pz = cls >> 1, py = cls & 1, g = tl / 3, xt = tl - 3 * g;
      const int iz = g >> py, iy = g & py;
      tap = (tap_k(pz, iz) * 3 + tap_k(py, iy)) * 3 + (xt == 0 ? 1 : xt == 1 ? 0 : 2);
      vc = p.mode == 3 ? cls * p.cpc + cc : cc;
    } else {
      while (cls < 7 && tg >= cls_taps_before(cls + 1)) cls++;
      tl = tg - cls_taps_before(cls);                       // local tap = (iz * ny + iy) * nx + ix
      const int pz = (cls >> 2) & 1, py = (cls >> 1) & 1, px = cls & 1;
      const int nx = 1 + px, ny = 1 + py;
      const int ix = tl % nx, iy = (tl / nx) % ny, iz = tl / (nx * ny);
      tap = (tap_k(pz, iz) * 3 + tap_k(py, iy)) * 3 + tap_k(px, ix);   // original (kz, ky, kx)
      // execution-order chunk index (sign alternation): MODE 1: class * cpc + cc; MODE 2: cc (one class per image)
      vc = p.mode == 1 ? cls * p.cpc + cc : cc;
    }
    const int n = p.n_off + nb * (32 * p.RT) + rt * 32 + row;
    const float* w = nullptr;
    long long kstride = 27;
    if (p.kind == SR3D_PACK_FWD) {
      if (n < p.N) w = p.w1 + (long long)n * p.Cin * 27;
    } else if (p.kind == SR3D_PACK_FWD_GATED) {
      const int co = (n >> 6) * 32 + (n & 31);
      if (co < p.Cout) w = ((n & 32) ? p.w2 : p.w1) + (long long)co * p.Cin * 27;
    } else if (n < p.N) {
      const int si = (n >= p.rbeg[1]) + (n >= p.rbeg[2]) + (n >= p.rbeg[3]);
      const int ci = p.cbeg[si] + (n - p.rbeg[si]);
      w = p.w1 + (long long)ci * 27;
      kstride = (long long)p.Cin * 27;
    }
    h8 hi, lo;
    bf8 wb;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      int k = cc * HKC + h * 8 + j;
      int tapj = tap;
      if (itc) {   // (forward kinds only)
        const int kk = 8 * h + j, pz = cls >> 1, py = cls & 1;
        k = kk < pair_taps(cls) ? cc * HKC + tl : p.K;   // beyond the pair's list: zero
        const int g = kk / 3, xt = kk - 3 * g, iz = g >> py, iy = g & py;
        tapj = (tap_k(pz, iz) * 3 + tap_k(py, iy)) * 3 + (xt == 0 ? 1 : xt == 1 ? 0 : 2);
      }
      float val = 0.f;
      if (w != nullptr && k < p.K) {
        if (p.kind == SR3D_PACK_BWD || p.kind == SR3D_PACK_BWD_GATED) {
          const float* src = k < p.Cout ? w + (long long)k * kstride : (p.w2 + (w - p.w1)) + (long long)(k - p.Cout) * kstride;
          val = src[tapj];
        } else {
          val = w[(long long)k * kstride + tapj];
        }
      }
      const float s = val * ((((vc >> S2FLIP_SH) & 1) && !p.bf) ? -w_mult : w_mult);
      const _Float16 a = (_Float16)s;
      hi[j] = a;
      lo[j] = (_Float16)(s - (float)a);
      wb[j] = (__bf16)s;
    }
    // piece index inside the image
    long long piece;
    if (p.mode == 4) {   // the 4 pair images one after the other, each [row block][cc][local tap]
      piece = (long long)pair_taps_before(cls) * p.cpc * p.nblk + ((long long)nb * p.cpc + cc) * pair_taps(cls) + tl;
    } else if (p.mode == 3) {   // row block: pairs one after the other, each [cc][local tap][part][rt]
      piece = (long long)nb * p.cpc * 27 + (long long)pair_taps_before(cls) * p.cpc + (long long)cc * pair_taps(cls) + tl;
    } else if (p.mode == 1) {   // row block: classes one after the other, each [cc][local tap][part][rt]
      piece = (long long)nb * p.cpc * 27 + (long long)cls_taps_before(cls) * p.cpc + (long long)cc * cls_taps(cls) + tl;
    } else {             // the 8 class images one after the other, each [row block][cc][local tap]
      piece = (long long)cls_taps_before(cls) * p.cpc * p.nblk + ((long long)nb * p.cpc + cc) * cls_taps(cls) + tl;
    }
    if (p.bf) {
      *reinterpret_cast<bf8*>(p.img + (piece * p.RT + rt) * 512 + (h * 32 + row) * 8) = wb;
      continue;
    }
    _Float16* dst = p.img + (piece * 2 * p.RT) * 512 + (h * 32 + row) * 8;
    *reinterpret_cast<h8*>(dst + (0 * p.RT + rt) * 512) = hi;
    *reinterpret_cast<h8*>(dst + (1 * p.RT + rt) * 512) = lo;
  }
}

inline void row_split(int rows, int* n2, int* n1) {
  const int nfull = rows / 64, rem = rows - nfull * 64;
  *n2 = nfull + (rem > 32 ? 1 : 0);
  *n1 = (rem > 0 && rem <= 32) ? 1 : 0;
}

template <int MODE, bool BF, bool QD = false>
void launch_rt(int rt, dim3 grid, hipStream_t st, const SrHconvS2Params& p) {
  constexpr size_t lds2 = SGeo<2, BF>::LDS, lds1 = SGeo<1, BF>::LDS;
  if (rt == 2)
    hipLaunchKernelGGL((hconv_s2_kernel<2, MODE, BF, QD>), grid, dim3(HNT), lds2, st, p);
  else
    hipLaunchKernelGGL((hconv_s2_kernel<1, MODE, BF, QD>), grid, dim3(HNT), lds1, st, p);
}

template <bool BF>
int set_attrs() {
  constexpr int lds2 = (int)SGeo<2, BF>::LDS, lds1 = (int)SGeo<1, BF>::LDS;
  SR3D_HIP(hipFuncSetAttribute((const void*)hconv_s2_kernel<2, 1, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
  SR3D_HIP(hipFuncSetAttribute((const void*)hconv_s2_kernel<1, 1, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
  SR3D_HIP(hipFuncSetAttribute((const void*)hconv_s2_kernel<2, 2, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
  SR3D_HIP(hipFuncSetAttribute((const void*)hconv_s2_kernel<1, 2, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
  SR3D_HIP(hipFuncSetAttribute((const void*)hconv_s2_kernel<2, 2, BF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
  SR3D_HIP(hipFuncSetAttribute((const void*)hconv_s2_kernel<1, 2, BF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
  SR3D_HIP(hipFuncSetAttribute((const void*)hconv_s2_bwd_pair_kernel<2, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PGeo<2, BF>::LDS));
  SR3D_HIP(hipFuncSetAttribute((const void*)hconv_s2_bwd_pair_kernel<1, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PGeo<1, BF>::LDS));
  SR3D_HIP(hipFuncSetAttribute((const void*)hconv_s2_fwd_kernel<2, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FGeo<2, BF>::LDS));
  SR3D_HIP(hipFuncSetAttribute((const void*)hconv_s2_fwd_kernel<1, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FGeo<1, BF>::LDS));
  return SR3D_OK;
}

}  // namespace

// forward: both x parities per load (pack order 3) when the rows can be fetched as quads; SR3D_HCONV_S2_CLASS_FWD=1 keeps the class form
// paired forward: im2col tail for a last chunk of 1 - 2 channels (SR3D_HCONV_NO_ITAIL=1: off)
bool sr3d_hconv_s2_fwd_itail(int K) { return (K & 15) >= 1 && (K & 15) <= 2 && getenv("SR3D_HCONV_NO_ITAIL") == nullptr; }
bool sr3d_hconv_s2_fwd_paired(int IX) { return IX % 4 == 0 && getenv("SR3D_HCONV_S2_CLASS_FWD") == nullptr; }

// input gradient: both x classes per workgroup (pack order 4, launch mode 4) when the dY rows can be fetched as quads;
// SR3D_HCONV_S2_CLASS_BWD=1 keeps the class form
bool sr3d_hconv_s2_bwd_paired(const SrHconvS2Params& p, bool bf) {
  bool quad = p.IX % 4 == 0 && getenv("SR3D_HCONV_NO_PAIR") == nullptr && getenv("SR3D_HCONV_S2_CLASS_BWD") == nullptr;
  for (int i = 0; i < p.in.n; i++) quad = quad && (reinterpret_cast<uintptr_t>(p.in.ptr[i]) & (bf ? 7 : 15)) == 0;
  return quad;
}

// header (64 bytes) + region A (64-row blocks) + region B (one block of <= 32 rows); 27 taps per (row block, chunk)
size_t sr3d_hconv_s2_image_bytes(int rows, int K, bool bf) {
  int n2, n1;
  row_split(rows, &n2, &n1);
  return 64 + (size_t)ceil_div(K, HKC) * 27 * (bf ? 1 : 2) * 1024 * ((size_t)n2 * 2 + (size_t)n1);
}

int sr3d_hconv_s2_pack(int mode, int kind, int Cout, int Cin, int rows, int K, const float* w1, const float* w2,
                       const int* rbeg, const int* cbeg, void* image, bool bf, hipStream_t st) {
  unsigned* hdr = (unsigned*)image;
  SrProfScope prof(SR3D_PROF_PACK, (bf ? 3.0 : 4.0) * (double)rows * K * 27 * 2, st);
  const long long nw = (long long)Cout * Cin * 27;
  if (!bf) {
    if (int rc = sr3d_zero_words(hdr, 16, st)) return rc;
    if (int rc = sr3d_absmax_launch(w1, nw, hdr, st)) return rc;
    if (w2 != nullptr)
      if (int rc = sr3d_absmax_launch(w2, nw, hdr, st)) return rc;
  }
  S2PackParams p{};
  p.bf = bf ? 1 : 0;
  p.itail = (mode == 3 && sr3d_hconv_s2_fwd_itail(K)) ? 1 : 0;
  p.w1 = w1, p.w2 = w2, p.absmax_w = (const float*)hdr;
  p.Cout = Cout, p.Cin = Cin, p.kind = kind, p.K = K, p.N = rows, p.cpc = ceil_div(K, HKC), p.mode = mode;
  for (int i = 0; i <= SR3D_MAX_SRC; i++) p.rbeg[i] = rbeg ? rbeg[i] : INT_MAX;
  for (int i = 0; i < SR3D_MAX_SRC; i++) p.cbeg[i] = cbeg ? cbeg[i] : 0;
  int n2, n1;
  row_split(rows, &n2, &n1);
  _Float16* body = (_Float16*)((unsigned char*)image + 64);
  for (int region = 0; region < 2; region++) {
    p.nblk = region == 0 ? n2 : n1;
    if (p.nblk == 0) continue;
    p.RT = region == 0 ? 2 : 1;
    p.n_off = region == 0 ? 0 : n2 * 64;
    p.img = body + (region == 0 ? 0 : (size_t)n2 * p.cpc * 27 * (bf ? 1 : 2) * 2 * 512);
    const long long total = (long long)p.nblk * p.cpc * 27 * p.RT * 64;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(hconv_s2_pack_kernel, dim3(blocks), dim3(256), 0, st, p);
    SR3D_HIP(hipGetLastError());
  }
  return SR3D_OK;
}

// p: in, K, IZ/IY/IX, Z/Y/X (mode 1: output grid; mode 2: fine grid of dx), N, n_off, epilogue fields, TZ_/TY_/TX_
int sr3d_hconv_s2_launch(int mode, SrHconvS2Params& p, const void* image, int B, bool bf, hipStream_t st) {
  SR3D_CHECK((long long)p.IZ * p.IY * p.IX < (1ll << 29), SR3D_E_ARG, "split-f16 conv: more than 2^29 voxels per channel");
  SR3D_CHECK(B <= 65535, SR3D_E_ARG, "split-f16 conv: batch too large");
  p.absmax_w = (const float*)image;
  p.nchunks = ceil_div(p.K, HKC);
  int n2, n1;
  row_split(p.N, &n2, &n1);
  static SrPerDevice setup;   // (the attribute is per device, not per thread)
  if (int rc = setup.once([&]() -> int {
        if (int rc2 = set_attrs<false>()) return rc2;
        return set_attrs<true>();
      }))
    return rc;
  int gz, gy, gx;   // tile space of the launch (mode 2: class 0 = even positions, the largest)
  if (mode == 1) {
    gz = p.Z, gy = p.Y, gx = p.X;
  } else {
    for (int c = 0; c < 8; c++)
      p.cZ[c] = (p.Z - ((c >> 2) & 1) + 1) / 2, p.cY[c] = (p.Y - ((c >> 1) & 1) + 1) / 2, p.cX[c] = (p.X - (c & 1) + 1) / 2;
    gz = p.cZ[0], gy = p.cY[0], gx = p.cX[0];
  }
  p.ntz = ceil_div(gz, 2), p.nty = ceil_div(gy, 4), p.ntx = ceil_div(gx, 32);
  const long long nsp = (long long)p.ntz * p.nty * p.ntx;
  SR3D_CHECK(nsp * (n2 + n1) < (1ll << 31), SR3D_E_ARG, "split-f16 conv: grid too large");
  void* tok = nullptr;
  if (sr3d_prof_active()) {
    const double rows = p.epi == SR3D_EPI_GATED ? 2.0 * p.Cg : (double)p.N;
    double vox = 0;
    if (mode == 1) {
      vox = 27.0 * p.Z * p.Y * p.X;
    } else {
      for (int c = 0; c < 8; c++) vox += (double)cls_taps(c) * p.cZ[c] * p.cY[c] * p.cX[c];
    }
    sr3d_prof_begin(mode == 1 ? SR3D_PROF_IGEMM_S2 : SR3D_PROF_IGEMM_BWD_S2, 2.0 * p.K * rows * vox * B, st, &tok);
  }
  const unsigned char* body = (const unsigned char*)image + 64;
  const size_t cpc = p.nchunks;
  for (int region = 0; region < 2; region++) {
    const int nb = region == 0 ? n2 : n1, rt = region == 0 ? 2 : 1;
    if (nb == 0) continue;
    SrHconvS2Params q = p;
    q.nblk = nb, q.nb_off = 0;
    q.n_off = p.n_off + (region == 0 ? 0 : n2 * 64);
    const size_t np = bf ? 1 : 2;
    q.wimg = body + (region == 0 ? 0 : (size_t)n2 * cpc * 27 * np * 2 * 1024);
    const size_t piece = np * rt * 1024;   // bytes of one tap (parts x rt fragments)
    if (mode == 1) {
      q.blk_stride = (long long)(cpc * 27 * piece);
    } else {
      for (int c = 0; c < 8; c++) {
        q.cls_off[c] = (long long)((size_t)cls_taps_before(c) * cpc * nb * piece);
        q.cls_blk[c] = (long long)(cpc * cls_taps(c) * piece);
      }
      if (mode == 4)
        for (int c = 0; c < 4; c++) q.cls_off[c] = (long long)((size_t)pair_taps_before(c) * cpc * nb * piece);
    }
    const dim3 grid((unsigned)(nsp * nb), B, mode == 2 ? 8 : mode == 4 ? 4 : 1);
    if (mode == 4) {
      constexpr size_t l2b = PGeo<2, true>::LDS, l1b = PGeo<1, true>::LDS, l2f = PGeo<2, false>::LDS, l1f = PGeo<1, false>::LDS;
      if (bf) {
        if (rt == 2) hipLaunchKernelGGL((hconv_s2_bwd_pair_kernel<2, true>), grid, dim3(HNT), l2b, st, q);
        else hipLaunchKernelGGL((hconv_s2_bwd_pair_kernel<1, true>), grid, dim3(HNT), l1b, st, q);
      } else {
        if (rt == 2) hipLaunchKernelGGL((hconv_s2_bwd_pair_kernel<2, false>), grid, dim3(HNT), l2f, st, q);
        else hipLaunchKernelGGL((hconv_s2_bwd_pair_kernel<1, false>), grid, dim3(HNT), l1f, st, q);
      }
      continue;
    }
    if (mode == 1 && sr3d_hconv_s2_fwd_paired(p.IX)) {   // (the image was packed in pair order: sr3d_pack_weights asks the same question)
      q.itail = sr3d_hconv_s2_fwd_itail(p.K) ? 1 : 0;
      for (int i = 0; i < p.in.n; i++)
        SR3D_CHECK((reinterpret_cast<uintptr_t>(p.in.ptr[i]) & (bf ? 7 : 15)) == 0, SR3D_E_ARG,
                   "stride-2 forward: x_srcs[%d] must be %d-byte aligned (X %% 4 == 0: rows are fetched as quads)", i, bf ? 8 : 16);
      constexpr size_t l2b = FGeo<2, true>::LDS, l1b = FGeo<1, true>::LDS, l2f = FGeo<2, false>::LDS, l1f = FGeo<1, false>::LDS;
      if (bf) {
        if (rt == 2) hipLaunchKernelGGL((hconv_s2_fwd_kernel<2, true>), grid, dim3(HNT), l2b, st, q);
        else hipLaunchKernelGGL((hconv_s2_fwd_kernel<1, true>), grid, dim3(HNT), l1b, st, q);
      } else {
        if (rt == 2) hipLaunchKernelGGL((hconv_s2_fwd_kernel<2, false>), grid, dim3(HNT), l2f, st, q);
        else hipLaunchKernelGGL((hconv_s2_fwd_kernel<1, false>), grid, dim3(HNT), l1f, st, q);
      }
    } else if (mode == 1)
      bf ? launch_rt<1, true>(rt, grid, st, q) : launch_rt<1, false>(rt, grid, st, q);
    else {
      // quad loads of the dY rows: IX % 4 == 0 and 16-byte (bf16: 8-byte) aligned tensors (then every channel row is)
      bool quad = p.IX % 4 == 0 && getenv("SR3D_HCONV_NO_PAIR") == nullptr;
      for (int i = 0; i < p.in.n; i++) quad = quad && (reinterpret_cast<uintptr_t>(p.in.ptr[i]) & (bf ? 7 : 15)) == 0;
      if (bf)
        quad ? launch_rt<2, true, true>(rt, grid, st, q) : launch_rt<2, true, false>(rt, grid, st, q);
      else
        quad ? launch_rt<2, false, true>(rt, grid, st, q) : launch_rt<2, false, false>(rt, grid, st, q);
    }
  }
  sr3d_prof_end(tok, st);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}
