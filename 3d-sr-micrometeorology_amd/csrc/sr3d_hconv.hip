// Stride-1 3x3x3 convolution (forward and input gradient) with fp32 operands split into two fp16 halves, on
// v_mfma_f32_16x16x32_f16.  Default for the layers that fill the chip (use_hconv in sr3d_igemm.hip); SR3D_SPLIT_F16=0
// puts every stride-1 layer back on the fp32 Winograd kernel, =2 forces this kernel for every eligible layer.
//
// Why: gfx950's fp32 MFMA runs at 1/16 of the f16 rate.  An fp32 value scaled into [2^13, 2^14) splits EXACTLY into
// hi = fp16(a) and lo = fp16(a - hi) with |a - hi - lo| <= 2^-22 |a|, and
//        a * b  =  hi_a hi_b + hi_a lo_b + lo_a hi_b  (+ lo_a lo_b ~ 2^-22 |a b|, dropped)
// accumulated in the MFMA's fp32 accumulator: three f16 MFMAs (32x32 x K = 16 in 32 cycles, or two 16x16 x K = 32 in 16
// cycles each) replace eight fp32 MFMAs (K = 2, 64 cycles), 5.3x less matrix-pipe time per product at a per-product error of ~2^-21, the size of fp32's own
// rounding in a 27 K-term sum.  The Winograd form of this does not fit the CU (K = 16 channels of V for 5 planes x
// 16 points x 32 tiles are 164 KB of LDS), so this is a DIRECT implicit GEMM: 81 f16 MFMAs per (32 rows x 32 voxels
// x 16 channels) = 2592 cycles, against 6144 for fp32 Winograd.
//
// Scaling (block floating point): every workgroup scales each 16-channel chunk of ITS halo tile by an exact power of
// two taken from the largest magnitude it has seen so far in its tile (wave maxima meet in LDS at a barrier that is
// there anyway), so that this magnitude lands in [2^13, 2^14): no fp16 overflow for any input, gradients of 1e-9 keep
// their 22 bits, elements 2^-14 below the local maximum lose only what is below 2^-36 of it.  When a larger chunk
// arrives the accumulators are rescaled by the (exact) ratio.  The weights carry one power of two per layer from
// max|w|, found at packing time.  The epilogue multiplies by 2^-(sx+sw).  (The first version took max|x| over the
// whole tensor in a separate HBM pass: 10 ms per training step.)
//
// One 256-thread workgroup, TWO per CU (80 KB of LDS each): 64 (or 32) output rows x 2 x 4 x 32 voxels; wave w owns
// voxel rows 2w, 2w+1 for all row tiles.  Chunk = 16 channels = K of one MFMA.  The two workgroups of a CU are
// independent, so the barriers and the halo refill of one are covered by the MFMAs of the other.  (One 512-thread
// workgroup per CU on 4 x 4 x 32 voxels measured the same 58-60 % matrix-pipe utilisation: the kernel is bound by the
// part's power limit; the small workgroups stayed because smaller launches still fill the chip.)
// BF = true (sr3d_conv_desc_t.dtype = SR3D_DTYPE_BF16, BASELINE configs[4]): activations are STORED as bfloat16.  The same
// kernel then needs no split and no scaling -- bf16 has fp32's exponent range, and a bf16 x bf16 product is exact in the
// MFMA's fp32 accumulator: ONE v_mfma_f32_16x16x32_bf16 per tile and tap pair instead of three, half the halo bytes
// from HBM, half the LDS (one part), outputs rounded to bf16 (RNE) in the epilogue.  Weights are rounded to bf16 once
// per call by the packing kernel (fp32 master weights stay with the optimizer).
//   halo      [part][channel half][voxel 4x6x34][8 ch] fp16 (56 KB): raw fp32 rows come in by buffer loads with
//             hardware range checks one chunk ahead (registers), are scaled, split and written as 16-byte pieces;
//             a B fragment is one conflict-free ds_read_b128 at (voxel + tap) * 16
//   weights   split + packed once per call: [row block][chunk][tap pair 14][part][16-row tile][64 lanes][8 ch] (a lane of
//             the 16x16x32 A operand: row, K group = tap of the pair x channel half); one pair (8 KB) arrives by LDS-DMA
//             while the previous one is multiplied (two buffers)
#include "sr3d_split_f16.h"

#include <limits.h>
#include <utility>
#include <stdlib.h>


// timing-only ablation builds (results WRONG by construction), a bit mask: 1 no halo refill, 2 no weight DMA, 4 no barriers,
// 8 no weight-fragment reads, 16 no halo-fragment reads, 32 one MFMA of the three, 64 no raw-row loads (the rest of the refill
// stays), 128 no maxima / scale / split arithmetic, 256 no accumulator flip, 512 no halo write + chunk barrier (tools/abl_hconv.sh)
#ifndef HCONV_NWB_F32
#define HCONV_NWB_F32 2
#endif
#ifndef HCONV_SETPRIO
#define HCONV_SETPRIO 0
#endif
#ifndef HCONV_NWB_BF
#define HCONV_NWB_BF 3
#endif
#ifndef HCONV_ABL
#define HCONV_ABL 0
#endif

namespace {


constexpr int HKC = 16;                        // channels per chunk
constexpr int HFLIP_SH = 1;                    // the packed weights and the accumulators change sign every 2^HFLIP_SH chunks
constexpr int HHZ = 4, HHY = 6, HHX = 34;      // halo of the 2 x 4 x 32 voxel tile
constexpr int HVOX = HHZ * HHY * HHX;          // 816
constexpr int HNR = 7;                         // staging rounds: 2 waves per channel half x 7 x 64 voxels
constexpr int HVP = 2 * HNR * 64;              // voxels per plane, padded (896)
constexpr int HPLANE = HVP * 16;               // bytes of one (part, channel half) plane
constexpr int HBYTES = 4 * HPLANE;             // 57344 (split-f16: [hi | lo] x [channel half]; bf16: 2 planes)
constexpr int HNT = 256;
constexpr int HPH = 14;                        // weight phases per chunk: tap pairs (2p, 2p + 1); tap 27 is a zero dummy
template <int RT, bool BF = false>
struct HGeo {
  static constexpr int NP = BF ? 1 : 2;        // operand parts: [hi | lo], or the bf16 value itself
  static constexpr int PIECES = NP * 2 * RT;   // 1 KB fragments of one phase: [part][16-row tile]
  static constexpr int WPHASE = PIECES * 1024;
  static constexpr int HB = NP * 2 * HPLANE;   // halo bytes
  // Weight buffers: the LDS-DMA of a phase is issued NWB - 1 phases before its fragments are read.  The split form has 48
  // MFMAs (768 cycles) per phase, enough to cover one DMA (issue -> landed 250-400 cycles from L2); the bf16 form has 16
  // (256 cycles) and was bound by exactly that latency with two buffers (606 TFLOP/s): it runs three phases ahead.
  // (split form with THREE buffers, two phases ahead -- 80 KB per workgroup, the CU's 160 KB exactly -- was measured too:
  // vector-memory operations complete in issue order, so waiting for the next phase's weights also waits for the raw-row
  // loads of the next chunk issued in phase 0, and two phases of distance leave those in flight one phase longer.  No gain
  // (up1 forward 42.1 vs 42.5 ms, profiles/r03f_layers_hconv_weight_buffers.log): the loads cost issue slots, not latency.
  // -DHCONV_NWB_F32=3 builds it.)
  // (bf16, end of round 3: THREE buffers, two phases ahead -- 52 KB per workgroup, and with 168 VGPRs three workgroups share
  // a CU instead of two: the bf16 form is bound by synchronisation around short MFMA bursts, and like the stride-2 kernel it
  // lives on the overlap between workgroups)
  static constexpr int NWB = BF ? HCONV_NWB_BF : HCONV_NWB_F32;
  // Tap pairs per phase (= per barrier, per weight DMA, per wait).  bf16: TWO -- with 16 MFMAs per phase the ~45 scalar and
  // ~40 vector bookkeeping instructions of a phase and its barrier were 3 + 2.4 per MFMA and the matrix pipe stood at 35 %
  // (profiles/r03d_pmc_instruction_mix_up1_bf16.json; weight DMA distance, halo double-buffering and wider halo loads had
  // each moved it by < 10 %).
  static constexpr int PP = BF ? 2 : 1;
  static constexpr int NPH = 14 / PP;          // phases per chunk
  static constexpr int WPH = PP * WPHASE;      // bytes of one phase's weights = one weight buffer
  // Halo buffers.  bf16: two (2 x 28 KB): the next chunk's halo is written while the current one is multiplied, and the
  // chunk boundary costs nothing -- with one buffer every chunk ended in "all waves done reading -> write -> barrier ->
  // first fragment reads", a pipeline drain per 14 x 16 MFMAs (the timing-only build without the refill ran 37 %
  // shorter, and the loads themselves were only 8 % of that).  The split form's halo is 56 KB: one buffer, two workgroups.
  static constexpr int HBUF = 1;
  static constexpr size_t LDS = HBUF * (size_t)HB + NWB * (size_t)WPH;
};
static_assert(2 * HGeo<2>::LDS <= 160 * 1024, "LDS budget: two workgroups per CU");
static_assert(HCONV_NWB_BF != 3 || 3 * HGeo<2, true>::LDS <= 160 * 1024, "LDS budget: three bf16 workgroups per CU");



template <class F, class T, int... I>
__device__ __forceinline__ void hconv_static_for(F& f, T tag, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}, tag), ...);
}
// HALF-CHUNK TAIL (split form): when the last chunk holds 1 .. 8 channels (K = 129, 194, 258: a mask or an odd feature
// channel behind a multiple of 16; K = 5 and 4: conv0 and the input gradient of `last`), its MFMAs take K = 4 taps x 8
// channels instead of 2 taps x 16: 7 phases instead of 14, no zero channel half multiplied (K = 129: 5.6 % of the layer's MFMAs).
__host__ __device__ inline bool hconv_tail(int K, bool bf, bool itail = false) { return !bf && !itail && (K & 15) >= 1 && (K & 15) <= 8; }
// IM2COL TAIL (round 4, both forms): when the last chunk holds 1 .. 5 channels -- the mask channel behind 128 or 256 features
// (K = 129, 257), two odd channels (194, 258, 386, 514), `last`'s input gradient (4), conv0 (5) -- the K = 32 of one MFMA are
// the 27 TAPS of ONE of these channels (+ 5 zeros) instead of 2 taps x 16 channels: one MFMA group per tail channel instead of
// 7 (half-chunk tail) or 14 (bf16 had no tail form at all: K = 129 cost 9 x 14 = 126 groups for 113 of work).  The weights
// of such a group are packed as [row][tap]; the B operand -- 8 taps of one channel at the lane's voxel -- is gathered from the
// halo with 2-byte LDS reads at the 8 tap offsets of the lane's K group (a 32-entry table in LDS): 32 (split: 64) small reads
// per group against the 6 .. 13 groups of 8 (16) ds_read_b128 + 16 (48) MFMAs it replaces.
constexpr int kITailMax = 5;
// (split form: only for 1 - 2 tail channels -- against the half-chunk tail's 7 groups the gather pays up to there: K = 129 forward
//  19.5 -> 19.15 ms, K = 5 (conv0) 2.65 -> 2.85 ms, profiles/r04az_ab_hconv_im2col_tail.log; bf16, against 14 groups, gains at 1 .. 5)
inline bool hconv_itail_host(int K, bool bf) {
  const int nt = K & 15;
  return nt >= 1 && nt <= (bf ? kITailMax : 2) && getenv("SR3D_HCONV_NO_ITAIL") == nullptr;
}

// s_waitcnt vmcnt(n) lgkmcnt(0) for the counts the kernel uses (n folds to a constant in the unrolled phase loop)
__device__ __forceinline__ void hconv_wait_vm(const int n) {
#define SR3D_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ") lgkmcnt(0)" ::: "memory"); break;
  switch (n) {
    SR3D_W(0) SR3D_W(1) SR3D_W(2) SR3D_W(3) SR3D_W(4) SR3D_W(6)
    SR3D_W(16) SR3D_W(17) SR3D_W(18) SR3D_W(20)
    SR3D_W(32) SR3D_W(33) SR3D_W(34) SR3D_W(35) SR3D_W(36) SR3D_W(38)
    SR3D_W(56) SR3D_W(57) SR3D_W(58) SR3D_W(59) SR3D_W(60) SR3D_W(62)
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
  }
#undef SR3D_W
}

// PAIR = wide halo loads.
// bf16 (even X): the halo is fetched as 4-byte PAIRS of x-neighbours (32 instead of 56 loads per lane and chunk).
// The bf16 form has a third of the MFMAs per chunk, and with one 2-byte load per voxel and channel the vector-memory
// instruction rate (~22 cycles per wave instruction, 8 waves per CU) became its limit: a timing-only build WITHOUT the halo
// refill ran 37 % shorter (profiles/r03d_ablation_hconv_kernel_bf16.log).
// fp32 (X % 4 == 0): 16-byte QUADS of x-neighbours, 16 instead of 56 loads per lane and chunk: a lane takes quad q of a
// halo row (x = x0 - 4 + 4 q .. + 3; the outer quads contribute one column each) for its 8 channels and holds 4 voxels x 8
// channels = four 16-byte pieces per part, no cross-lane exchange.  The timing-only build of the split form without the
// raw-row loads (everything else kept) ran 20 % shorter, and more prefetch distance did not help: the loads cost
// instruction slots of the texture path, not latency (profiles/r03e_ablation_hconv_kernel_fp32.log).
// WIDE = 2 (bf16, X % 4 == 0, 8-byte aligned tensors; round 4): 8-byte QUADS of x-neighbours, 16 instead of 32 loads per lane and
// chunk with the task geometry of the fp32 quads: a lane holds 4 voxels of its 8 channels as 8 x 2 dwords and builds the four
// 16-byte pieces (voxel, 8 channels) with the same 32 v_perm per chunk as the pair form -- the refill costs vector-memory
// INSTRUCTIONS (the timing-only build without it ran 37 % shorter), so half of them is the lever.
template <int RT, bool BF, int WIDE>
__global__ __launch_bounds__(HNT, (BF && HCONV_NWB_BF == 3) ? 3 : 2) void hconv_kernel(const SrHconvParams p) {
  using G = HGeo<RT, BF>;
  constexpr bool PAIR = WIDE != 0;
  constexpr bool BQUAD = BF && WIDE == 2;
  constexpr bool BPAIR = BF && WIDE == 1, QUAD = !BF && PAIR;
  static_assert(BF || WIDE <= 1, "fp32: WIDE = 1 are the 16-byte quads");
  constexpr int NR = BPAIR ? 4 : (QUAD || BQUAD) ? 2 : HNR;   // staging rounds
  constexpr int RV = QUAD ? 4 : 1;             // voxels per lane and round and channel in `raw`
  constexpr int HPX = 18;                      // bf16 pairs per halo row, covering hx = -1 .. 34
  constexpr int HQX = 10;                      // fp32 quads per halo row, covering hx = -3 .. 36
  constexpr int NP = G::NP;
  constexpr int ESZ = BF ? 2 : 4;              // bytes per activation element
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Hs = lds;
  unsigned char* Ws = lds + G::HBUF * G::HB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < HNT / 64);

  // workgroup -> (row block, voxel tile); the row blocks of one tile run next to each other on one XCD
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
  const int z0 = tiz * 2, y0 = tiy * 4, x0 = tix * 32;
  const long long ZYX = (long long)p.Z * p.Y * p.X;
  const int chan_bytes = (int)(ZYX * ESZ);

  int sw = BF ? 0 : split_scale_exp(*p.absmax_w);
  if (sw == kSplitScaleNone) sw = 0;
  // exchange slots for the wave maxima [chunk parity][wave]: in the padding behind the 816 voxels of halo plane 0
  float* xmax = reinterpret_cast<float*>(Hs + HVOX * 16);

  // ---- staging geometry: this wave stages channel half `sh` of every chunk, voxel blocks r * 2 + (wave >> 1)
  const int sh = wave & 1;
  unsigned soff[NR];    // byte offset inside a channel volume, 0xffffffff = zero padding
  int swr[NR];          // byte offset of the 16-byte piece inside a halo plane (PAIR: of the pair's first voxel; < 0: none)
  int swr2[NR];         // bf16 PAIR: of the pair's second voxel (< 0: none); fp32 QUAD: 4-bit mask of the quad's voxels inside the halo
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int e = (r * 2 + (wave >> 1)) * 64 + lane;
    if constexpr (QUAD || BQUAD) {   // quad q of row (hz, hy): x = x0 - 4 + 4 q .. + 3  <->  halo columns hx = 4 q - 3 .. 4 q
      const int row = e / HQX, q = e - row * HQX;
      const int hz = row / HHY, hy = row - hz * HHY;
      const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 4 + 4 * q;
      const bool inrow = e < HHZ * HHY * HQX;
      const bool ok = inrow && (unsigned)gz < (unsigned)p.Z && (unsigned)gy < (unsigned)p.Y && (unsigned)gx < (unsigned)p.X;   // X % 4 == 0: all four or none
      soff[r] = ok ? (unsigned)((gz * p.Y + gy) * p.X + gx) * (unsigned)ESZ : 0xffffffffu;
      swr[r] = ((hz * HHY + hy) * HHX + 4 * q - 3) * 16;   // (of the quad's first voxel; may lie outside the row)
      swr2[r] = !inrow ? 0 : q == 0 ? 8 : q == HQX - 1 ? 1 : 15;
    } else if constexpr (BPAIR) {   // pair pp of row (hz, hy): x = x0 - 2 + 2 pp, + 1  <->  halo columns hx = 2 pp - 1, 2 pp
      const int hz = e / (HHY * HPX), r2 = e - hz * (HHY * HPX);
      const int hy = r2 / HPX, pp = r2 - hy * HPX;
      const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 2 + 2 * pp;
      const bool row = e < HHZ * HHY * HPX;
      const bool ok = row && (unsigned)gz < (unsigned)p.Z && (unsigned)gy < (unsigned)p.Y && (unsigned)gx < (unsigned)p.X;   // X even: both or none
      soff[r] = ok ? (unsigned)((gz * p.Y + gy) * p.X + gx) * 2u : 0xffffffffu;
      const int v0 = (hz * HHY + hy) * HHX + 2 * pp - 1;
      swr[r] = (row && pp > 0) ? v0 * 16 : -1;
      swr2[r] = (row && pp < HPX - 1) ? (v0 + 1) * 16 : -1;
    } else {
      const int hz = e / (HHY * HHX), r2 = e - hz * (HHY * HHX);
      const int hy = r2 / HHX, hx = r2 - hy * HHX;
      const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = e < HVOX && (unsigned)gz < (unsigned)p.Z && (unsigned)gy < (unsigned)p.Y && (unsigned)gx < (unsigned)p.X;
      soff[r] = ok ? (unsigned)((gz * p.Y + gy) * p.X + gx) * (unsigned)ESZ : 0xffffffffu;
      swr[r] = e * 16;
      swr2[r] = -1;
    }
  }
  // Per-slice base pointers of this sample, pinned in scalar registers: left to itself hipcc turns the slice selects into
  // dependent kernel-argument loads (two or three ~200-cycle scalar-load round trips per channel, in every wave, at the
  // start of every chunk).
  // (separate variables, not arrays: an array would be indexed in scratch memory)
#define SR3D_SLICE_BASE(i) (reinterpret_cast<unsigned long long>(p.in.ptr[i]) + (unsigned long long)((long long)b * p.in.bstride[i]) * ESZ)
  unsigned long long sb0 = SR3D_SLICE_BASE(0), sb1 = SR3D_SLICE_BASE(1), sb2 = SR3D_SLICE_BASE(2), sb3 = SR3D_SLICE_BASE(3);
#undef SR3D_SLICE_BASE
  int cb0 = p.in.cbeg[0], cb1 = p.in.cbeg[1], cb2 = p.in.cbeg[2], cb3 = p.in.cbeg[3];
  split_pin_scalar(sb0), split_pin_scalar(sb1), split_pin_scalar(sb2), split_pin_scalar(sb3);
  split_pin_scalar(cb0), split_pin_scalar(cb1), split_pin_scalar(cb2), split_pin_scalar(cb3);
  auto slice_of = [&](const int gc) { return (gc >= cb1) + (gc >= cb2) + (gc >= cb3); };
  // (mask arithmetic instead of selects: hipcc turns a select chain over four values into a lookup table in scratch)
  unsigned long long dsb1 = sb1 - sb0, dsb2 = sb2 - sb1, dsb3 = sb3 - sb2;
  int dcb1 = cb1 - cb0, dcb2 = cb2 - cb1, dcb3 = cb3 - cb2;
  split_pin_scalar(dsb1), split_pin_scalar(dsb2), split_pin_scalar(dsb3), split_pin_scalar(dcb1), split_pin_scalar(dcb2), split_pin_scalar(dcb3);
  auto chan_base = [&](const int gc) {   // gc is wave-uniform
    const long long m1 = -(long long)(gc >= cb1), m2 = -(long long)(gc >= cb2), m3 = -(long long)(gc >= cb3);
    const unsigned long long base = sb0 + (dsb1 & (unsigned long long)m1) + (dsb2 & (unsigned long long)m2) + (dsb3 & (unsigned long long)m3);
    const int c0 = cb0 + (dcb1 & (int)m1) + (dcb2 & (int)m2) + (dcb3 & (int)m3);
    return base + (unsigned long long)(unsigned)(gc - c0) * (unsigned long long)(unsigned)chan_bytes;
  };
  float raw[NR][BQUAD ? 16 : 8 * RV];   // [round][channel][voxel of the quad]; bf16 quads: [round][channel][pair of the quad]
  // QUAD: channel c of this wave's 8, both rounds (two 16-byte loads per lane)
  auto load_raw_q = [&](const int chunk, const int c) {
    if constexpr (QUAD) {
      const int gc = chunk * HKC + sh * 8 + c;   // wave-uniform
      const unsigned long long base = chan_base(gc < p.K ? gc : p.K - 1);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, gc < p.K ? chan_bytes : 0, 0x00020000);
#pragma unroll
      for (int r = 0; r < NR; r++) {
        if constexpr ((HCONV_ABL & 64) != 0) {
#pragma unroll
          for (int v = 0; v < 4; v++) asm volatile("" : "=v"(raw[r][c * 4 + v]));
          continue;
        }
        const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, soff[r], 0, 0);
#pragma unroll
        for (int v = 0; v < 4; v++) raw[r][c * 4 + v] = __builtin_bit_cast(float, (unsigned)t[v]);
      }
    }
  };
  auto load_raw = [&](const int chunk) {
    if constexpr (QUAD) {
#pragma unroll
      for (int c = 0; c < 8; c++) load_raw_q(chunk, c);
      return;
    }
    if constexpr (BQUAD) {   // 8 channels x NR rounds of 8-byte loads (4 bf16 x-neighbours)
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const int gc = chunk * HKC + sh * 8 + c;   // wave-uniform
        const unsigned long long base = chan_base(gc < p.K ? gc : p.K - 1);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, gc < p.K ? chan_bytes : 0, 0x00020000);
#pragma unroll
        for (int r = 0; r < NR; r++) {
          if constexpr ((HCONV_ABL & 64) != 0) {
            asm volatile("" : "=v"(raw[r][2 * c]), "=v"(raw[r][2 * c + 1]));
            continue;
          }
          const auto t = __builtin_amdgcn_raw_buffer_load_b64(rs, soff[r], 0, 0);
          raw[r][2 * c] = __builtin_bit_cast(float, (unsigned)t[0]);
          raw[r][2 * c + 1] = __builtin_bit_cast(float, (unsigned)t[1]);
        }
      }
      return;
    }
    const int gc0 = chunk * HKC + sh * 8;      // wave-uniform
    const int first = gc0 < p.K ? gc0 : p.K - 1, last = gc0 + 7 < p.K ? gc0 + 7 : p.K - 1;
    unsigned long long cbase[8];
    if (slice_of(first) == slice_of(last)) {   // the usual case: 8 consecutive channels of one tensor
      const unsigned long long b0 = chan_base(first);
#pragma unroll
      for (int c = 0; c < 8; c++) cbase[c] = b0 + (unsigned long long)c * (unsigned)chan_bytes;
    } else {
#pragma unroll
      for (int c = 0; c < 8; c++) cbase[c] = chan_base(gc0 + c < p.K ? gc0 + c : p.K - 1);
    }
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)cbase[c], 0, gc0 + c < p.K ? chan_bytes : 0, 0x00020000);
#pragma unroll
      for (int r = 0; r < NR; r++) {
        if constexpr ((HCONV_ABL & 64) != 0) {   // ablation: an opaque definition, no load
#pragma unroll
          for (int v = 0; v < RV; v++) asm volatile("" : "=v"(raw[r][c * RV + v]));
          continue;
        }
        if constexpr (BF && !PAIR)   // the 16 bits of the bf16 element, zero-extended
          raw[r][c] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, soff[r], 0, 0));
        else                         // fp32 element, or two bf16 x-neighbours
          raw[r][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, soff[r], 0, 0));
      }
    }
  };
  // largest |value| among this wave's rows of the chunk in `raw` -> its exchange slot
  // per-slice maxima of |x| over this workgroup's halo tiles: exported at the end for the split-f16 weight gradient of the
  // same layer (sr3d_hwgrad.hip needs max |x| per slice; this kernel sees every element anyway).  Only the workgroups of
  // the first row block export.
  const bool export_max = !BF && p.amax_out != nullptr && nblk == 0;
  float rmax0 = 0.f, rmax1 = 0.f, rmax2 = 0.f, rmax3 = 0.f;
  auto publish_max = [&](const int parity, const int ck) {
    if constexpr (BF) return;   // no scaling: bf16 has fp32's exponent range
    if constexpr ((HCONV_ABL & 128) != 0) {   // ablation: the loads are waited for and consumed, no arithmetic
#pragma unroll
      for (int r = 0; r < NR; r++)
#pragma unroll
        for (int c = 0; c < 8 * RV; c++) asm volatile("" ::"v"(raw[r][c]));
      return;
    }
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < NR; r++)   // (QUAD: the outer quads bring up to three columns from beyond the halo: elements of the
#pragma unroll                     //  same tensor, so the tile scale is only more cautious and the exported maxima stay exact)
      for (int c = 0; c < 8 * RV; c += 2) m = fmaxf(fmaxf(m, fabsf(raw[r][c])), fabsf(raw[r][c + 1]));
    m = split_wave_max(m);
    if (lane == 0) xmax[parity * 4 + wave] = m;
    if (export_max) {
      const int gc0 = ck * HKC + sh * 8;
      const int sa = slice_of(gc0 < p.K ? gc0 : p.K - 1), sb = slice_of(gc0 + 7 < p.K ? gc0 + 7 : p.K - 1);
      auto credit = [&](const int sl, const float mv) {
        rmax0 = sl == 0 ? fmaxf(rmax0, mv) : rmax0;
        rmax1 = sl == 1 ? fmaxf(rmax1, mv) : rmax1;
        rmax2 = sl == 2 ? fmaxf(rmax2, mv) : rmax2;
        rmax3 = sl == 3 ? fmaxf(rmax3, mv) : rmax3;
      };
      if (sa == sb) {
        credit(sa, m);
      } else {   // the 8 channels straddle a slice boundary (rare: once or twice per layer): one maximum per channel
#pragma unroll
        for (int c = 0; c < 8; c++) {
          float mc = 0.f;
#pragma unroll
          for (int r = 0; r < NR; r++)
#pragma unroll
            for (int v = 0; v < RV; v++) mc = fmaxf(mc, fabsf(raw[r][c * RV + v]));
          mc = split_wave_max(mc);
          credit(slice_of(gc0 + c < p.K ? gc0 + c : p.K - 1), mc);
        }
      }
    }
  };
  // running scale exponent: the largest chunk magnitude seen so far decides (kSplitScaleNone until a non-zero chunk came)
  auto next_scale = [&](const int parity, const int s_run) {
    if constexpr (BF) return 0;
    const float m = fmaxf(fmaxf(xmax[parity * 4 + 0], xmax[parity * 4 + 1]), fmaxf(xmax[parity * 4 + 2], xmax[parity * 4 + 3]));
    const int s_c = __builtin_amdgcn_readfirstlane(split_scale_exp(m));
    return s_c < s_run ? s_c : s_run;
  };
  constexpr int NPC = (QUAD || BQUAD) ? NR * 4 : NR;   // 16-byte pieces a lane builds per chunk and part
  h8 chi[NPC], clo[BQUAD ? 1 : NPC];   // halo pieces of the next chunk, split (bf16 PAIR: the two voxels of the pair; QUAD / bf16 quads: [round][voxel])
  // QUAD: one voxel of one round: 8 channels scaled and split = 24 vector instructions, placed next to 24 MFMAs
  auto convert_sub = [&](const int r, const int v, const float in_mult) {
    if constexpr (QUAD) {
      if constexpr ((HCONV_ABL & 128) != 0) {
        asm volatile("" : "=v"(chi[r * 4 + v]), "=v"(clo[r * 4 + v]));
        return;
      }
      // (split_piece: two v_fma_mix per element, no packed fp32 beside the MFMAs)
      split_piece(&raw[r][v], 4, in_mult, chi[r * 4 + v], clo[r * 4 + v]);
    }
  };
  auto convert = [&](const float in_mult) {
    if constexpr ((HCONV_ABL & 128) != 0) {
#pragma unroll
      for (int r = 0; r < (BQUAD ? NPC : NR); r++) asm volatile("" : "=v"(chi[r]), "=v"(clo[BQUAD ? 0 : r]));
      return;
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
      if constexpr (QUAD) {   // (prologue only: the chunk loop spreads convert_sub over four phases)
#pragma unroll
        for (int v = 0; v < 4; v++) {
          convert_sub(r, v, in_mult);
          asm volatile("" : "+v"(chi[r * 4 + v]), "+v"(clo[r * 4 + v]));
          __builtin_amdgcn_sched_barrier(0);
        }
        continue;
      }
      if constexpr (BQUAD) {   // dwords (2c, 2c + 1) = voxels (0, 1), (2, 3) of channel c  ->  four 16-byte pieces of 8 channels
#pragma unroll
        for (int v = 0; v < 4; v++) {
          u32x4 pc;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const unsigned a = __builtin_bit_cast(unsigned, raw[r][2 * (2 * k) + (v >> 1)]);
            const unsigned bq = __builtin_bit_cast(unsigned, raw[r][2 * (2 * k + 1) + (v >> 1)]);
            pc[k] = __builtin_amdgcn_perm(bq, a, (v & 1) ? 0x07060302u : 0x05040100u);   // channels 2k, 2k + 1 at voxel v
          }
          chi[r * 4 + v] = __builtin_bit_cast(h8, pc);
          asm volatile("" : "+v"(chi[r * 4 + v]));
        }
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      if constexpr (BPAIR) {   // dword c = (voxel 0, voxel 1) of channel c  ->  two 16-byte pieces of 8 channels
        u32x4 p0, p1;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const unsigned a = __builtin_bit_cast(unsigned, raw[r][2 * c]), bq = __builtin_bit_cast(unsigned, raw[r][2 * c + 1]);
          p0[c] = __builtin_amdgcn_perm(bq, a, 0x05040100u);   // low halves: voxel 0 of channels 2c, 2c + 1
          p1[c] = __builtin_amdgcn_perm(bq, a, 0x07060302u);   // high halves: voxel 1
        }
        chi[r] = __builtin_bit_cast(h8, p0);
        clo[r] = __builtin_bit_cast(h8, p1);
        asm volatile("" : "+v"(chi[r]), "+v"(clo[r]));
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      if constexpr (BF) {   // pack the 8 channels of a voxel: 16 bytes, the MFMA operand as it is
        u32x4 pk;
#pragma unroll
        for (int c = 0; c < 4; c++)
          pk[c] = __builtin_bit_cast(unsigned, raw[r][2 * c]) | (__builtin_bit_cast(unsigned, raw[r][2 * c + 1]) << 16);
        chi[r] = __builtin_bit_cast(h8, pk);
        asm volatile("" : "+v"(chi[r]));
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      split_piece(&raw[r][0], 1, in_mult, chi[r], clo[r]);
      asm volatile("" : "+v"(chi[r]), "+v"(clo[r]));   // here, between the MFMAs -- not sunk to the stores behind the barrier
      __builtin_amdgcn_sched_barrier(0);               // one round at a time: its temporaries die before the next starts
    }
  };
  auto write_halo = [&](const int buf) {   // buf: halo buffer (bf16: chunk parity; split form: 0)
    unsigned char* Hw = Hs + buf * G::HB;
#pragma unroll
    for (int r = 0; r < NR; r++) {
      if constexpr (QUAD) {
#pragma unroll
        for (int v = 0; v < 4; v++)
          if ((swr2[r] >> v) & 1) {
            *reinterpret_cast<h8*>(Hw + (0 * 2 + sh) * HPLANE + swr[r] + v * 16) = chi[r * 4 + v];
            *reinterpret_cast<h8*>(Hw + (1 * 2 + sh) * HPLANE + swr[r] + v * 16) = clo[r * 4 + v];
          }
        continue;
      }
      if constexpr (BQUAD) {
#pragma unroll
        for (int v = 0; v < 4; v++)
          if ((swr2[r] >> v) & 1) *reinterpret_cast<h8*>(Hw + sh * HPLANE + swr[r] + v * 16) = chi[r * 4 + v];
        continue;
      }
      if constexpr (BPAIR) {
        if (swr[r] >= 0) *reinterpret_cast<h8*>(Hw + sh * HPLANE + swr[r]) = chi[r];
        if (swr2[r] >= 0) *reinterpret_cast<h8*>(Hw + sh * HPLANE + swr2[r]) = clo[r];
        continue;
      }
      if (r == HNR - 1 && swr[r] >= HVOX * 16) continue;   // padding voxels: the exchange slots live there
      *reinterpret_cast<h8*>(Hw + (0 * 2 + sh) * HPLANE + swr[r]) = chi[r];
      if constexpr (!BF) *reinterpret_cast<h8*>(Hs + (1 * 2 + sh) * HPLANE + swr[r]) = clo[r];
    }
  };
  // one kz phase of the packed weights: PIECES contiguous 1 KB fragments, LDS-DMA.  (The buffer form on purpose: the
  // global_load_lds form is a FLAT instruction that touches two address spaces, and while one is pending hipcc turns
  // every LDS wait into lgkmcnt(0) -- the register double-buffering of the fragments below would wait for the reads
  // it has just issued.)
  const unsigned char* wblock = reinterpret_cast<const unsigned char*>(p.wimg) + (size_t)(p.nb_off + nblk) * p.nchunks * HPH * G::WPHASE;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wblock, 0, p.nchunks * HPH * G::WPHASE, 0x00020000);
  auto dma_w = [&](const int phase, unsigned char* W) {   // phase = chunk * NPH + index of its first tap pair / PP
#pragma unroll
    for (int ii = 0; ii < (G::PP * G::PIECES + 3) / 4; ii++) {
      const int i = wave + 4 * ii;
      if (i < G::PP * G::PIECES) split_lds_dma16(wrs, (lds_p)(W + i * 1024), phase * G::WPH + i * 1024 + lane * 16);
    }
  };

  // v_mfma_f32_16x16x32_f16 (14 % more power-efficient than 32x32x16 on this part, tools/mfma_rate.hip): K = 32 = the 16
  // channels of the chunk at TWO taps.  Lane l of an operand: row / voxel l & 15, K group l >> 4 = 2 * (tap of the pair)
  // + (channel half).  The wave's tile is 32 * RT rows x 2 voxel rows x 32 x = (2 RT) x 4 tiles of 16 x 16.
  constexpr int NRT = 2 * RT;
  f32x4 acc[NRT][4];
#pragma unroll
  for (int i = 0; i < NRT; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // B fragment bases: voxel tile j = (voxel row 2 * wave + (j >> 1), x half j & 1)
  int bbase[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int vt = 2 * wave + (j >> 1);
    bbase[j] = ((lane >> 4) & 1) * HPLANE + (((vt >> 2) * HHY + (vt & 3)) * HHX + (j & 1) * 16 + (lane & 15)) * 16;
  }
  const bool tap_b = lane >= 32;   // this lane supplies the second tap of a pair
  const int abase = lane * 16;

  // ---- prologue
  constexpr int NWB = G::NWB, AHEAD = NWB - 1;
  constexpr int PP = G::PP, NPH = G::NPH;
  const bool itail = p.itail != 0;
  const bool tail = hconv_tail(p.K, BF, itail);
  const int nt = p.K & 15;                                  // channels of the last chunk (itail: 1 .. 5)
  const int nph_it = (nt + PP - 1) / PP;                    // its phases in the im2col form: PP channels each
  const int nphases = itail ? (p.nchunks - 1) * NPH + nph_it : p.nchunks * NPH - (tail ? NPH / 2 : 0);
  // tap offsets for the im2col tail's gather: behind the exchange slots in the padding of halo plane 0
  int* lut = reinterpret_cast<int*>(Hs + HVOX * 16 + 64);
  if (itail && tid < 32) {   // (here, in front of the prologue's barriers: a one-chunk layer -- conv0 -- has no other before its tail)
    const int u = tid > 26 ? 26 : tid;   // taps 27 .. 31 read where tap 26 does (their weights are zero)
    lut[tid] = (((u / 9) * HHY + (u / 3) % 3) * HHX + u % 3) * 16;
  }
#pragma unroll
  for (int a = 0; a < AHEAD; a++)
    if (a < nphases) dma_w(a, Ws + a * G::WPH);
  load_raw(0);
  publish_max(0, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int s_run = next_scale(0, kSplitScaleNone);   // exponent of the scale the accumulators are in
  convert(ldexpf(1.f, s_run));
  write_halo(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int s_next = s_run;
  float in_mult_next = 1.f;

  // halo byte offset of tap t = (kz, ky, kx); the dummy tap 27 reads where tap 26 does (its weights are zero)
  auto tap_off = [](const int t) { const int u = t > 26 ? 26 : t; return (((u / 9) * HHY + (u / 3) % 3) * HHX + u % 3) * 16; };
  int phase = 0;
  int wb = 0;            // weight buffer of the current phase = phase % NWB (running: NWB = 3 is not a power of two)
  auto wb_plus = [](const int w, const int a) { const int t = w + a; return t >= NWB ? t - NWB : t; };
  int chunk = 0;
  const int nfull = p.nchunks - ((tail || itail) ? 1 : 0);   // full 16-channel chunks; the tail follows the loop
  // (the phases as instantiations of one generic lambda: kzy is a compile-time constant in each -- a `#pragma unroll` loop
  // is only a request, and rolled the next chunk's raw rows and their split form would both be live in every phase)
  auto phase_body = [&](auto kz, auto tl) {
      constexpr int kzy = decltype(kz)::value;
      constexpr bool TAIL = decltype(tl)::value;   // the half-chunk tail: phase kzy holds taps 4 kzy .. + 3 of channels 0..7
      const unsigned char* W0 = Ws + wb * G::WPH + abase;
      // Phase 1: this wave's share of the next chunk has landed (issued in phase 0): publish its largest magnitude.
      // Phase 2 (behind the barrier of phase 1): all four maxima -> scale of the next chunk; split its rows.  Both before
      // this phase's DMA is issued: hipcc does not see the hand-written waits and guards the first use of `raw` with its
      // own vmcnt(0).
      // (bf16: nothing to publish or scale; the rows are packed in the LAST phase, so that the raw registers and their
      // packed form are never live together and the loads have the whole chunk to land)
      // QUAD: the raw rows of the next chunk come one channel (two 16-byte loads) per phase in phases 0..7 -- issued in one
      // burst, the 8 waves' loads stood in the texture path in front of the following phases' weight DMAs --, the maxima are
      // published at the end of phase 8, and phases 9..12 scale and split two voxels each NEXT to their MFMAs (one vector
      // instruction per MFMA, in its shadow) instead of in a block of ~200 in front of them.
      constexpr int QLOAD = 8, QPUB = 8, QCONV = 9;
      if (!(HCONV_ABL & 1) && !TAIL && QUAD && kzy == QCONV) {
        s_next = next_scale((chunk + 1) & 1, s_run);
        in_mult_next = ldexpf(1.f, s_next);
      }
      if (!(HCONV_ABL & 1) && !TAIL && !QUAD && kzy == AHEAD + 1) {
        s_next = next_scale((chunk + 1) & 1, s_run);
        convert(ldexpf(1.f, s_next));
        // bf16: straight into the OTHER halo buffer (nobody has read it since the previous chunk; the per-phase barriers
        // that follow publish it long before the next chunk's first fragment read)
        if constexpr (BF && G::HBUF == 2) write_halo((chunk + 1) & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(HCONV_ABL & 2) && phase + AHEAD < nphases) dma_w(phase + AHEAD, Ws + wb_plus(wb, AHEAD) * G::WPH);
      __builtin_amdgcn_sched_barrier(0);    // (the wait below counts on this order)
      if (!(HCONV_ABL & 1) && !TAIL && !QUAD && kzy == 0) load_raw(chunk + 1);    // (past the end: empty descriptors, zeros)
      if (!(HCONV_ABL & 1) && !TAIL && QUAD && kzy < QLOAD) load_raw_q(chunk + 1, kzy);
#pragma unroll
      for (int sub = 0; sub < PP; sub++) {
      const int pair = kzy * PP + sub;
      const unsigned char* W = W0 + sub * G::WPHASE;
      // this lane's halo offset for the pair: first tap on lanes 0..31, second on 32..63
      const unsigned char* Hk;
      if constexpr (TAIL) {   // lane group g = lane >> 4 supplies tap 4 kzy + g, always of channel half 0 (bbase carries (g & 1) halves)
        const int g = lane >> 4;
        const int o = g == 0 ? tap_off(4 * kzy) : g == 1 ? tap_off(4 * kzy + 1) : g == 2 ? tap_off(4 * kzy + 2) : tap_off(4 * kzy + 3);
        Hk = Hs + o - (g & 1) * HPLANE;
      } else {
        Hk = Hs + (G::HBUF == 2 ? (chunk & 1) * G::HB : 0) + (tap_b ? tap_off(2 * pair + 1) : tap_off(2 * pair));
      }
      h8 fb[NP][4], fa[NP][2];   // [hi | lo][voxel tile]; [hi | lo][row tile of the current half]
#pragma unroll
      for (int part = 0; part < NP; part++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (!(HCONV_ABL & 16) || pair == 0) fb[part][j] = *reinterpret_cast<const h8*>(Hk + part * (2 * HPLANE) + bbase[j]);
      // row tiles two at a time (12 LDS reads in flight at most: 16 overflow the lgkmcnt counter model)
#pragma unroll
      for (int ih = 0; ih < NRT; ih += 2) {
#pragma unroll
        for (int part = 0; part < NP; part++)
#pragma unroll
          for (int i = 0; i < 2; i++)
            if (!(HCONV_ABL & 8) || (pair == 0 && ih == 0)) fa[part][i] = *reinterpret_cast<const h8*>(W + (part * NRT + ih + i) * 1024);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (QUAD && !TAIL && !(HCONV_ABL & 1)) {
          if (kzy >= QCONV && kzy < QCONV + 4) {   // 8 (round, voxel) units over 4 phases: one per half (RT = 2), two (RT = 1)
            constexpr int UPH = 2 / (NRT / 2);     // units per half
#pragma unroll
            for (int u = 0; u < UPH; u++) {
              const int unit = (kzy - QCONV) * 2 + (ih / 2) * UPH + u;
              convert_sub(unit >> 2, unit & 3, in_mult_next);
            }
#pragma unroll
            for (int g = 0; g < 24; g++) {         // MFMA, vector op, MFMA, vector op, ...
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, UPH, 0);
            }
          }
        }
        if constexpr (HCONV_SETPRIO != 0) __builtin_amdgcn_s_setprio(1);   // (the MFMA burst ahead of the other workgroup's vector work)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if constexpr (BF) {
              acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, fa[0][i]), __builtin_bit_cast(bf8, fb[0][j]),
                                                                       acc[ih + i][j], 0, 0, 0);
            } else {
              if (!(HCONV_ABL & 32)) {
                acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[NP - 1][j], acc[ih + i][j], 0, 0, 0);
                acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[NP - 1][i], fb[0][j], acc[ih + i][j], 0, 0, 0);
              }
              acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[0][j], acc[ih + i][j], 0, 0, 0);
            }
          }
        if constexpr (QUAD && !TAIL && !(HCONV_ABL & 1)) {   // the split pieces exist HERE (hipcc would sink the arithmetic to the stores behind the chunk)
          if (kzy >= QCONV && kzy < QCONV + 4) {
            constexpr int UPH = 2 / (NRT / 2);
#pragma unroll
            for (int u = 0; u < UPH; u++) {
              const int unit = (kzy - QCONV) * 2 + (ih / 2) * UPH + u;
              asm volatile("" : "+v"(chi[unit]), "+v"(clo[unit]));
            }
          }
        }
        if constexpr (HCONV_SETPRIO != 0) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
      }   // sub
      // The NEXT phase's weights have landed (vector-memory operations complete in issue order).  Younger than their DMA
      // and allowed to stay in flight: the DMAs of the AHEAD - 1 phases after it (ND instructions per wave and phase: 64-row
      // blocks of the split form 8 pieces over 4 waves, ...), and -- in the first AHEAD phases of a chunk -- the 8 * NR
      // raw-row loads of the next chunk, which are issued right after the DMA of phase 0.  In phase AHEAD they have landed
      // (waited for together with the DMA issued behind them): this wave's share of the next chunk publishes its largest
      // magnitude before the barrier, and phase AHEAD + 1 picks the four maxima up, scales and splits.
      {
        constexpr int ND = (G::PP * G::PIECES + 3) / 4, NRAW = 8 * NR;
        static_assert(G::PP * G::PIECES % 4 == 0 || G::PP * G::PIECES < 4, "every wave issues the same number of DMAs");
        if (!(HCONV_ABL & 1) && !TAIL && !BF && !QUAD && kzy == AHEAD) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND * (AHEAD - 1)) : "memory");   // (the compiler's own wait would be vmcnt(0))
          publish_max((chunk + 1) & 1, chunk + 1);
        }
        if (!(HCONV_ABL & 1) && !TAIL && QUAD && kzy == QPUB) publish_max((chunk + 1) & 1, chunk + 1);   // (no load was issued in this phase)
        if (phase + AHEAD >= nphases)              // the last phases: nothing new was issued
          hconv_wait_vm(0);
        else if constexpr (TAIL)                   // (the last chunk: no rows are fetched)
          hconv_wait_vm(ND * (AHEAD - 1));
        else if constexpr (QUAD)                   // behind the DMA: this phase's two loads
          hconv_wait_vm(ND * (AHEAD - 1) + (kzy < QLOAD ? NR : 0));
        else
          hconv_wait_vm(ND * (AHEAD - 1) + (kzy < AHEAD ? NRAW : 0));
      }
      if (!(HCONV_ABL & 4)) __builtin_amdgcn_s_barrier();
      wb = wb_plus(wb, 1);
      phase++;
    };
  for (; chunk < nfull; chunk++) {
    hconv_static_for(phase_body, std::false_type{}, std::make_integer_sequence<int, NPH>{});
    if constexpr (G::HBUF == 2) continue;   // (double-buffered halo: written during the chunk)
    if (!(HCONV_ABL & 1) && chunk + 1 < p.nchunks) {
      // The f16 MFMA truncates inside its adder tree: every accumulation step leaves a tiny NEGATIVE error whatever the
      // sign of the sum (measured: mean error -5e-7 of the output rms at K = 1032, against 3e-10 for the fp32 MFMA; the
      // normwise error is the same).  A bias adds up coherently in sums over a million voxels (bias gradients were
      // 5e-5 off).  So the packed weights alternate sign from chunk to chunk and the accumulators are negated in
      // between: the result is unchanged and the truncation errors of successive chunks cancel.  (The same multiply
      // moves the accumulators to the next chunk's scale when that chunk is larger than everything before it.)
      // (every 2^HFLIP_SH = 2 chunks: negating 64 accumulator registers is 32 packed multiplies, ~700 cycles of vector issue per
      //  chunk; the errors of pairs of chunks cancel like those of single chunks: mean error 1e-8 of the rms against -5e-7 without,
      //  tools/hconv_check.py deep -- the bias test of
      //  tests/test_gpu_split_f16.py holds -- and the multiply is skipped when there is neither a flip nor a rescale)
      const bool turn = (((chunk + 1) >> HFLIP_SH) ^ (chunk >> HFLIP_SH)) & 1;
      if (!BF && !(HCONV_ABL & 256) && (turn || s_next != s_run)) {   // (bf16: the weights keep their sign, nothing to rescale)
        const float flip = ldexpf(turn ? -1.f : 1.f, s_next - s_run);
#pragma unroll
        for (int i = 0; i < NRT; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i][j] *= flip;
      }
      s_run = s_next;
      if (!(HCONV_ABL & 512)) {
        write_halo(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  if constexpr (!BF) {   // the half-chunk tail (the refill state of the loop above is dead here)
    if (tail) hconv_static_for(phase_body, std::true_type{}, std::make_integer_sequence<int, NPH / 2>{});
  }
  if (itail) {   // the im2col tail: phase ph = tail channels ph * PP .. + PP - 1, one MFMA group (K = its 27 taps) each
    const int g = lane >> 4;
    int toff[8];
    {
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      const i32x4 a = *reinterpret_cast<const i32x4*>(lut + 8 * g), c4 = *reinterpret_cast<const i32x4*>(lut + 8 * g + 4);
      toff[0] = a[0], toff[1] = a[1], toff[2] = a[2], toff[3] = a[3], toff[4] = c4[0], toff[5] = c4[1], toff[6] = c4[2], toff[7] = c4[3];
    }
    for (int ph = 0; ph < nph_it; ph++) {
      const unsigned char* W0 = Ws + wb * G::WPH + abase;
      if (phase + AHEAD < nphases) dma_w(phase + AHEAD, Ws + wb_plus(wb, AHEAD) * G::WPH);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sub = 0; sub < PP; sub++) {
        const int sc = ph * PP + sub;   // tail channel (wave-uniform)
        if (sc >= nt) break;
        const unsigned char* W = W0 + sub * G::WPHASE;
        h8 fb[NP][4];
#pragma unroll
        for (int part = 0; part < NP; part++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const unsigned char* hb = Hs + part * (2 * HPLANE) + (bbase[j] - ((lane >> 4) & 1) * HPLANE) + sc * 2;   // channel half 0
            u32x4 pk;
#pragma unroll
            for (int k = 0; k < 4; k++)
              pk[k] = (unsigned)*reinterpret_cast<const unsigned short*>(hb + toff[2 * k]) |
                      ((unsigned)*reinterpret_cast<const unsigned short*>(hb + toff[2 * k + 1]) << 16);
            fb[part][j] = __builtin_bit_cast(h8, pk);
          }
#pragma unroll
        for (int ih = 0; ih < NRT; ih += 2) {
          h8 fa[NP][2];
#pragma unroll
          for (int part = 0; part < NP; part++)
#pragma unroll
            for (int i = 0; i < 2; i++) fa[part][i] = *reinterpret_cast<const h8*>(W + (part * NRT + ih + i) * 1024);
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if constexpr (BF) {
                acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, fa[0][i]), __builtin_bit_cast(bf8, fb[0][j]),
                                                                         acc[ih + i][j], 0, 0, 0);
              } else {
                acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[NP - 1][j], acc[ih + i][j], 0, 0, 0);
                acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[NP - 1][i], fb[0][j], acc[ih + i][j], 0, 0, 0);
                acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[0][j], acc[ih + i][j], 0, 0, 0);
              }
            }
        }
      }
      {
        constexpr int ND = (G::PP * G::PIECES + 3) / 4;
        hconv_wait_vm(phase + AHEAD >= nphases ? 0 : ND * (AHEAD - 1));
      }
      __builtin_amdgcn_s_barrier();
      wb = wb_plus(wb, 1);
      phase++;
    }
  }

  if (export_max && lane == 0) {   // (bits of a non-negative float order like unsigned integers)
    unsigned* slot = p.amax_out + ((v / p.nblk) & 63);
    if (rmax0 > 0.f) atomicMax(slot, __float_as_uint(rmax0));
    if (rmax1 > 0.f) atomicMax(slot + 64, __float_as_uint(rmax1));
    if (rmax2 > 0.f) atomicMax(slot + 128, __float_as_uint(rmax2));
    if (rmax3 > 0.f) atomicMax(slot + 192, __float_as_uint(rmax3));
  }
  // ------------------------------------------------------------------ epilogue (16 x 16 tiles: column = lane & 15 = voxel,
  // row = 4 (lane >> 4) + register)
  // (sign: the accumulators changed sign nchunks - 1 times)
  const float out_mult = BF ? 1.f : ldexpf((((p.nchunks - 1) >> HFLIP_SH) & 1) ? -1.f : 1.f, -((s_run == kSplitScaleNone ? 0 : s_run) + sw));
  const int rblock = p.n_off + (p.nb_off + nblk) * 64;        // first GEMM row of this workgroup
  const long long TZYX = (long long)p.TZ_ * p.TY_ * p.TX_;
  const int rlane = 4 * (lane >> 4);
  // ---- 16-byte epilogue (plain and gated, X % 4 == 0, aligned tensors).  A 16 x 16 accumulator tile has its 4 registers on 4
  // ROWS: stored as it stands, every value is its own 4-byte access (64 stores per lane, + 64 loads with the fused activation
  // backward), and the thin launches (conv0's forward, the input gradient of `last`, K = 4) are bound by exactly that number.
  // Each wave transposes its tiles through its own 1.25 KB of the (now idle) LDS -- 4 ds_write_b32, 1 ds_read_b128, conflict-
  // free with a pitch of 20 floats, no barrier: a wave's LDS operations execute in order -- and holds 4 x-NEIGHBOURS of one row.
  if (p.vec_epi && (p.epi == SR3D_EPI_PLAIN || (p.epi == SR3D_EPI_GATED && RT == 2))) {
    float* scr = reinterpret_cast<float*>(lds) + wave * 320;      // 16 rows x 20 floats
    const int tr = lane >> 2, tc = 4 * (lane & 3);                // after the transpose: row of the tile, first of 4 columns
    auto transpose = [&](const f32x4 a) -> f32x4 {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 4; r++) scr[(rlane + r) * 20 + (lane & 15)] = a[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const f32x4 t = *reinterpret_cast<const f32x4*>(scr + tr * 20 + tc);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      return t;
    };
    auto store4 = [&](float* base, const long long o, const f32x4 v, const bool f32dst) {
      if (BF && !f32dst) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[0]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[1]) << 16);
        const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[2]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[3]) << 16);
        *reinterpret_cast<u32x2*>(reinterpret_cast<__bf16*>(base) + o) = u32x2{lo, hi};
      } else {
        *reinterpret_cast<f32x4*>(base + o) = v;
      }
    };
    // The y values of the activation-fused slice, ALL of this wave's 16 tiles in one burst before anything is stored: fetched
    // tile by tile behind the transposes, each load exposed a memory latency (16 per workgroup against the ~20 us a K = 64
    // workgroup lives; hconv + 7 ms per step).  The accumulators are the only other live registers here.
    f32x4 yfuse[4][NRT];
    if (p.epi == SR3D_EPI_PLAIN && p.act_slice1 != 0) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int vt = 2 * wave + (j >> 1);
        const int oz = z0 + (vt >> 2), oy = y0 + (vt & 3), ox = x0 + (j & 1) * 16 + tc;
        const bool ok = oz < p.Z && oy < p.Y && ox < p.X;
        const long long sp = ((long long)oz * p.TY_ + oy) * p.TX_ + ox;
#pragma unroll
        for (int i = 0; i < NRT; i++) {
          const int n = rblock + i * 16 + tr;
          const int si = n < p.N ? cat_find(p.out, n) : -1;
          yfuse[j][i] = f32x4{1.f, 1.f, 1.f, 1.f};
          if (ok && si >= 0 && si + 1 == p.act_slice1) {
            const long long o = (long long)b * cat_bstride(p.out, si) + (long long)(n - cat_cbeg(p.out, si)) * TZYX + sp;
            if constexpr (BF) {
              typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
              const u32x2 yy = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(p.act_y) + o);
              yfuse[j][i] = f32x4{__builtin_bit_cast(float, yy[0] << 16), __builtin_bit_cast(float, yy[0] & 0xffff0000u),
                                  __builtin_bit_cast(float, yy[1] << 16), __builtin_bit_cast(float, yy[1] & 0xffff0000u)};
            } else {
              yfuse[j][i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.act_y) + o);
            }
          }
        }
      }
    }
    float amax_act = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int vt = 2 * wave + (j >> 1);
      const int oz = z0 + (vt >> 2), oy = y0 + (vt & 3), ox = x0 + (j & 1) * 16 + tc;
      const bool ok = oz < p.Z && oy < p.Y && ox < p.X;             // (X % 4 == 0: all four or none)
      const long long sp = ((long long)oz * p.TY_ + oy) * p.TX_ + ox;
      if (p.epi == SR3D_EPI_GATED) {
        if constexpr (RT == 2) {
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const f32x4 fa = transpose(acc[i][j] * out_mult), ga = transpose(acc[2 + i][j] * out_mult);
            const int co = rblock / 2 + i * 16 + tr;
            if (ok && co < p.Cg) {
              const float bf_ = p.bias ? p.bias[co] : 0.f, bg_ = p.bias2 ? p.bias2[co] : 0.f;
              f32x4 yv, fv, sv;
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const float sg = 1.f / (1.f + expf(-(ga[e] + bg_)));
                const float f = split_act(fa[e] + bf_, p.act);
                yv[e] = sg * f, fv[e] = f, sv[e] = sg;
              }
              const long long o = ((long long)b * p.Cg + co) * TZYX + sp;
              store4(p.y, o, yv, false);
              if (p.save_f) store4(p.save_f, o, fv, false);
              if (p.save_s) store4(p.save_s, o, sv, false);
            }
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NRT; i++) {
          const f32x4 t = transpose(acc[i][j] * out_mult);
          const int n = rblock + i * 16 + tr;
          if (!ok || n >= p.N) continue;
          const int si = cat_find(p.out, n);
          float* base = cat_ptr(p.out, si);
          if (base == nullptr) continue;
          const long long o = (long long)b * cat_bstride(p.out, si) + (long long)(n - cat_cbeg(p.out, si)) * TZYX + sp;
          const float bv = p.bias ? p.bias[n] : 0.f;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = split_act(t[e] + bv, p.act);
          if (si + 1 == p.act_slice1) {       // the fused activation backward of the producing layer (see the scalar form below)
            const f32x4 yv = yfuse[j][i];
#pragma unroll
            for (int e = 0; e < 4; e++) {
              v[e] = yv[e] > 0.f ? v[e] : 0.01f * v[e];
              amax_act = fmaxf(amax_act, fabsf(v[e]));
            }
            if (p.act_unsh) {
              // the producer is an unshuffle layer: its dL/dpre lives on the COARSE grid with 8 C channels, channel
              // ((fz * 2 + fy) * 2 + fx) * C + c.  The lane's 4 fine x-neighbours are coarse x = ox / 2, ox / 2 + 1 of fx = 0
              // (elements 0, 2) and of fx = 1 (elements 1, 3): two 8-byte (bf16: 4-byte) stores.
              const long long cvox = TZYX >> 3;                                   // coarse voxels per channel
              const int C = (int)(cat_bstride(p.out, si) / TZYX);                 // channels of the slice
              const int c = n - cat_cbeg(p.out, si);
              const int f0 = ((oz & 1) * 2 + (oy & 1)) * 2;
              const long long co = ((long long)b * 8 * C + (long long)f0 * C + c) * cvox +
                                   ((long long)(oz >> 1) * (p.TY_ >> 1) + (oy >> 1)) * (p.TX_ >> 1) + (ox >> 1);
              if constexpr (BF) {
                __bf16* d0 = reinterpret_cast<__bf16*>(base) + co;
                const unsigned e02 = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[0]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[2]) << 16);
                const unsigned e13 = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[1]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[3]) << 16);
                *reinterpret_cast<unsigned*>(d0) = e02;
                *reinterpret_cast<unsigned*>(d0 + (long long)C * cvox) = e13;
              } else {
                *reinterpret_cast<float2*>(base + co) = float2{v[0], v[2]};
                *reinterpret_cast<float2*>(base + co + (long long)C * cvox) = float2{v[1], v[3]};
              }
              continue;
            }
          }
          store4(base, o, v, p.out_f32 != 0);
        }
      }
    }
    if (!BF && p.act_amax != nullptr) {
      amax_act = split_wave_max(amax_act);
      if (lane == 0 && amax_act > 0.f) atomicMax(p.act_amax + (blockIdx.x & 63), __float_as_uint(amax_act));
    }
    return;
  }
  if (p.epi == SR3D_EPI_GATED) {
    if constexpr (RT == 2) {   // rows 0..31 of the block: features, 32..63: gates of the same 32 channels
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int vt = 2 * wave + (j >> 1);
        const int oz = z0 + (vt >> 2), oy = y0 + (vt & 3), ox = x0 + (j & 1) * 16 + (lane & 15);
        if (oz >= p.Z || oy >= p.Y || ox >= p.X) continue;
        const long long sp = ((long long)oz * p.TY_ + oy) * p.TX_ + ox;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int co = rblock / 2 + i * 16 + rlane + r;
            if (co < p.Cg) {
              float f = acc[i][j][r] * out_mult;
              if (p.bias) f += p.bias[co];
              const float g = acc[2 + i][j][r] * out_mult + (p.bias2 ? p.bias2[co] : 0.f);
              const float sg = 1.f / (1.f + expf(-g));
              f = split_act(f, p.act);
              const long long o = ((long long)b * p.Cg + co) * TZYX + sp;
              st_act<BF>(p.y, o, sg * f);
              if (p.save_f) st_act<BF>(p.save_f, o, f);
              if (p.save_s) st_act<BF>(p.save_s, o, sg);
            }
          }
      }
    }
  } else if (p.epi == SR3D_EPI_UNSHUFFLE && reinterpret_cast<const unsigned*>(p.absmax_w)[1] != 0u) {
    // Rows in unshuffle order (packed with SR3D_PACK_FWD_UNSHUFFLE): row = c * 8 + f, f = (fz * 2 + fy) * 2 + fx.  A lane's four
    // registers of a 16-row tile are f = 4 (q & 1) + r of ONE channel: registers (0, 1) and (2, 3) are x-neighbours of
    // the fine grid -> one 8-byte (bf16: 4-byte) store each, 16 lanes = 128 (64) contiguous bytes.  (In channel order
    // every value was its own 4-byte store with a stride of 8 bytes, the other half of the line coming from another
    // workgroup: up1.up.0 forward took 3.4 ms longer than its input gradient.)
    const int q = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int vt = 2 * wave + (j >> 1);
      const int oz = z0 + (vt >> 2), oy = y0 + (vt & 3), ox = x0 + (j & 1) * 16 + (lane & 15);
      if (oz >= p.Z || oy >= p.Y || ox >= p.X) continue;
#pragma unroll
      for (int i = 0; i < NRT; i++) {
        const int row0 = rblock + i * 16 + 4 * q;      // f = 4 (q & 1) .. + 3
        const int c = row0 >> 3, fz = q & 1;
        if (c >= p.unsh_C) continue;
#pragma unroll
        for (int fy = 0; fy < 2; fy++) {
          const int f0 = (fz * 2 + fy) * 2;
          const float v0 = split_act(acc[i][j][2 * fy] * out_mult + p.bias[f0 * p.unsh_C + c], p.act);
          const float v1 = split_act(acc[i][j][2 * fy + 1] * out_mult + p.bias[(f0 + 1) * p.unsh_C + c], p.act);
          const long long o = ((long long)b * p.unsh_C + c) * TZYX + ((long long)(2 * oz + fz) * p.TY_ + (2 * oy + fy)) * p.TX_ + 2 * ox;
          if constexpr (BF) {
            const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v0) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v1) << 16);
            *reinterpret_cast<unsigned*>(reinterpret_cast<__bf16*>(p.y) + o) = pk;
          } else {
            *reinterpret_cast<float2*>(p.y + o) = float2{v0, v1};
          }
        }
      }
    }
  } else if (p.epi == SR3D_EPI_UNSHUFFLE) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int vt = 2 * wave + (j >> 1);
      const int oz = z0 + (vt >> 2), oy = y0 + (vt & 3), ox = x0 + (j & 1) * 16 + (lane & 15);
      if (oz >= p.Z || oy >= p.Y || ox >= p.X) continue;
#pragma unroll
      for (int i = 0; i < NRT; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int n = rblock + i * 16 + rlane + r;
          if (n < p.N) {
            const float val = split_act(acc[i][j][r] * out_mult + p.bias[n], p.act);
            const int f = n / p.unsh_C, c = n - f * p.unsh_C;
            const long long o = ((long long)b * p.unsh_C + c) * TZYX +
                                ((long long)(2 * oz + (f >> 2)) * p.TY_ + (2 * oy + ((f >> 1) & 1))) * p.TX_ + (2 * ox + (f & 1));
            st_act<BF>(p.y, o, val);
          }
        }
    }
  } else {
    float amax_act = 0.f;   // largest |value| this thread stored into the activation-fused slice
    // (voxel positions of this lane's four tiles, once)
    long long spo[4];
    bool sok[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int vt = 2 * wave + (j >> 1);
      const int oz = z0 + (vt >> 2), oy = y0 + (vt & 3), ox = x0 + (j & 1) * 16 + (lane & 15);
      sok[j] = oz < p.Z && oy < p.Y && ox < p.X;
      spo[j] = ((long long)oz * p.TY_ + oy) * p.TX_ + ox;
    }
#pragma unroll
    for (int i = 0; i < NRT; i++) {
      // The slice that is the output y of a LeakyReLU layer: store result * lrelu'(y) -- that layer's dL/dpre -- instead of
      // dL/dy, with y read at the element's own position (same shape as the destination).  The 16 y values of a row tile are
      // fetched in ONE batch before its stores: a load behind every store (what the loop below would compile to: the
      // destination and y may alias as far as the compiler knows) exposed a memory latency per element.
      float yv[4][4];
      bool fuse_r[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = rblock + i * 16 + rlane + r;
        const int si = n < p.N ? cat_find(p.out, n) : -1;
        fuse_r[r] = si >= 0 && si + 1 == p.act_slice1 && cat_ptr(p.out, si) != nullptr;
        const long long boff = fuse_r[r] ? (long long)b * cat_bstride(p.out, si) + (long long)(n - cat_cbeg(p.out, si)) * TZYX : 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          yv[r][j] = 1.f;
          if (fuse_r[r] && sok[j]) {
            if constexpr (BF)
              yv[r][j] = __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(p.act_y)[boff + spo[j]] << 16);
            else
              yv[r][j] = reinterpret_cast<const float*>(p.act_y)[boff + spo[j]];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = rblock + i * 16 + rlane + r;
        if (n >= p.N) continue;
        const int si = cat_find(p.out, n);
        float* base = cat_ptr(p.out, si);
        if (base == nullptr) continue;
        const long long boff = (long long)b * cat_bstride(p.out, si) + (long long)(n - cat_cbeg(p.out, si)) * TZYX;   // elements
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (sok[j]) {
            float val = split_act(acc[i][j][r] * out_mult + bv, p.act);
            if (fuse_r[r]) {
              val = yv[r][j] > 0.f ? val : 0.01f * val;       // (the expression of lrelu_bwd_kernel: bit-identical in fp32)
              amax_act = fmaxf(amax_act, fabsf(val));
            }
            if (BF && p.out_f32)
              base[boff + spo[j]] = val;      // (fp32 destination of a bf16-storage layer)
            else
              st_act<BF>(base, boff + spo[j], val);
          }
        }
      }
    }
    if (!BF && p.act_amax != nullptr) {   // wave-uniform condition
      amax_act = split_wave_max(amax_act);
      if (lane == 0 && amax_act > 0.f) atomicMax(p.act_amax + (blockIdx.x & 63), __float_as_uint(amax_act));
    }
  }
}

// ---- max |x| of a tensor into *slot (bits of a non-negative float order like unsigned integers)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, unsigned* slot) {
  float m = 0.f;
  const long long n4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n / 4 : 0;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 q = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  // one atomic per workgroup: with one per wave, 8192 atomics on ONE address took 50 us per call (43 calls per step)
  __shared__ float wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (m > 0.f) atomicMax(slot, __float_as_uint(m));
  }
}

// ---- weight split + packing: image [row block][chunk][tap 27][part][row tile][channel half][32 rows][8 ch] fp16
struct HPackParams {
  const float* w1;
  const float* w2;
  const float* absmax_w;
  unsigned* hdr;   // image header: [0] max |w| (bits), [1] 1 = rows in voxel-unshuffle order
  _Float16* img;
  int Cout, Cin, kind, K, N, nchunks, nblk, RT, n_off;
  int bf;   // 1: one bf16 part per weight (no scaling) instead of the [hi | lo] fp16 pair
  int itail;   // 1: the last chunk (1 .. 5 channels) in im2col form (see hconv_itail_host)
  int unsh_C;   // > 0: rows in voxel-unshuffle order, GEMM row r = c * 8 + f holds output channel f * unsh_C + c
  int rbeg[SR3D_MAX_SRC + 1];
  int cbeg[SR3D_MAX_SRC];
};

__global__ __launch_bounds__(256) void hconv_pack_kernel(const HPackParams p) {
  const int sw = p.bf ? 0 : split_scale_exp(*p.absmax_w);
  const float w_mult = ldexpf(1.f, sw == kSplitScaleNone ? 0 : sw);
  // items of 8 channels: (row block, chunk, 16-row tile, channel half, row, tap 0..27); tap 27 is the zero dummy
  const long long total = (long long)p.nblk * p.nchunks * 28 * p.RT * 64;
  if (blockIdx.x == 0 && threadIdx.x == 0) p.hdr[1] = p.unsh_C > 0 ? 1u : 0u;   // row order of the image (read by the kernel)
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    // the tap runs fastest over the threads: the 27 taps of one (row, channel) are contiguous in the weight tensor, so a
    // wave's reads are 108-byte runs (with the row fastest every lane read its own 108-byte segment: 4.5 ms per step)
    long long r = e;
    const int tap = r % 28;
    r /= 28;
    const int row = r % 16;
    r /= 16;
    const int h = r % 2;
    r /= 2;
    const int rt = r % (2 * p.RT);          // 16-row tile inside the block
    r /= 2 * p.RT;
    const int chunk = r % p.nchunks;
    const int nb = r / p.nchunks;
    const int n = p.n_off + nb * (32 * p.RT) + rt * 16 + row;
    const bool tailc = hconv_tail(p.K, p.bf != 0, p.itail != 0) && chunk + 1 == p.nchunks;   // half-chunk tail: 4 taps x 8 channels per MFMA
    if (tailc && h == 1) continue;                                             // (its second channel half does not exist)
    const bool itc = p.itail != 0 && chunk + 1 == p.nchunks;   // im2col tail: the item (tap, h) stands for (tail channel sc, K group g)
    int it_sc = 0, it_g = 0;
    if (itc) {
      const int idx = tap * 2 + h;
      if (idx >= 4 * (p.K & 15)) continue;
      it_sc = idx >> 2, it_g = idx & 3;
    }
    const float* w = nullptr;   // -> w[.][k = 0][tap 0]; element (k, tap) at w[k * kstride + tapidx]
    long long kstride = 27;
    int tapidx = tap;
    if (tap < 27 || itc) {
      if (p.kind == SR3D_PACK_FWD) {
        // (unshuffle order: the two x-neighbours (f & 1) of an output voxel sit in adjacent accumulator registers)
        const int nsrc = p.unsh_C > 0 ? (n & 7) * p.unsh_C + (n >> 3) : n;
        if (n < p.N) w = p.w1 + (long long)nsrc * p.Cin * 27;
      } else if (p.kind == SR3D_PACK_FWD_GATED) {
        const int co = (n >> 6) * 32 + (n & 31);   // 64-row block = 32 feature rows, then the 32 gate rows
        if (co < p.Cout) w = ((n & 32) ? p.w2 : p.w1) + (long long)co * p.Cin * 27;
      } else if (n < p.N) {   // input gradient: rows = input channels that need a gradient, K = output channels, taps mirrored
        const int si = (n >= p.rbeg[1]) + (n >= p.rbeg[2]) + (n >= p.rbeg[3]);
        const int ci = p.cbeg[si] + (n - p.rbeg[si]);
        w = p.w1 + (long long)ci * 27;
        kstride = (long long)p.Cin * 27;
        tapidx = 26 - tap;
      }
    }
    h8 hi, lo;
    bf8 wb;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      int k = chunk * HKC + h * 8 + j;
      if (itc) {   // element j = tap 8 g + j of tail channel sc (taps 27 .. 31: zeros)
        k = chunk * HKC + it_sc;
        const int t = 8 * it_g + j;
        tapidx = (p.kind == SR3D_PACK_BWD || p.kind == SR3D_PACK_BWD_GATED) ? 26 - t : t;
        if (t > 26) k = p.K;   // -> 0
      }
      float val = 0.f;
      if (w != nullptr && k < p.K) {
        if (p.kind == SR3D_PACK_BWD || p.kind == SR3D_PACK_BWD_GATED) {
          const float* src = k < p.Cout ? w + (long long)k * kstride : (p.w2 + (w - p.w1)) + (long long)(k - p.Cout) * kstride;
          val = src[tapidx];
        } else {
          val = w[(long long)k * kstride + tapidx];
        }
      }
      // alternating sign: see the kernel's chunk loop (not in the bf16 form: the truncation bias it cancels, ~5e-7 of the
      // output's rms, is far below the bf16 rounding of the output)
      const float s = val * ((((chunk >> HFLIP_SH) & 1) && !p.bf) ? -w_mult : w_mult);
      const _Float16 a = (_Float16)s;
      hi[j] = a;
      lo[j] = (_Float16)(s - (float)a);
      wb[j] = (__bf16)s;
    }
    // fragment of (pair = tap / 2, part, tile): lane = row + 16 * (2 * (tap & 1) + h)  [MFMA K group = tap of the pair, half]
    int l = row + 16 * (2 * (tap & 1) + h);
    int phs = tap / 2;
    if (tailc) l = row + 16 * (tap & 3), phs = tap / 4;   // lane group = tap of the quad; phases 0 .. 6 of the chunk
    if (itc) l = row + 16 * it_g, phs = it_sc;            // K group = 8 taps of the channel; one pair slot per tail channel
    if (p.bf) {
      const long long frag = (((long long)nb * p.nchunks + chunk) * 14 + (itc ? phs : tap / 2)) * (2 * p.RT) + rt;
      *reinterpret_cast<bf8*>(p.img + frag * 512 + l * 8) = wb;
      continue;
    }
    const long long frag0 = ((((long long)nb * p.nchunks + chunk) * 14 + phs) * 2 + 0) * (2 * p.RT) + rt;
    const long long frag1 = frag0 + 2 * p.RT;
    *reinterpret_cast<h8*>(p.img + frag0 * 512 + l * 8) = hi;
    *reinterpret_cast<h8*>(p.img + frag1 * 512 + l * 8) = lo;
  }
}

}  // namespace

int sr3d_absmax_launch(const float* x, long long n, unsigned* slot, hipStream_t st) {
  long long blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, slot);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

namespace {

inline void row_split(int rows, int* n2, int* n1) {   // 64-row blocks, and one last block of <= 32 rows
  const int nfull = rows / 64, rem = rows - nfull * 64;
  *n2 = nfull + (rem > 32 ? 1 : 0);
  *n1 = (rem > 0 && rem <= 32) ? 1 : 0;
}

}  // namespace

// SR3D_SPLIT_F16: 0 = off (fp32 Winograd everywhere), 1 / unset = on where the launch fills the chip, 2 = on for every
// eligible layer whatever its size (tests).  Read per call: tests and tools switch it at run time.
int sr3d_hconv_mode() {
  const char* e = getenv("SR3D_SPLIT_F16");
  return e == nullptr ? 1 : atoi(e);
}

// header (64 bytes: max |w|) + region A (64-row blocks) + region B (one 32-row block)
size_t sr3d_hconv_image_bytes(int rows, int K, bool bf) {
  int n2, n1;
  row_split(rows, &n2, &n1);
  const size_t w2 = bf ? HGeo<2, true>::WPHASE : HGeo<2>::WPHASE, w1 = bf ? HGeo<1, true>::WPHASE : HGeo<1>::WPHASE;
  return 64 + (size_t)ceil_div(K, HKC) * HPH * ((size_t)n2 * w2 + (size_t)n1 * w1);
}

int sr3d_hconv_pack(int kind, int Cout, int Cin, int rows, int K, const float* w1, const float* w2, const int* rbeg,
                    const int* cbeg, void* image, bool bf, hipStream_t st, int unsh_C) {
  unsigned* hdr = (unsigned*)image;
  SrProfScope prof(SR3D_PROF_PACK, (bf ? 3.0 : 4.0) * (double)rows * K * 27 * 2, st);
  const long long nw = (long long)Cout * Cin * 27;
  if (!bf) {   // max |w| -> the layer's power-of-two scale (bf16 weights are not scaled)
    if (int rc = sr3d_zero_words(hdr, 16, st)) return rc;
    if (int rc = sr3d_absmax_launch(w1, nw, hdr, st)) return rc;
    if (w2 != nullptr)
      if (int rc = sr3d_absmax_launch(w2, nw, hdr, st)) return rc;
  }
  HPackParams p{};
  p.bf = bf ? 1 : 0;
  p.itail = hconv_itail_host(K, bf) ? 1 : 0;
  p.hdr = hdr, p.unsh_C = unsh_C;
  SR3D_CHECK(unsh_C == 0 || (kind == SR3D_PACK_FWD && rows == 8 * unsh_C), SR3D_E_ARG, "hconv pack: unshuffle order needs a plain forward image with 8 * C rows");
  p.w1 = w1, p.w2 = w2, p.absmax_w = (const float*)hdr;
  p.Cout = Cout, p.Cin = Cin, p.kind = kind, p.K = K, p.N = rows, p.nchunks = ceil_div(K, HKC);
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
    p.img = body + (region == 0 ? 0 : (size_t)n2 * p.nchunks * HPH * ((bf ? HGeo<2, true>::WPHASE : HGeo<2>::WPHASE) / 2));
    const long long total = (long long)p.nblk * p.nchunks * 28 * p.RT * 64;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(hconv_pack_kernel, dim3(blocks), dim3(256), 0, st, p);
    SR3D_HIP(hipGetLastError());
  }
  return SR3D_OK;
}

namespace {

template <bool BF, int PAIR>
int hconv_launch_t(SrHconvParams& p, int B, int n2, int n1, long long nsp, hipStream_t st) {
  static SrPerDevice setup;   // (the attribute is per device, not per thread)
  if (int rc = setup.once([&]() -> int {
        SR3D_HIP(hipFuncSetAttribute((const void*)hconv_kernel<2, BF, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HGeo<2, BF>::LDS));
        SR3D_HIP(hipFuncSetAttribute((const void*)hconv_kernel<1, BF, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HGeo<1, BF>::LDS));
        return SR3D_OK;
      }))
    return rc;
  void* tok = nullptr;
  if (sr3d_prof_active()) {
    const double rows = p.epi == SR3D_EPI_GATED ? 2.0 * p.Cg : (double)p.N;
    // (launches that do not fill the chip's 512 workgroup slots -- the small grids of U-Net levels 3-4 -- are a family of their own)
    const bool fills = nsp * (n2 + n1) * B >= 448;
    sr3d_prof_begin(fills ? SR3D_PROF_HCONV : SR3D_PROF_HCONV_SMALL, 2.0 * 27 * p.K * rows * (double)p.Z * p.Y * p.X * B, st, &tok);
  }
  constexpr size_t lds2 = HGeo<2, BF>::LDS, lds1 = HGeo<1, BF>::LDS;
  constexpr size_t wphase2 = HGeo<2, BF>::WPHASE;
  if (n2 > 0) {
    p.nblk = n2, p.nb_off = 0;
    hipLaunchKernelGGL((hconv_kernel<2, BF, PAIR>), dim3((unsigned)(nsp * n2), B), dim3(HNT), lds2, st, p);
  }
  if (n1 > 0) {
    // region B: its blocks are 32 rows; express the offsets in the kernel's own units
    SrHconvParams q = p;
    q.nblk = 1, q.nb_off = 0;
    if (n2 > 0) q.amax_out = nullptr;   // (region A's first row block exports the maxima of x)
    q.wimg = (const unsigned char*)p.wimg + (size_t)n2 * p.nchunks * HPH * wphase2;
    q.n_off = p.n_off + n2 * 64;
    hipLaunchKernelGGL((hconv_kernel<1, BF, PAIR>), dim3((unsigned)nsp, B), dim3(HNT), lds1, st, q);
  }
  sr3d_prof_end(tok, st);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

}  // namespace

int sr3d_hconv_launch(SrHconvParams& p, const void* image, int B, bool bf, hipStream_t st) {
  SR3D_CHECK((long long)p.Z * p.Y * p.X < (1ll << 29), SR3D_E_ARG, "split-f16 conv: more than 2^29 voxels per channel");
  SR3D_CHECK(B <= 65535, SR3D_E_ARG, "split-f16 conv: batch too large");
  p.absmax_w = (const float*)image;
  p.wimg = (const unsigned char*)image + 64;
  p.ntz = ceil_div(p.Z, 2), p.nty = ceil_div(p.Y, 4), p.ntx = ceil_div(p.X, 32);
  p.nchunks = ceil_div(p.K, HKC);
  p.itail = hconv_itail_host(p.K, bf) ? 1 : 0;   // (sr3d_hconv_pack asked the same question)
  int n2, n1;
  row_split(p.N, &n2, &n1);
  const long long nsp = (long long)p.ntz * p.nty * p.ntx;
  SR3D_CHECK(nsp * (n2 + n1) < (1ll << 31), SR3D_E_ARG, "split-f16 conv: grid too large");
  // wide halo loads: fp32 quads need X % 4 == 0 and 16-byte aligned tensors (then every channel row is), bf16 pairs even
  // rows and 4-byte aligned tensors
  // (bf16: 8-byte quads where X % 4 == 0 and the tensors are 8-byte aligned, else 4-byte pairs; SR3D_HCONV_BF16_WIDE=1 keeps the pairs)
  const int wide = bf ? 2 : 4;
  bool pair = p.X % wide == 0 && getenv("SR3D_HCONV_NO_PAIR") == nullptr;
  bool bquad = bf && p.X % 4 == 0 && !(getenv("SR3D_HCONV_BF16_WIDE") && atoi(getenv("SR3D_HCONV_BF16_WIDE")) == 1);
  for (int i = 0; i < p.in.n; i++) {
    pair = pair && (reinterpret_cast<uintptr_t>(p.in.ptr[i]) & (bf ? 3 : 15)) == 0;
    bquad = bquad && (reinterpret_cast<uintptr_t>(p.in.ptr[i]) & 7) == 0;
  }
  SR3D_CHECK(!p.act_unsh || (p.X % 4 == 0 && p.Z % 2 == 0 && p.Y % 2 == 0), SR3D_E_ARG, "split-f16 conv: the unshuffle-fused slice needs an even grid with X %% 4 == 0");
  // the 16-byte epilogue: rows of 4 x-neighbours, every destination (and the tensors read or written next to it) aligned
  {
    const uintptr_t am = (bf && !p.out_f32) ? 7 : 15;
    bool vec = p.X % 4 == 0 && p.TX_ == p.X && getenv("SR3D_HCONV_SCALAR_EPILOGUE") == nullptr && p.epi != SR3D_EPI_UNSHUFFLE;
    if (p.epi == SR3D_EPI_GATED) {
      vec = vec && (reinterpret_cast<uintptr_t>(p.y) & am) == 0 && (reinterpret_cast<uintptr_t>(p.save_f) & am) == 0 &&
            (reinterpret_cast<uintptr_t>(p.save_s) & am) == 0;
    } else {
      for (int i = 0; i < p.out.n; i++) vec = vec && (reinterpret_cast<uintptr_t>(p.out.ptr[i]) & am) == 0;
      vec = vec && (reinterpret_cast<uintptr_t>(p.act_y) & (bf ? 7 : 15)) == 0;
    }
    p.vec_epi = vec ? 1 : 0;
    SR3D_CHECK(!p.act_unsh || vec, SR3D_E_ARG, "split-f16 conv: the unshuffle-fused slice needs the 16-byte epilogue (aligned tensors)");
  }
  if (!bf) return pair ? hconv_launch_t<false, 1>(p, B, n2, n1, nsp, st) : hconv_launch_t<false, 0>(p, B, n2, n1, nsp, st);
  if (pair && bquad) return hconv_launch_t<true, 2>(p, B, n2, n1, nsp, st);
  return pair ? hconv_launch_t<true, 1>(p, B, n2, n1, nsp, st) : hconv_launch_t<true, 0>(p, B, n2, n1, nsp, st);
}
