// Weight gradient of the stride-1 3x3x3 convolutions on the split-f16 scheme of sr3d_hconv.hip (fp32 operands as two
// fp16 halves, three v_mfma_f32_16x16x32_f16 per product group, fp32 accumulate):
//
//   dW[n][c][kz,ky,kx] = sum_{b,z,y,x'} dY[n][z][y][x' - kx + 1] * X[c][z + kz - 1][y + ky - 1][x']
//
// One MFMA reduces over the K = 32 consecutive x' of a row segment: A = a dY row segment (16 rows n), B = an X row
// segment (16 columns c).  The tap shifts in z and y pick WHICH X row a wave reads; the shift in x would make one operand start one
// element off a 16-byte boundary -- so the smaller operand (dY) is staged three times, shifted by -1, 0, +1, and every
// fragment is one aligned ds_read_b128.
//
// Workgroup = 12 waves: an (n-block of 32 * RT rows) x (c-block of 32 channels) x (x segment of 32 voxels) x (a range
// of (b, z, y) rows, split-K).  Wave w owns one 16-row tile of the n block and one kz plane: 9 taps x 2 channel tiles,
// 54 MFMAs per step on every wave (see "wave roles" in the kernel).
// It marches along y: per step ONE new X row of each of the 3 z planes (4 y slots per plane in LDS) and one dY row
// (3 shifted copies, double-buffered) are staged, by waves 0-5 (X) and 6-9 (dY), one 8-voxel piece per thread,
// loaded a step ahead into registers, then scaled, split and written; one barrier per step.
// Scales: one power of two per tensor SLICE of the virtual concats from max|x_i| / max|dY_i| (separate pass,
// sr3d_absmax): exact, no overflow; the feature and gate gradients of a gated layer, or features and the building mask,
// differ by orders of magnitude and each keeps its own 22 bits.
// The f16 MFMA's truncation bias (sr3d_hconv.hip) is cancelled by flipping the sign of the dY rows and of the
// accumulators every 32 rows.  Partial sums per (split, x segment) go to a slab [.][tap][n][c]; a second kernel adds them
// in a fixed order (deterministic), undoes the scaling and writes dW[n][c][tap].
#include "sr3d_split_f16.h"

#include <limits.h>
#include <stdlib.h>

#include <type_traits>

namespace {


constexpr int WNT = 768;
// Bytes per 32-voxel fp16 row in LDS.  A fragment read is ds_read_b128 at (lane & 15) * PITCH + (lane >> 4) * 16, and
// gfx950 services it in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): rows
// {0-3, 12-15} at K group 0 together with rows {4-11} at K group 1.  In 16-byte slots mod 16 a pitch of 96 (6 slots)
// puts the first set on the even and the second on the odd slots: conflict-free.  (80 bytes, the first version, paired
// row r with row r + 3 of the other K group: SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE.)
constexpr int PITCH = 96;
// BF = true: x and dY are STORED as bfloat16 (sr3d_conv_desc_t.dtype): one part, no scaling, one
// v_mfma_f32_16x16x32_bf16 per tile; the three shifted dY copies are made with v_alignbit on the packed pairs.
template <int RT, bool BF = false>
struct WGeo {
  static constexpr int NP = BF ? 1 : 2;                // operand parts
  static constexpr int XROW = NP * 32 * PITCH;         // one X row: [part][c 32][PITCH]
  static constexpr int XBYTES = 12 * XROW;             // 3 planes x 4 y slots
  static constexpr int DCOPY = NP * 32 * RT * PITCH;   // one shifted copy of a dY row: [part][n][PITCH]
  static constexpr int DROW = 3 * DCOPY;
  static constexpr size_t LDS = XBYTES + 2 * (size_t)DROW;
  static constexpr int NDY = 32 * RT * 4;              // dY staging items per step (n, 8-voxel piece)
};

// exponent of the per-tensor scale (0, i.e. no scaling, for an all-zero tensor)
__host__ __device__ inline int scale_exp_of(float amax) {
  const int s = split_scale_exp(amax);
  return s == kSplitScaleNone ? 0 : s;
}

struct HwParams {
  ChanCat x, dy;
  int cu, N;                 // channels / rows covered
  int B, Z, Y, X;
  int nnb, ncb, nseg, S;     // row blocks, channel blocks, x segments, splits
  long long rows_per_split;  // (b, z, y) rows per split
  int Npad, Cpad;
  float* slab;               // [S * nseg][27][Npad][Cpad]
  const float* amax;         // [0..3] = max|x slice i|, [4..7] = max|dy slice i|
  int xcd_order;             // consecutive virtual workgroup ids on one XCD (see the kernel)
};

// PF: how many steps AHEAD of the usual one the global loads of a row are issued (register ring of PF + 1 pieces).  With
// PF = 0 a piece is loaded in step t and written to LDS in step t + 1: one step time (~1.3 us of MFMAs at best) must
// cover the whole HBM / L2 latency of the slowest of the 640 staging threads, and every step ends in a barrier -- the
// bf16 form of this kernel, with a third of the MFMAs and half the bytes, ran only 16 % faster than the split-f16 form.
template <int RT, bool BF, int PF>
__global__ __launch_bounds__(WNT) void hwgrad_kernel(const HwParams p) {
  constexpr int NS = PF + 1;
  using G = WGeo<RT, BF>;
  constexpr int XROW = G::XROW, XBYTES = G::XBYTES;
  constexpr int ESZ = BF ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* Xs = lds;
  unsigned char* Ds = lds + XBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < WNT / 64);

  // workgroup -> (row block, channel block, x segment, split).  The nnb x ncb workgroups of one (segment, split) read the
  // SAME X and dY rows: they get consecutive virtual ids, and consecutive virtual ids go to ONE XCD (hardware deals
  // block ids round-robin over the 8 XCDs, each with its own L2), so that the group's re-reads are L2 hits of that XCD
  // instead of eight HBM fetches (measured before: 13.8 GB fetched per launch for 2.1 GB of operands).
  int v;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    v = p.xcd_order ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3) : bid;
  }
  const int nb = v % p.nnb;
  v /= p.nnb;
  const int cb = v % p.ncb;
  v /= p.ncb;
  const int seg = v % p.nseg;
  const int split = v / p.nseg;
  const int x0 = seg * 32;
  const long long YX = (long long)p.Y * p.X, ZYX = YX * p.Z;

  float mx = 1.f, md = 1.f;   // this thread's staging item: scale of ITS slice

  // ---- staging role of this thread (fixed for the whole kernel)
  //   waves 0..5: X item  = (plane dz 0..2, channel 0..31, piece q 0..3)
  //   waves 6..9: dY item = (row n, piece q), 32 * RT * 4 of them
  const bool is_x = wave < 6;
  int it_c = 0, it_q = 0, it_dz = 0, it_n = 0;
  const unsigned char* src = nullptr;   // channel / row base of sample 0 (per lane), as bytes
  long long src_b = 0;                  // elements between samples
  bool it_on = false;
  if (is_x) {
    it_dz = tid / 128, it_c = (tid % 128) / 4, it_q = tid & 3;
    const int c = cb * 32 + it_c;
    if (c < p.cu) {
      const int si = cat_find(p.x, c);
      src = reinterpret_cast<const unsigned char*>(cat_ptr(p.x, si)) + (long long)(c - cat_cbeg(p.x, si)) * ZYX * ESZ;
      src_b = cat_bstride(p.x, si);
      if constexpr (!BF) mx = ldexpf(1.f, scale_exp_of(p.amax[si]));
      it_on = x0 + 8 * it_q < p.X;
    }
  } else {
    const int i = tid - 384;
    if (i < G::NDY) {
      it_n = i / 4, it_q = i & 3;
      const int n = nb * (32 * RT) + it_n;
      if (n < p.N) {
        const int si = cat_find(p.dy, n);
        src = reinterpret_cast<const unsigned char*>(cat_ptr(p.dy, si)) + (long long)(n - cat_cbeg(p.dy, si)) * ZYX * ESZ;
        src_b = cat_bstride(p.dy, si);
        if constexpr (!BF) md = ldexpf(1.f, scale_exp_of(p.amax[4 + si]));
        it_on = x0 + 8 * it_q < p.X;
      }
    }
  }
  const bool stager = is_x || (tid - 384) < G::NDY;
  const int xq = x0 + 8 * it_q;
  float pvr[NS][BF ? 1 : 10];   // piece: elements -1 .. 8 (X items use 0..7); ring of NS pieces in flight
  u32x4 pqr[NS];                // BF: the 8 bf16 elements of the piece as they are, and its two neighbours
  unsigned pprevr[NS], pnextr[NS];
#pragma unroll
  for (int u = 0; u < NS; u++) {
#pragma unroll
    for (int j = 0; j < (BF ? 1 : 10); j++) pvr[u][j] = 0.f;
    pqr[u] = u32x4{0u, 0u, 0u, 0u}, pprevr[u] = pnextr[u] = 0u;
  }
  // loads the piece of row (b, z, y) into ring entry st (zeros when the row is outside the grid)
  auto load_piece = [&](const int st, const int b, const int z, const int y) {
    float* pv = pvr[st];
    u32x4& pq = pqr[st];
    unsigned &pprev = pprevr[st], &pnext = pnextr[st];
    const bool ok = it_on && (unsigned)z < (unsigned)p.Z && (unsigned)y < (unsigned)p.Y;
    if constexpr (BF) {
      if (ok) {
        const unsigned short* r = reinterpret_cast<const unsigned short*>(src) + (long long)b * src_b + (long long)z * YX + (long long)y * p.X + xq;
        pq = *reinterpret_cast<const u32x4*>(r);
        if (!is_x) {
          pprev = xq > 0 ? (unsigned)r[-1] : 0u;
          pnext = xq + 8 < p.X ? (unsigned)r[8] : 0u;
        }
      } else {
        pq = u32x4{0u, 0u, 0u, 0u};
        pprev = pnext = 0u;
      }
      return;
    }
    if (ok) {
      const float* r = reinterpret_cast<const float*>(src) + (long long)b * src_b + (long long)z * YX + (long long)y * p.X + xq;
      const f32x4 a = *reinterpret_cast<const f32x4*>(r), c4 = *reinterpret_cast<const f32x4*>(r + 4);
      pv[1] = a.x, pv[2] = a.y, pv[3] = a.z, pv[4] = a.w, pv[5] = c4.x, pv[6] = c4.y, pv[7] = c4.z, pv[8] = c4.w;
      if (!is_x) {
        pv[0] = xq > 0 ? r[-1] : 0.f;
        pv[9] = xq + 8 < p.X ? r[8] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 10; j++) pv[j] = 0.f;
    }
  };
  auto split2 = [&](const float a, const float b, const float mult, unsigned& hi, unsigned& lo) { split_pair(a, b, mult, hi, lo); };
  // X row (plane slot, y slot) / dY buffer addresses
  auto write_piece = [&](const int st, const int yslot_x, const int dbuf, const float dsign) {
    if (!stager) return;
    const float* pv = pvr[st];
    const u32x4 pq = pqr[st];
    const unsigned pprev = pprevr[st], pnext = pnextr[st];
    if constexpr (BF) {
      if (is_x) {
        *reinterpret_cast<u32x4*>(Xs + (it_dz * 4 + yslot_x) * XROW + it_c * PITCH + it_q * 16) = pq;
      } else {
        const unsigned sm = dsign < 0.f ? 0x80008000u : 0u;   // sign flip of both bf16 halves
        const unsigned d0 = pq[0] ^ sm, d1 = pq[1] ^ sm, d2 = pq[2] ^ sm, d3 = pq[3] ^ sm;
        const unsigned pl = (pprev ^ (sm & 0xffffu)) << 16, nh = pnext ^ (sm & 0xffffu);
        // copy cp holds dY[x' - cp + 1]: cp = 0 -> elements 1..8, cp = 1 -> 0..7, cp = 2 -> -1..6
        const u32x4 c0 = {__builtin_amdgcn_alignbit(d1, d0, 16), __builtin_amdgcn_alignbit(d2, d1, 16),
                          __builtin_amdgcn_alignbit(d3, d2, 16), __builtin_amdgcn_alignbit(nh, d3, 16)};
        const u32x4 c1 = {d0, d1, d2, d3};
        const u32x4 c2 = {__builtin_amdgcn_alignbit(d0, pl, 16), __builtin_amdgcn_alignbit(d1, d0, 16),
                          __builtin_amdgcn_alignbit(d2, d1, 16), __builtin_amdgcn_alignbit(d3, d2, 16)};
        unsigned char* d = Ds + dbuf * G::DROW + it_n * PITCH + it_q * 16;
        *reinterpret_cast<u32x4*>(d) = c0;
        *reinterpret_cast<u32x4*>(d + G::DCOPY) = c1;
        *reinterpret_cast<u32x4*>(d + 2 * G::DCOPY) = c2;
      }
      return;
    }
    if (is_x) {
      unsigned hi[4], lo[4];
#pragma unroll
      for (int k = 0; k < 4; k++) split2(pv[1 + 2 * k], pv[2 + 2 * k], mx, hi[k], lo[k]);
      unsigned char* d = Xs + (it_dz * 4 + yslot_x) * XROW + it_c * PITCH + it_q * 16;
      *reinterpret_cast<u32x4*>(d) = u32x4{hi[0], hi[1], hi[2], hi[3]};
      *reinterpret_cast<u32x4*>(d + 32 * PITCH) = u32x4{lo[0], lo[1], lo[2], lo[3]};
    } else {
      // the 10 elements -1 .. 8 are split ONCE, as the pairs (-1, 0), (1, 2), ..., (7, 8); copy cp holds dY[x' - cp + 1]:
      //   cp = 0: elements 1 .. 8 = pairs 1 .. 4;  cp = 2: elements -1 .. 6 = pairs 0 .. 3;
      //   cp = 1: elements 0 .. 7 = the high half of pair k with the low half of pair k + 1 (v_alignbit)
      const float m = md * dsign;
      unsigned ph[5], pl[5];
#pragma unroll
      for (int k = 0; k < 5; k++) split2(pv[2 * k], pv[2 * k + 1], m, ph[k], pl[k]);
      unsigned char* d = Ds + dbuf * G::DROW + it_n * PITCH + it_q * 16;
      *reinterpret_cast<u32x4*>(d) = u32x4{ph[1], ph[2], ph[3], ph[4]};
      *reinterpret_cast<u32x4*>(d + 32 * RT * PITCH) = u32x4{pl[1], pl[2], pl[3], pl[4]};
      u32x4 mh, ml;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        mh[k] = __builtin_amdgcn_alignbit(ph[k + 1], ph[k], 16);
        ml[k] = __builtin_amdgcn_alignbit(pl[k + 1], pl[k], 16);
      }
      *reinterpret_cast<u32x4*>(d + G::DCOPY) = mh;
      *reinterpret_cast<u32x4*>(d + G::DCOPY + 32 * RT * PITCH) = ml;
      *reinterpret_cast<u32x4*>(d + 2 * G::DCOPY) = u32x4{ph[0], ph[1], ph[2], ph[3]};
      *reinterpret_cast<u32x4*>(d + 2 * G::DCOPY + 32 * RT * PITCH) = u32x4{pl[0], pl[1], pl[2], pl[3]};
    }
  };

  // ---- wave roles.  Wave w = (16-row tile i of the n block, kz plane[, 16-channel tile]): all 9 (ky, kx) taps of ITS plane
  // for ITS rows.  Per step it reads the three shifted dY copies of its row tile once (6 fragments) and per ky one X row
  // of its plane (2 channel tiles x [hi | lo] = 4 fragments; 32-row blocks: one channel tile, 2), and multiplies
  // 9 x 2 x 3 = 54 MFMAs: 18 fragment reads per 54 MFMAs, the same 54 on every wave and SIMD.  (The first version gave
  // each wave a (kz, ky) group with kx = 0, 1 and three waves the kx = 2 taps: 24 - 36 reads per 48 - 72 MFMAs, every wave
  // re-reading all four row tiles -- a third more LDS traffic -- and 144 / 168 / 168 / 168 MFMAs on the four SIMDs.)
  // v_mfma_f32_16x16x32_f16: one MFMA reduces over the WHOLE 32-voxel row segment; the tiles are 16 rows x 16 channels
  // (4 accumulator registers); the 16x16x32 shape is 14 % more power-efficient than 32x32x16 on this part
  // (tools/mfma_rate.hip), and this kernel runs at the power limit.
  constexpr int NT = 2 * RT;                  // 16-row tiles of the n block
  constexpr int JW = 12 / (3 * NT);           // waves sharing a (row tile, plane): 1 (64-row blocks) or 2 (32-row blocks)
  constexpr int NJ = 2 / JW;                  // 16-channel tiles per wave
  const int wi = wave % NT, wkz = (wave / NT) % 3, wj = wave / (3 * NT);
  f32x4 acc[3][3][NJ];                        // [ky][kx][channel tile]
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int jj = 0; jj < NJ; jj++) acc[a][k][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
  float acc_sign = 1.f;
  const int fr = (lane & 15) * PITCH + (lane >> 4) * 16;   // fragment: row / channel lane & 15, voxels 8 (lane >> 4) .. +8
  const int fa = wi * 16 * PITCH + fr;                      // this wave's row tile inside a dY copy
  const int fb = wkz * 4 * XROW + wj * NJ * 16 * PITCH + fr;   // its plane and first channel tile inside the X rows

  // ---- rows of this split
  const long long rows_total = (long long)p.B * p.Z * p.Y;
  long long r0 = (long long)split * p.rows_per_split, r1 = r0 + p.rows_per_split;
  if (r1 > rows_total) r1 = rows_total;
  while (r0 < r1) {
    const long long plane = r0 / p.Y;                 // (b, z)
    const int b = (int)(plane / p.Z), z = (int)(plane - (long long)b * p.Z);
    const int ya = (int)(r0 - plane * p.Y);
    const long long pend = (plane + 1) * p.Y;
    const int yb = (int)((r1 < pend ? r1 : pend) - plane * p.Y);
    // step t: write what was loaded in step t - NS (X rows t+2, dY row t+1) from ring entry u, refill that entry with
    // the rows of step t + NS (X rows t+2+NS, dY row t+1+NS), multiply row t
    for (int t0 = ya - 3 - NS; t0 < yb; t0 += NS) {
#pragma unroll
    for (int u = 0; u < NS; u++) {
      const int t = t0 + u;
      if (t >= yb) break;
      // (staging AFTER the step's MFMAs instead of before them -- legal, the rows go to slots row t does not read -- was
      // tried: no change in the split form, 30 % slower in the bf16 form, profiles/r03d_ab_hwgrad_late_staging.log; STAGGERED --
      // the X stagers before, the dY stagers after their MFMAs, two early waves and a late one on every SIMD -- 6 % / 10 %
      // slower, profiles/r03f_ab_hwgrad_staggered_staging.log)
      if (t > ya - 4) {
        const long long rr = plane * p.Y + (t + 1);
        write_piece(u, (t + 2) & 3, (t + 1) & 1, ((rr >> 5) & 1) ? -1.f : 1.f);
      }
      if (is_x)
        load_piece(u, b, z + it_dz - 1, t + 2 + NS);
      else
        load_piece(u, b, z, (t + 1 + NS >= ya && t + 1 + NS < yb) ? t + 1 + NS : -1);
      if (t >= ya) {
        const long long rr = plane * p.Y + t;
        const float sgn = ((rr >> 5) & 1) ? -1.f : 1.f;
        if (sgn != acc_sign) {   // (wave-uniform) sign alternation, see the header
#pragma unroll
          for (int a = 0; a < 3; a++)
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
              for (int jj = 0; jj < NJ; jj++) acc[a][k][jj] = -acc[a][k][jj];
          acc_sign = sgn;
        }
        const unsigned char* da = Ds + (t & 1) * G::DROW + fa;
        h8 ah[3], al[3];   // the three shifted copies of this wave's dY rows: copy k holds dY[x' - k + 1]
#pragma unroll
        for (int k = 0; k < 3; k++) {
          ah[k] = *reinterpret_cast<const h8*>(da + k * G::DCOPY);
          if constexpr (!BF) al[k] = *reinterpret_cast<const h8*>(da + k * G::DCOPY + 32 * RT * PITCH);
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {   // ky: X row t + ky - 1 of the plane
          const unsigned char* xb = Xs + fb + ((t + a - 1) & 3) * XROW;
          h8 bq[NJ][2];   // [channel tile][hi | lo]
#pragma unroll
          for (int jj = 0; jj < NJ; jj++) {
            bq[jj][0] = *reinterpret_cast<const h8*>(xb + jj * 16 * PITCH);
            if constexpr (!BF) bq[jj][1] = *reinterpret_cast<const h8*>(xb + 32 * PITCH + jj * 16 * PITCH);
          }
#pragma unroll
          for (int k = 0; k < 3; k++)
#pragma unroll
            for (int jj = 0; jj < NJ; jj++) {
              if constexpr (BF) {
                acc[a][k][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, ah[k]), __builtin_bit_cast(bf8, bq[jj][0]),
                                                                        acc[a][k][jj], 0, 0, 0);
              } else {
                acc[a][k][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[k], bq[jj][1], acc[a][k][jj], 0, 0, 0);
                acc[a][k][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[k], bq[jj][0], acc[a][k][jj], 0, 0, 0);
                acc[a][k][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[k], bq[jj][0], acc[a][k][jj], 0, 0, 0);
              }
            }
        }
      }
      // (not __syncthreads(): that would drain vmcnt and expose the latency of the loads issued above in every step;
      // they are only needed by write_piece of the next step, where hipcc waits for them itself)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    }
    r0 = plane * p.Y + yb;
  }

  // ---- partial block -> slab[split, segment][tap][n][c]; 16x16 tile: column = lane & 15, row = 4 (lane >> 4) + register
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int tap = (wkz * 3 + a) * 3 + k;
      float* out = p.slab + ((long long)(split * p.nseg + seg) * 27 + tap) * p.Npad * p.Cpad;   // x segments are splits too
#pragma unroll
      for (int jj = 0; jj < NJ; jj++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int n = nb * (32 * RT) + wi * 16 + 4 * (lane >> 4) + r;
          const int c = cb * 32 + (wj * NJ + jj) * 16 + (lane & 15);
          out[(long long)n * p.Cpad + c] = acc[a][k][jj][r] * acc_sign;
        }
    }
}

// amax[i] = max over the 64 hashed slots of slice i, for the slices whose maxima another kernel exported
// (x: sr3d_hconv.hip forward; dY: the activation-backward kernels) -- instead of a sweep of the tensor
__global__ __launch_bounds__(64) void hw_gather_amax_kernel(const unsigned* xs, int nx, const unsigned* ds, int nd, unsigned* amax) {
  const int lane = threadIdx.x;
  for (int i = 0; i < nx + nd; i++) {
    const unsigned* src = i < nx ? (xs ? xs + i * 64 : nullptr) : (ds ? ds + (i - nx) * 64 : nullptr);
    if (src == nullptr) continue;
    unsigned m = src[lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned t = __shfl_xor(m, o, 64);
      m = t > m ? t : m;
    }
    if (lane == 0) amax[i < nx ? i : 4 + (i - nx)] = m;
  }
}

struct HwSliceMap {
  int xcb[SR3D_MAX_SRC], dcb[SR3D_MAX_SRC];   // first channel / row of every slice (INT_MAX: unused)
};

// dW[n][c][tap] = 2^-(sx(c)+sd(n)) * sum_s slab[s][tap][n][c]; one thread per output, c fastest (coalesced slab reads)
__global__ __launch_bounds__(256) void hwgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int N,
                                                            int cu, int ldc, int Npad, int Cpad, const float* amax,
                                                            const HwSliceMap sm) {
  const long long plane = (long long)Npad * Cpad;
  const long long total = (long long)N * cu * 27;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % cu);
    const long long r = e / cu;
    const int n = (int)(r % N), tap = (int)(r / N);
    const int xi = (c >= sm.xcb[1]) + (c >= sm.xcb[2]) + (c >= sm.xcb[3]), di = (n >= sm.dcb[1]) + (n >= sm.dcb[2]) + (n >= sm.dcb[3]);
    const float mult = amax ? ldexpf(1.f, -(scale_exp_of(amax[xi]) + scale_exp_of(amax[4 + di]))) : 1.f;   // (bf16: unscaled)
    const float* s0 = slab + (long long)tap * plane + (long long)n * Cpad + c;
    float s = 0.f;
    int k = 0;
    for (; k + 4 <= S; k += 4) {   // fixed order: deterministic
      const float v0 = s0[(long long)k * 27 * plane], v1 = s0[(long long)(k + 1) * 27 * plane];
      const float v2 = s0[(long long)(k + 2) * 27 * plane], v3 = s0[(long long)(k + 3) * 27 * plane];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; k < S; k++) s += s0[(long long)k * 27 * plane];
    dw[((long long)n * ldc + c) * 27 + tap] = s * mult;
  }
}

struct HwPlan {
  int rt, nnb, ncb, nseg, S, Npad, Cpad;
  long long rows_per_split;
};

HwPlan hw_plan(const sr3d_conv_desc_t* d, int n_total, int c_used) {
  HwPlan g;
  g.rt = n_total > 32 ? 2 : 1;
  g.nnb = ceil_div(n_total, 32 * g.rt), g.ncb = ceil_div(c_used, 32), g.nseg = ceil_div(d->X, 32);
  g.Npad = g.nnb * 32 * g.rt, g.Cpad = g.ncb * 32;
  const long long rows = (long long)d->B * d->Z * d->Y;
  const long long cols = (long long)g.nnb * g.ncb * g.nseg;
  // 2 .. 6 rounds over the 256 CUs (one workgroup per CU), the count that leaves the last round fullest; at least 24
  // rows per split (4 warm-up steps per plane segment)
  long long S = 1;
  double best = -1.0;
  const long long smax = rows / 24 > 0 ? rows / 24 : 1;
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

template <bool BF, int PF>
int hw_launch(int rt, long long nwg, const HwParams& p, hipStream_t st) {
  constexpr size_t l2 = WGeo<2, BF>::LDS, l1 = WGeo<1, BF>::LDS;
  static SrPerDevice setup;   // (the attribute is per device, not per thread)
  if (int rc = setup.once([&]() -> int {
        SR3D_HIP(hipFuncSetAttribute((const void*)hwgrad_kernel<2, BF, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2));
        SR3D_HIP(hipFuncSetAttribute((const void*)hwgrad_kernel<1, BF, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1));
        return SR3D_OK;
      }))
    return rc;
  if (rt == 2)
    hipLaunchKernelGGL((hwgrad_kernel<2, BF, PF>), dim3((unsigned)nwg), dim3(WNT), l2, st, p);
  else
    hipLaunchKernelGGL((hwgrad_kernel<1, BF, PF>), dim3((unsigned)nwg), dim3(WNT), l1, st, p);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

}  // namespace

int sr3d_gather_absmax(const unsigned* x_absmax, int nx, const unsigned* dy_absmax, int nd, unsigned* amax, hipStream_t st) {
  hipLaunchKernelGGL(hw_gather_amax_kernel, dim3(1), dim3(64), 0, st, x_absmax, nx, dy_absmax, nd, amax);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

// slab + 256 bytes for the two maxima
size_t sr3d_hwgrad_ws_bytes(const sr3d_conv_desc_t* d, int n_total, int c_used) {
  const HwPlan g = hw_plan(d, n_total, c_used);
  return 256 + (size_t)g.S * g.nseg * 27 * g.Npad * g.Cpad * 4;
}

bool sr3d_hwgrad_ok(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy) {
  if (d->stride != 1 || d->X % 8 != 0) return false;
  for (int i = 0; i < x.n; i++)
    if (reinterpret_cast<uintptr_t>(x.ptr[i]) & 15) return false;
  for (int i = 0; i < dy.n; i++)
    if (reinterpret_cast<uintptr_t>(dy.ptr[i]) & 15) return false;
  return true;
}

int sr3d_hwgrad(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy, int n_total, int c_used, float* dw, float* ws,
                hipStream_t st, const unsigned* x_absmax, const unsigned* dy_absmax) {
  const HwPlan g = hw_plan(d, n_total, c_used);
  const bool bf = d->dtype == SR3D_DTYPE_BF16;
  unsigned* amax = (unsigned*)ws;
  const long long vox = (long long)d->Z * d->Y * d->X;
  if (!bf) {   // (bf16 operands are not scaled: no maxima pass)
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
  HwParams p{};
  p.x = x, p.dy = dy, p.cu = c_used, p.N = n_total;
  p.B = d->B, p.Z = d->Z, p.Y = d->Y, p.X = d->X;
  p.nnb = g.nnb, p.ncb = g.ncb, p.nseg = g.nseg, p.S = g.S, p.rows_per_split = g.rows_per_split;
  p.Npad = g.Npad, p.Cpad = g.Cpad;
  p.slab = ws + 64, p.amax = (const float*)amax;
  // (measured on the level-0/1 layers: 1-5 % faster in the bf16 form; in the split form +-3 % per layer in the first half of
  //  round 3, 1 % faster after the kernel's vector work was cut -- profiles/r03f_ab_hwgrad_xcd_order_fp32.log)
  p.xcd_order = getenv("SR3D_HWGRAD_XCD") ? atoi(getenv("SR3D_HWGRAD_XCD")) : 1;
  const long long nwg = (long long)g.nnb * g.ncb * g.nseg * g.S;
  SR3D_CHECK(nwg < (1ll << 31), SR3D_E_ARG, "split-f16 weight gradient: grid too large");
  // (a register ring that issued the loads one or two steps further ahead -- template parameter PF -- measured within 1 % in
  // both forms, profiles/r03a_ab_hwgrad_prefetch_distance_xcd_order.log: the kernel is not latency-bound; only PF = 0 is built)
  (void)vox;
  {
    SrProfScope prof(SR3D_PROF_WGRAD, 2.0 * 27 * c_used * (double)n_total * (double)d->Z * d->Y * d->X * d->B, st);
    int rc;
    rc = bf ? hw_launch<true, 0>(g.rt, nwg, p, st) : hw_launch<false, 0>(g.rt, nwg, p, st);
    if (rc) return rc;
  }
  SrProfScope prof(SR3D_PROF_PACK, 4.0 * ((double)g.S * g.nseg + 1) * 27 * g.Npad * g.Cpad, st);
  const long long total = (long long)n_total * c_used * 27;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  HwSliceMap sm;
  for (int i = 0; i < SR3D_MAX_SRC; i++) sm.xcb[i] = x.cbeg[i], sm.dcb[i] = dy.cbeg[i];
  hipLaunchKernelGGL(hwgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)p.slab, dw, g.S * g.nseg, n_total, c_used,
                     d->Cin, g.Npad, g.Cpad, bf ? (const float*)nullptr : (const float*)amax, sm);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}
