// Weight gradient of the stride-1 3x3x3 convolutions in the Winograd F(2x2, 3x3) domain (fp32 MFMA).
//
//   dU[kz][xi][n][c] = sum over 2x2 output tiles  dM[xi][n][tile] * V[kz][xi][c][tile]
//   dM = A dY_tile A^T  (2x2 -> 4x4),   V = B^T d_tile B  (4x4 input patch of input plane z + kz - 1)
//   dW[n][c][kz] = G^T dU[kz] G         (4x4 -> 3x3, done by the reduce kernel after the split-K sum)
//
// 16 products per tile, channel pair and kz instead of 36: 2.25x fewer MFMA issues than the direct kernel
// (sr3d_wgrad.hip), same fp32 arithmetic.
//
// One 512-thread workgroup (the only one on its CU: 12 accumulator tiles per wave) owns a 64(n) x 32(c) block of
// dU for all 3 x 16 (kz, xi): wave w keeps xi = 2w, 2w+1 for the three kz and both 32-row tiles.  The reduction
// walks columns of the grid (8 tiles = 2 x 16 voxels of one plane, z fastest) as a stream of PLANE STEPS:
// step s brings ONE new input plane into the Winograd domain (V slot s & 3; the other two planes of the 3-plane
// window were transformed by steps s-1, s-2) and the dY plane of the output that step s completes (dM buffer
// s & 1).  The step is software-pipelined so that the single workgroup of the CU never idles its MFMA pipe:
//     iteration s:  dM(s) from registers  |  global loads of step s+1 (x plane, dY)  |
//                   MFMAs of step s-1's output, the V transform of step s scheduled between them  |
//                   raw x rows of step s+1 -> LDS  |  ONE barrier
// All LDS layouts are bank-conflict-free by construction (see the pitch constants), the global base pointers are
// wave-uniform, everything depending only on the lane is computed once per column.
#include <algorithm>
#include "sr3d_common.h"

namespace {

constexpr int GT = 8;                       // Winograd tiles per strip: 1 x 8 tiles = 2 x 16 voxels (SS = 0), or
                                            // 2 x 4 tiles = 4 x 8 voxels (SS = 1: 40- and 20-wide grids pad less)
template <int SS>
struct StripGeo {
  static constexpr int VY = SS ? 4 : 2, VX = SS ? 8 : 16;   // voxels per strip
  static constexpr int RW = VX + 2, RR = VY + 2;            // raw patch: RR rows x RW columns (4 x 18 / 6 x 10)
  static constexpr int TC = SS ? 4 : 8;                     // tile columns
};
constexpr int GXP = 74;                     // raw X pitch per channel (>= 72, even -> 8-byte aligned patch rows,
                                            // 2 * 37: the 32 channels of a b64 read hit 32 different bank pairs)
constexpr int GNB = 64;                     // rows (output channels) per workgroup
constexpr int GVS = 16 * GT * 32;           // one V plane slot: [xi][tile][32 c] floats
// dM buffer: [xi] pitch 545 : [k-step (tile pair)] pitch 136 : [row tile] 64 : [tile parity] 32 : [n & 31].
// A fragment = 64 consecutive floats; the pitches spread the transform's writes (4 rows x 8 tiles x 2 dY rows per
// wave instruction) over all 64 banks.
constexpr int GMK = 136, GMX = 4 * GMK + 1, GMB = 16 * GMX;
constexpr int GXB = 32 * GXP;               // one raw X buffer
constexpr int kGLdsFloats = 4 * GVS + 2 * GMB + 3 * GXB;
constexpr size_t kGLds = (size_t)kGLdsFloats * 4;
static_assert(kGLds <= 160 * 1024, "LDS budget");

typedef const __attribute__((address_space(1))) float* gfloat_p;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) f32x2* gfloat2_p;

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

__device__ __forceinline__ float dpp_xor8(float v) {   // value of lane ^ 8 (row_ror:8 inside a row of 16 lanes)
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));
}

template <int SS>
__global__ __launch_bounds__(512, 2) void wino_wgrad_kernel(const WinoWgradParams p) {
  using SG = StripGeo<SS>;
  constexpr int GXW = SG::RW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Vs = lds;                    // 4 slots [xi][tile][c]
  float* Ms = lds + 4 * GVS;          // 2 buffers
  float* Xr = Ms + 2 * GMB;           // 3 buffers of raw input rows [c][4 rows][18]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blocks of one split are neighbours in launch order: they walk the same columns at the same time, so the x rows
  // (shared by all row blocks) and dY rows (shared by all channel blocks) are served by L2 / MALL
  const int nblk_ = p.Npad / GNB, cblk_ = p.Cpad / 32;
  // ... and of one XCD: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2, so the
  // logical index is made contiguous per XCD (ids congruent mod 8 -> one run of nwg / 8 logical blocks)
  int v;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  }
  const int split = v / (nblk_ * cblk_), q_ = v % (nblk_ * cblk_);
  const int cb = q_ / nblk_, nb = q_ % nblk_;
  const long long ZYX = (long long)p.Z * p.Y * p.X;
  const int YX = p.Y * p.X;

  f32x16 acc[12];   // [kz][xl][nt]
#pragma unroll
  for (int i = 0; i < 12; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;

  // ---- staging maps.
  // X: wave w stages channels w, w+8, w+16, w+24 of the block; a channel's plane piece is 4 rows x 18 cols = 72
  //    elements: lanes 0..63 take element `lane`, lanes 0..7 also element 64 + lane.
  // dY: wave w feeds rows 8w .. 8w+7 as two groups of 4; lane = (row in group) * 16 + (dY row of the tile) * 8 + tile:
  //    one float2 per lane and group, the other row of the 2x2 block comes from lane ^ 8.
  // All global reads are raw BUFFER loads: the descriptor (scalar registers) covers exactly one channel plane
  // (x) / the group's rows of the sample (dY), so padding, ragged tiles, planes -1 and Z and channels beyond
  // Cin / N need no masks at all: their byte offset is out of range and the hardware returns 0.
  constexpr int kRawE = SG::RR * GXW;   // 72 / 60 elements per channel: piece 0 = lanes 0..63, piece 1 the rest
  const int xr0 = lane / GXW, xc0 = lane - xr0 * GXW;
  const int xr1 = (64 + lane) / GXW, xc1 = (64 + lane) - xr1 * GXW;
  const bool x1on = 64 + lane < kRawE;
  const int dn4 = lane >> 4, dr = (lane >> 3) & 1, dtl = lane & 7;
  // Base pointers of the current sample: wave-uniform (scalar registers), rebuilt when the batch index changes.
  // The 4 dY rows of a group sit in the same slice (slice widths are multiples of 4: host dispatch), so row i is
  // the group pointer + i * ZYX: part of the per-lane byte offset.
  gfloat_p xcur[4], dcur[2];
  auto set_batch = [&](const int b) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int gc = cb * 32 + wave + 8 * k;
      xcur[k] = nullptr;
      if (gc < p.Cin) {
        const int si = cat_find(p.x, gc);
        xcur[k] = (gfloat_p)cat_ptr(p.x, si) + (long long)(gc - cat_cbeg(p.x, si)) * ZYX + (long long)b * cat_bstride(p.x, si);
      }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int gn = nb * GNB + wave * 8 + 4 * k;
      dcur[k] = nullptr;
      if (gn < p.N) {
        const int si = cat_find(p.dy, gn);
        dcur[k] = (gfloat_p)cat_ptr(p.dy, si) + (long long)(gn - cat_cbeg(p.dy, si)) * ZYX + (long long)b * cat_bstride(p.dy, si);
      }
    }
  };
  const unsigned kOob = 0xffffffffu;
  const int plane_bytes = YX * 4;
  const int dy_bytes = (int)(ZYX * 16);   // 4 rows of the sample (host dispatch guarantees ZYX * 16 < 2^31)

  // ---- step generator (scalar state).  A segment = consecutive output planes oz_lo..oz_hi of one column; its
  // steps bring planes oz_lo-1 .. oz_hi+1; the step of plane u completes output u-1 when u >= oz_lo+1.
  const long long t_begin = (long long)split * p.per_split;
  long long t_end = t_begin + p.per_split;
  if (t_end > p.ntiles) t_end = p.ntiles;
  long long t_next = t_begin;
  int g_oz, g_tix, g_tiy, g_b;
  {
    long long r = t_begin;
    g_oz = (int)(r % p.Z);
    r /= p.Z;
    g_tix = (int)(r % p.ntx);
    r /= p.ntx;
    g_tiy = (int)(r % p.nty);
    g_b = (int)(r / p.nty);
  }
  int s_u = 0, s_hi = -1, s_lo1 = 0, s_b = -1;
  unsigned xoff0 = kOob, xoff1 = kOob, doffc = kOob;   // per-lane byte offsets inside the plane / the dY rows
  bool n_valid = false, n_out = false;                 // descriptor of the step produced by next_step()
  auto next_step = [&]() {
    if (s_u < s_hi) {
      s_u++;
      n_valid = true;
    } else if (t_next < t_end) {
      const long long left = t_end - t_next;
      const int len = (long long)(p.Z - g_oz) < left ? p.Z - g_oz : (int)left;
      s_u = g_oz - 1, s_hi = g_oz + len, s_lo1 = g_oz + 1;
      t_next += len;
      if (g_b != s_b) set_batch(g_b), s_b = g_b;
      const int y0 = g_tiy * SG::VY, x0 = g_tix * SG::VX;
      g_oz = 0;
      if (++g_tix == p.ntx) {
        g_tix = 0;
        if (++g_tiy == p.nty) g_tiy = 0, ++g_b;
      }
      const int gy0 = y0 - 1 + xr0, gx0 = x0 - 1 + xc0, gy1 = y0 - 1 + xr1, gx1 = x0 - 1 + xc1;
      const bool ok0 = lane < kRawE && (unsigned)gy0 < (unsigned)p.Y && (unsigned)gx0 < (unsigned)p.X;
      const bool ok1 = x1on && (unsigned)gy1 < (unsigned)p.Y && (unsigned)gx1 < (unsigned)p.X;
      xoff0 = ok0 ? (unsigned)(gy0 * p.X + gx0) * 4u : kOob;
      xoff1 = ok1 ? (unsigned)(gy1 * p.X + gx1) * 4u : kOob;
      const int gy = y0 + 2 * (dtl / SG::TC) + dr, gx = x0 + 2 * (dtl % SG::TC);
      const bool dok = gy < p.Y && gx < p.X;   // X is even (host dispatch): gx + 1 is valid with gx
      doffc = dok ? (unsigned)(dn4 * (int)ZYX + gy * p.X + gx) * 4u : kOob;
      n_valid = true;
    } else {
      n_valid = false;
    }
    n_out = n_valid && s_u >= s_lo1;
  };

  f32x2 pd[2];
  typedef __attribute__((address_space(3))) void* lds_p;
  // raw rows of input plane s_u of the current column -> LDS buffer `buf`, straight from the buffer load (no
  // registers, no ds_write): lane i of the wave instruction lands at row + 4 i bytes, which is the row layout.
  auto load_x = [&](float* buf) {
    const bool zok = n_valid && (unsigned)s_u < (unsigned)p.Z;
    const unsigned zo = zok ? (unsigned)s_u * (unsigned)YX : 0u;   // 32-bit: Z*Y*X < 2^27 (host dispatch)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(xcur[k] + zo), 0, (zok && xcur[k] != nullptr) ? plane_bytes : 0, 0x00020000);
      float* row = buf + (wave + 8 * k) * GXP;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_p)row, 4, xoff0, 0, 0, 0);
      if (kRawE > 64 && x1on) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_p)(row + 64), 4, xoff1, 0, 0, 0);
    }
  };
  // dY rows of the output completed by the step BEFORE the generator's current one (the generator runs two steps
  // ahead for x, dY is fetched one step ahead): lagged copy of its state
  int lag_u = 0;
  bool lag_out = false;
  unsigned lag_doffc = kOob;
  gfloat_p lag_d[2] = {nullptr, nullptr};
  auto lag_state = [&]() { lag_u = s_u, lag_out = n_out, lag_doffc = doffc, lag_d[0] = dcur[0], lag_d[1] = dcur[1]; };
  auto load_dy = [&]() {
    const unsigned zo = lag_out ? (unsigned)(lag_u - 1) * (unsigned)YX : 0u;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(lag_d[k] + zo), 0, (lag_out && lag_d[k] != nullptr) ? dy_bytes : 0, 0x00020000);
      pd[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, lag_doffc, 0, 0));
    }
  };
  // dM = A dY A^T with A = [1 0; 1 1; 1 -1; 0 -1].  With a = T(own row), b = T(other row), T(v) = (x, x+y, x-y, -y):
  //   dY row 0 lanes write xi_y = 0: a,   xi_y = 1: a + b;     dY row 1 lanes write xi_y = 3: -a,   xi_y = 2: b - a.
  const float msign = dr ? -1.f : 1.f;
  const int m_p = (dr ? 12 : 0) * GMX + (dtl >> 1) * GMK + (wave >> 2) * 64 + (dtl & 1) * 32 + ((wave * 8 + dn4) & 31);
  const int m_q = (dr ? 8 : 4) * GMX + (dtl >> 1) * GMK + (wave >> 2) * 64 + (dtl & 1) * 32 + ((wave * 8 + dn4) & 31);
  auto transform_m = [&](float* M) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const float rx = pd[k].x, ry = pd[k].y;
      const float tx = dpp_xor8(rx), ty = dpp_xor8(ry);   // the other dY row of the 2x2 block
      const float ox = rx * msign, oy = ry * msign;       // sigma * own row
      const float a[4] = {ox, ox + oy, ox - oy, -oy};
      const float b[4] = {tx, tx + ty, tx - ty, -ty};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        M[m_p + 4 * k + j * GMX] = a[j];
        M[m_q + 4 * k + j * GMX] = b[j] + a[j];
      }
    }
  };
  // V = B^T d B of the (c, tile) patch, half of the columns per thread: waves 0-3 produce xi_x = 0, 1 (from patch
  // columns 0, 1, 2), waves 4-7 xi_x = 2, 3 (from columns 2, 3 and 1); same code, different offsets and one sign.
  const int vh = wave >> 2;
  const int tc = tid & 31, ttl = (tid >> 5) & 7;
  const int v_rd0 = tc * GXP + 2 * (ttl / SG::TC) * GXW + 2 * (ttl % SG::TC);   // patch origin of tile ttl
  const int v_rd2 = v_rd0 + (vh ? 2 : 0), v_rd1 = v_rd0 + (vh ? 1 : 2);
  const int v_wr = (2 * vh) * (GT * 32) + ttl * 32 + tc;
  const float vsign = vh ? -1.f : 1.f;
  float vt[4][2];
  auto tv_read = [&](const float* raw) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const f32x2 pq = *reinterpret_cast<const f32x2*>(raw + v_rd2 + i * GXW);
      const float r = raw[v_rd1 + i * GXW];
      vt[i][0] = pq.x - r;
      vt[i][1] = fmaf(pq.y, vsign, r);
    }
  };
  auto tv_write = [&](float* V) {
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      V[v_wr + (0 * 4 + jj) * (GT * 32)] = vt[0][jj] - vt[2][jj];
      V[v_wr + (1 * 4 + jj) * (GT * 32)] = vt[1][jj] + vt[2][jj];
      V[v_wr + (2 * 4 + jj) * (GT * 32)] = vt[2][jj] - vt[1][jj];
      V[v_wr + (3 * 4 + jj) * (GT * 32)] = vt[1][jj] - vt[3][jj];
    }
  };

  // ---- prologue: LDS starts finite (the first MFMAs of a segment multiply stale V planes by a zero dM), raw rows
  // of step 0
  for (int i = tid; i < 4 * GVS + 2 * GMB; i += 512) lds[i] = 0.f;
  next_step();
  bool cur_valid = n_valid, prev_out = false, cur_out = n_out;
  load_x(Xr);                      // step 0
  next_step();
  bool nx_valid = n_valid, nx_out = n_out;
  load_x(Xr + GXB);                // step 1
  pd[0] = pd[1] = f32x2{0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- main loop: ONE straight-line block per plane step.  Every stage always runs; a stage without work sees
  // zeros (empty buffer descriptors) instead of being branched around, so that the dM transform, the descriptor
  // arithmetic, the global loads and the V transform sit in the shadow of the MFMAs.  Fragments are double-
  // buffered in registers: a wave that is alone on its SIMD still issues MFMAs back to back.
  // Global loads run two steps ahead for x (three raw LDS buffers) and one full step ahead for dY.
  const int fl = lane;   // fragments are 64 consecutive floats
  int rb = 0;            // raw buffer of step s
  // The last 12 MFMAs of a step are issued AFTER its barrier, at the top of the next iteration: they only read
  // registers, and the matrix pipe works through them while the wave does the scalar bookkeeping and issues the
  // loads of the next step (which would otherwise run with an idle pipe).  Zero fragments for the first iteration.
  float a[2][2][2], bq[2][2][3];
#pragma unroll
  for (int xl = 0; xl < 2; xl++) {
    a[1][xl][0] = a[1][xl][1] = 0.f;
    bq[1][xl][0] = bq[1][xl][1] = bq[1][xl][2] = 0.f;
  }
  auto mfmas = [&](const int set) {
#pragma unroll
    for (int kz = 0; kz < 3; kz++)
#pragma unroll
      for (int xl = 0; xl < 2; xl++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
          acc[(kz * 2 + xl) * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
              a[set][xl][nt], bq[set][xl][kz], acc[(kz * 2 + xl) * 2 + nt], 0, 0, 0);
  };
  for (int s = 0; cur_valid || prev_out; s++) {
    mfmas(1);                     // fourth k-step of the previous iteration
    const int rb2 = rb >= 1 ? rb - 1 : 2;   // (rb + 2) % 3
    lag_state();                  // step s+1
    next_step();                  // -> step s+2 (scalar; per-lane offsets only when a new column starts)
    load_x(Xr + rb2 * GXB);       // its rows: global -> LDS, two steps of latency budget
    const float* v0p = Vs + ((s + 1) & 3) * GVS + fl;
    const float* v1p = Vs + ((s + 2) & 3) * GVS + fl;
    const float* v2p = Vs + ((s + 3) & 3) * GVS + fl;
    const float* mp = Ms + ((s + 1) & 1) * GMB + fl;
    auto frags = [&](const int ks, const int set) {
#pragma unroll
      for (int xl = 0; xl < 2; xl++) {
        const int xi = 2 * wave + xl;
        a[set][xl][0] = mp[xi * GMX + ks * GMK];
        a[set][xl][1] = mp[xi * GMX + ks * GMK + 64];
        const int o = (xi * GT + 2 * ks) * 32;
        bq[set][xl][0] = v0p[o], bq[set][xl][1] = v1p[o], bq[set][xl][2] = v2p[o];
      }
    };
    frags(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    frags(1, 1);
    tv_read(Xr + rb * GXB);
    mfmas(0);                             // output completed by step s-1 (zero dM if there is none)
    tv_write(Vs + (s & 3) * GVS);         // V of step s's plane
    __builtin_amdgcn_sched_barrier(0);
    frags(2, 0);
    mfmas(1);
    transform_m(Ms + (s & 1) * GMB);      // dM of step s's output (dY rows loaded during step s-1)
    __builtin_amdgcn_sched_barrier(0);
    load_dy();                            // dY rows of step s+1's output: a full step of latency budget
    frags(3, 1);
    mfmas(0);
    __builtin_amdgcn_sched_barrier(0);
    // the rows of step s+1 (issued one step ago) have landed; younger loads (12 per wave) may stay in flight
    // (a bare s_barrier: __syncthreads() carries a release fence that would drain the younger LDS-DMA loads too)
    if constexpr (kRawE > 64)
      asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else   // one x DMA per channel: 4 + 2 + 2 younger loads
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    prev_out = cur_out;
    cur_valid = nx_valid, cur_out = nx_out;
    nx_valid = n_valid, nx_out = n_out;
    rb = rb == 2 ? 0 : rb + 1;
  }
  mfmas(1);   // fourth k-step of the last iteration

  // ---- partial dU block -> slab[split][kz*16 + xi][n][c]
  const int c = cb * 32 + (lane & 31);
#pragma unroll
  for (int kz = 0; kz < 3; kz++)
#pragma unroll
    for (int xl = 0; xl < 2; xl++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        const int q = kz * 16 + 2 * wave + xl;
        float* dst = p.slab + (((long long)split * 48 + q) * p.Npad + nb * GNB + nt * 32) * p.Cpad + c;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          dst[(long long)n * p.Cpad] = acc[(kz * 2 + xl) * 2 + nt][r];
        }
      }
}

// slab[0][e] = sum_s slab[s][e]  (fixed order: deterministic; in place: a thread only touches its own elements)
__global__ __launch_bounds__(256) void wino_wgrad_sum_kernel(float* __restrict__ slab, int S, long long count) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < count;
       e += (long long)gridDim.x * blockDim.x) {
    const float* src = slab + e;
    float s = src[0];
    int k = 1;
    for (; k + 4 <= S; k += 4) {
      const float v0 = src[(long long)k * count], v1 = src[(long long)(k + 1) * count];
      const float v2 = src[(long long)(k + 2) * count], v3 = src[(long long)(k + 3) * count];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; k < S; k++) s += src[(long long)k * count];
    slab[e] = s;
  }
}

// dW[n][c][kz][ky][kx] = sum_{xi} G[xi_y][ky] G[xi_x][kx] * (sum_s slab[s][kz*16 + xi][n][c])
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                               int S, int N, int Cin, int ldc, int Npad, int Cpad) {
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
    float* out = dw + ((long long)n * ldc + c) * 27 + kz * 9;
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

// strip shape: 2 x 16 voxels, or 4 x 8 where that pads the (y, x) plane less
int strip_shape(const sr3d_conv_desc_t* d) {
  const long long pad0 = (long long)ceil_div(d->Y, 2) * 2 * ceil_div(d->X, 16) * 16;
  const long long pad1 = (long long)ceil_div(d->Y, 4) * 4 * ceil_div(d->X, 8) * 8;
  return pad1 < pad0 ? 1 : 0;
}

GPlan gplan(const sr3d_conv_desc_t* d, int n_total, int c_used) {
  GPlan g;
  const int ss = strip_shape(d);
  g.nblk = ceil_div(n_total, GNB), g.cblk = ceil_div(c_used, 32);
  g.Npad = g.nblk * GNB, g.Cpad = g.cblk * 32;
  g.nty = ceil_div(d->Y, ss ? 4 : 2), g.ntx = ceil_div(d->X, ss ? 8 : 16);
  g.ntiles = (long long)d->B * g.nty * g.ntx * d->Z;
  // One workgroup per CU and every workgroup does the same amount of work: make the grid a whole number of rounds
  // over the 256 CUs (5 rounds if the slab cap allows it), otherwise as many splits as the cap allows.
  const int R = g.nblk * g.cblk;
  const long long slab_one = (long long)48 * g.Npad * g.Cpad * 4;
  const long long cap = std::max(1ll, (384ll << 20) / slab_one);
  long long want = 1;
  for (int rounds = 5; rounds >= 1; rounds--) {
    want = std::max(1ll, (256ll * rounds) / R);
    if (want <= cap) break;
  }
  if (want > cap) want = cap;
  if (want > g.ntiles) want = g.ntiles;
  g.per_split = (g.ntiles + want - 1) / want;
  g.S = (int)((g.ntiles + g.per_split - 1) / g.per_split);
  return g;
}

}  // namespace

size_t sr3d_wino_wgrad_ws_bytes(const sr3d_conv_desc_t* d, int n_total, int c_used) {
  const GPlan g = gplan(d, n_total, c_used);
  return (size_t)g.S * 48 * g.Npad * g.Cpad * 4;
}

int sr3d_wino_wgrad(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy, int n_total, int c_used, float* dw,
                    float* ws, hipStream_t st) {
  const GPlan g = gplan(d, n_total, c_used);
  WinoWgradParams p{};
  p.x = x, p.dy = dy, p.Cin = c_used, p.N = n_total;
  p.Z = d->Z, p.Y = d->Y, p.X = d->X;
  p.nty = g.nty, p.ntx = g.ntx, p.ntiles = g.ntiles, p.per_split = g.per_split;
  p.slab = ws, p.Npad = g.Npad, p.Cpad = g.Cpad;
  SR3D_CHECK(g.cblk <= 65535 && g.nblk <= 65535, SR3D_E_ARG, "winograd wgrad: too many blocks");
  static SrPerDevice setup;   // (the attribute is per device, not per thread)
  if (int rc = setup.once([&]() -> int {
        SR3D_HIP(hipFuncSetAttribute((const void*)wino_wgrad_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGLds));
        SR3D_HIP(hipFuncSetAttribute((const void*)wino_wgrad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGLds));
        return SR3D_OK;
      }))
    return rc;
  void* tok = nullptr;
  if (sr3d_prof_active())
    sr3d_prof_begin(SR3D_PROF_WGRAD, 2.0 * 27 * c_used * (double)n_total * (double)d->Z * d->Y * d->X * d->B, st, &tok);
  if (strip_shape(d))
    hipLaunchKernelGGL(wino_wgrad_kernel<1>, dim3(g.S * g.cblk * g.nblk), dim3(512), kGLds, st, p);
  else
    hipLaunchKernelGGL(wino_wgrad_kernel<0>, dim3(g.S * g.cblk * g.nblk), dim3(512), kGLds, st, p);
  sr3d_prof_end(tok, st);
  SR3D_HIP(hipGetLastError());
  const long long total = (long long)n_total * c_used * 3;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  SrProfScope prof(SR3D_PROF_PACK, 4.0 * g.S * 48.0 * g.Npad * g.Cpad, st);   // split-K slabs read once
  if (g.S > 1) {
    const long long count = (long long)48 * g.Npad * g.Cpad;
    const int sblocks = (int)std::min<long long>((count + 255) / 256, 16384);
    hipLaunchKernelGGL(wino_wgrad_sum_kernel, dim3(sblocks), dim3(256), 0, st, ws, g.S, count);
    SR3D_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw, 1, n_total,
                     c_used, d->Cin, g.Npad, g.Cpad);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}
