// Weight gradient of the 3x3x3 Conv3d on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   dW[n][c][t] = sum_{b, voxel o} dY[b][n][o] * X[b][c][o*stride + d_t]
//
// GEMM view: rows n (A = dY), cols c for one tap (B = X shifted by the tap),
// reduction over voxels (2 per MFMA).  One workgroup owns a 32(n) x 32(c) x 27(tap)
// block of dW in registers -- wave g holds taps 7g..7g+6 -- and walks a
// contiguous range of spatial tiles (z fastest, so consecutive tiles re-read two
// of their three input planes from L2).  The voxel reduction is split over S
// workgroups; partial blocks go to a slab [S][27][Npad][Cpad] that a second,
// fixed-order kernel sums and transposes into PyTorch's (n, c, kz, ky, kx)
// layout, so the result is bit-reproducible (the reference runs with
// use_deterministic=True: utils.py:70-92).
// X and dY are virtual channel concatenations (sr3d_common.h), e.g. dY =
// [d_feat ; d_gate] of a gated layer yields dW = [dWf ; dWg] in one pass.
#include <algorithm>
#include <cstdint>
#include "sr3d_common.h"

#include <stdlib.h>

namespace {

typedef const __attribute__((address_space(1))) float* gfloat_p;

struct WgradParams {
  ChanCat x;
  ChanCat dy;
  int Cin, N, J;       // J = Cin * 27 flattened (channel, tap) columns
  int IZ, IY, IX;
  int OZ, OY, OX;
  int nty, ntx;        // tiles per (y, x); z tiles = OZ
  long long ntiles;    // B * nty * ntx * OZ
  long long per_split; // tiles per workgroup
  float* slab;         // [S][Npad][Jpad]
  int Npad, Jpad;
};

// One 512-thread workgroup per CU: WAVES_N waves split the rows (32 each), 8 / WAVES_N waves split the
// columns (7 tiles of 32 each); two waves per SIMD share one LDS image.
// LDS image of the input: per channel 3 planes x HY rows, every row starts at the 16-byte aligned
// voxel x0*S - 4 and is RW floats long, so it is filled with float4 loads / ds_write_b128.  Plane and
// channel pitches are padded (multiples of 4 floats) so that 32 consecutive (channel, tap) columns hit
// the banks at most 2-way.
template <int S_IN, int TY, int WAVES_N, int CTW_ = 7>
struct WgradCfg {
  static constexpr int WAVES_C = 8 / WAVES_N;
  static constexpr int CTW = CTW_;                      // column tiles per wave
  static constexpr int COLS = WAVES_C * CTW * 32;       // columns per workgroup
  static constexpr int ROWS = WAVES_N * 32;
  static constexpr int NCH = (COLS + 26) / 27 + 1;      // channels a column block can touch
  static constexpr int HY = (TY - 1) * S_IN + 3;
  static constexpr int RW = S_IN == 1 ? 40 : 68;        // floats per row: [x0*S - 4, x0*S - 4 + RW)
  static constexpr int RQ = RW / 4;
  static constexpr int PZ = S_IN == 1 ? (TY == 2 ? 164 : HY * RW + 4) : 204;
  static constexpr int PH = S_IN == 1 ? (TY == 2 ? 500 : 3 * PZ + 8) : 612;
  static constexpr int CHQ = 3 * HY * RQ;               // float4 pieces per channel
  static constexpr int VT = TY * 32;                    // voxels per tile
  static constexpr int PV = VT + 1;
  static constexpr int XS = NCH * PH;
  static constexpr int DS = ROWS * PV;
  static constexpr int TBL = 2 * (NCH + ROWS);  // 64-bit entries: per-channel / per-row base pointer and batch stride
  static constexpr size_t lds_bytes = (size_t)((XS + DS + 3) & ~3) * 4 + (size_t)TBL * 8;
  static_assert(HY * RW <= PZ && 3 * PZ <= PH, "pitches too small");
};

// T: storage type of x and dy (float, or bf16raw with sr3d_conv_desc_t.dtype = bf16: loaded, widened, multiplied in fp32)
template <int S_IN, int TY, int WAVES_N, bool VEC, int CTW_, typename T>
__global__ __launch_bounds__(512, 2) void wgrad_kernel(const WgradParams p) {
  using C = WgradCfg<S_IN, TY, WAVES_N, CTW_>;
  constexpr int HY = C::HY, RW = C::RW, RQ = C::RQ, PZ = C::PZ, PH = C::PH, PV = C::PV, VT = C::VT, NCH = C::NCH;
  constexpr int CHQ = C::CHQ;
  constexpr int CTW = C::CTW, ROWS = C::ROWS, WAVES_C = C::WAVES_C;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;
  float* Ds = lds + C::XS;
  // tables of global pointers (sample 0) and per-sample strides, filled once
  const T** xptr = reinterpret_cast<const T**>(lds + ((C::XS + C::DS + 3) & ~3));
  long long* xbs = reinterpret_cast<long long*>(xptr + NCH);
  const T** dptr = reinterpret_cast<const T**>(xbs + NCH);
  long long* dbs = reinterpret_cast<long long*>(dptr + ROWS);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / WAVES_C, wc = wave % WAVES_C;
  const int split = blockIdx.x, jb = blockIdx.y, nb = blockIdx.z;
  const long long IZYX = (long long)p.IZ * p.IY * p.IX;
  const long long OZYX = (long long)p.OZ * p.OY * p.OX;
  const int j_begin = jb * C::COLS;
  const int c_lo = j_begin / 27;

  if (tid < NCH) {
    const int gc = c_lo + tid;
    const T* ptr = nullptr;
    long long bs = 0;
    if (gc < p.Cin) {
      const int si = cat_find(p.x, gc);
      ptr = reinterpret_cast<const T*>(cat_ptr(p.x, si)) + (long long)(gc - cat_cbeg(p.x, si)) * IZYX;
      bs = cat_bstride(p.x, si);
    }
    xptr[tid] = ptr, xbs[tid] = bs;
  } else if (tid >= 64 && tid < 64 + ROWS) {
    const int r = tid - 64, gn = nb * ROWS + r;
    const T* ptr = nullptr;
    long long bs = 0;
    if (gn < p.N) {
      const int si = cat_find(p.dy, gn);
      ptr = reinterpret_cast<const T*>(cat_ptr(p.dy, si)) + (long long)(gn - cat_cbeg(p.dy, si)) * OZYX;
      bs = cat_bstride(p.dy, si);
    }
    dptr[r] = ptr, dbs[r] = bs;
  }

  f32x16 acc[CTW];
#pragma unroll
  for (int t = 0; t < CTW; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

  // LDS read bases: A = dY[n = lane&31][v + (lane>>5)], B = X[column = lane&31][pos + (lane>>5)*S]
  // where a column is a (channel, tap) pair.  The three input planes of a tile live in three LDS slots
  // that ROTATE from tile to tile (z is the fastest tile index): only the S_IN new planes are loaded per
  // tile, the other 3 - S_IN are re-used in place.  b_base holds everything of a column's offset except
  // its plane, kzsel its kz; the plane offset is added per tile.
  const int a_base = (wn * 32 + (lane & 31)) * PV + (lane >> 5);
  int b_base[CTW], kzsel[CTW];
#pragma unroll
  for (int t = 0; t < CTW; t++) {
    int j = j_begin + (wc * CTW + t) * 32 + (lane & 31);
    j = j < p.J ? j : p.J - 1;  // padded columns read something valid; they are never stored
    const int c = j / 27, tap = j - c * 27;
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    b_base[t] = (c - c_lo) * PH + (lane >> 5) * S_IN + ky * RW + kx + 3;
    kzsel[t] = kz;
  }

  constexpr int NT = 512;
  // Input staging: wave w owns channels w, w+8, ... of the block (channel = wave-uniform, so its global base
  // pointer is scalar arithmetic); one plane of a channel is HY rows x RQ float4 = PLQ pieces <= 64 lanes.
  constexpr int CPW = (NCH + 7) / 8;        // channels per wave
  constexpr int PLQ = HY * RQ;              // float4 pieces per plane and channel
  static_assert(PLQ <= 64, "one plane of one channel must fit one wave instruction");
  constexpr int NEWP = S_IN;                // new planes per tile in steady state
  constexpr int PER = (ROWS * VT) / (4 * NT);  // float4 pieces of the dY tile per thread
  static_assert((ROWS * VT) % (4 * NT) == 0, "dY tile must split into float4 per thread");
  const int l_hy = lane / RQ, l_q = lane - l_hy * RQ;   // this lane's piece of a plane
  const bool l_on = lane < PLQ;
  const int l_rel = l_hy * p.IX + 4 * l_q;              // offset inside a plane (floats)
  const int l_lds = l_hy * RW + 4 * l_q;
  f32x4 vx[NEWP][CPW] = {};
  f32x4 vd[PER] = {};

  const long long t_begin = (long long)split * p.per_split;
  long long t_end = t_begin + p.per_split;
  if (t_end > p.ntiles) t_end = p.ntiles;

  // coordinates of the next tile to prefetch: decomposed once (64-bit divisions are ~200 scalar
  // instructions each), then advanced incrementally (z fastest, then x, y, sample)
  int n_oz, n_tix, n_tiy, n_b;
  {
    long long r = t_begin;
    n_oz = (int)(r % p.OZ);
    r /= p.OZ;
    n_tix = (int)(r % p.ntx);
    r /= p.ntx;
    n_tiy = (int)(r % p.nty);
    n_b = (int)(r / p.nty);
  }
  int c_b = 0, c_oz = 0, c_oy0 = 0, c_ox0 = 0;   // tile whose data is (being) prefetched into vx / vd
  bool c_rowok = false;
  int c_rowoff = 0;

  auto prep_next = [&]() {
    c_oz = n_oz, c_b = n_b;
    c_oy0 = n_tiy * TY, c_ox0 = n_tix * 32;
    if (++n_oz == p.OZ) {
      n_oz = 0;
      if (++n_tix == p.ntx) {
        n_tix = 0;
        if (++n_tiy == p.nty) n_tiy = 0, ++n_b;
      }
    }
    const int gy = c_oy0 * S_IN - 1 + l_hy, xs = c_ox0 * S_IN - 4 + 4 * l_q;
    c_rowok = l_on && (unsigned)gy < (unsigned)p.IY;
    if (VEC) c_rowok = c_rowok && xs >= 0 && xs + 3 < p.IX;
    c_rowoff = (c_oy0 * S_IN - 1) * p.IX + c_ox0 * S_IN - 4 + l_rel;   // + gz * IY * IX
  };

  // issue the loads of input plane `gz` (all channels of this wave) of the prefetch tile into vx[slot]
  auto load_plane = [&](f32x4 (&dst)[CPW], const int gz) {
    const bool zok = (unsigned)gz < (unsigned)p.IZ;
    const int off = gz * p.IY * p.IX + c_rowoff;
#pragma unroll
    for (int k = 0; k < CPW; k++) {
      const int cl = wave + 8 * k;     // wave-uniform
      const int gc = c_lo + cl;
      const T* base = nullptr;
      if (cl < NCH && gc < p.Cin && zok) {
        const int si = cat_find(p.x, gc);
        base = reinterpret_cast<const T*>(cat_ptr(p.x, si)) + ((long long)(gc - cat_cbeg(p.x, si)) * IZYX + (long long)c_b * cat_bstride(p.x, si));
      }
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (base != nullptr && c_rowok) {
        if (VEC) {
          v = ActIo<T>::ld4(base + off);
        } else {
          const int xs = c_ox0 * S_IN - 4 + 4 * l_q;
          if ((unsigned)(xs + 0) < (unsigned)p.IX) v.x = ActIo<T>::ld(base + off + 0);
          if ((unsigned)(xs + 1) < (unsigned)p.IX) v.y = ActIo<T>::ld(base + off + 1);
          if ((unsigned)(xs + 2) < (unsigned)p.IX) v.z = ActIo<T>::ld(base + off + 2);
          if ((unsigned)(xs + 3) < (unsigned)p.IX) v.w = ActIo<T>::ld(base + off + 3);
        }
      }
      dst[k] = v;
    }
  };

  auto store_plane = [&](const f32x4 (&src)[CPW], const int slot) {
#pragma unroll
    for (int k = 0; k < CPW; k++) {
      const int cl = wave + 8 * k;
      if (cl < NCH && l_on) *reinterpret_cast<f32x4*>(&Xs[cl * PH + slot * PZ + l_lds]) = src[k];
    }
  };

  auto load_dy = [&]() {
#pragma unroll
    for (int i = 0; i < PER; i++) {
      const int e = (tid + i * NT) * 4;
      const int n = e / VT, vv = e % VT;
      const int oy = c_oy0 + vv / 32, ox = c_ox0 + (vv & 31);
      const T* base = dptr[n];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (base != nullptr && oy < p.OY) {
        const T* row = base + (long long)c_b * dbs[n] + ((long long)c_oz * p.OY + oy) * p.OX;
        if (VEC) {
          if (ox + 3 < p.OX) v = ActIo<T>::ld4(row + ox);
        } else {
          if (ox + 0 < p.OX) v.x = ActIo<T>::ld(row + ox + 0);
          if (ox + 1 < p.OX) v.y = ActIo<T>::ld(row + ox + 1);
          if (ox + 2 < p.OX) v.z = ActIo<T>::ld(row + ox + 2);
          if (ox + 3 < p.OX) v.w = ActIo<T>::ld(row + ox + 3);
        }
      }
      vd[i] = v;
    }
  };

  __syncthreads();  // pointer tables are visible
  bool fresh = true;   // the tile about to be stored starts a new z column: all three planes are new
  int s0 = 0;          // LDS slot of plane kz = 0 of the current tile
  if (t_begin < t_end) {
    prep_next();
    load_dy();
  }

  constexpr int KS = TY * 16;        // k-steps (2 voxels each) per tile
  constexpr int NG = KS / 2;         // MFMA groups (2 k-steps each)
  static_assert(NG >= NEWP + 1, "not enough MFMA groups to spread the prefetch over");
  constexpr int GSTEP = NG / (NEWP + 1);  // a prefetch piece after every GSTEP-th group

  for (long long tile = t_begin; tile < t_end; tile++) {
    __syncthreads();  // previous tile fully consumed
    if (fresh) {
      // start of a z column (or of this workgroup's range): fill all three slots, synchronously
      s0 = 0;
#pragma unroll
      for (int k0 = 0; k0 < 3; k0 += NEWP) {
#pragma unroll
        for (int q = 0; q < NEWP; q++)
          if (k0 + q < 3) load_plane(vx[q], c_oz * S_IN - 1 + k0 + q);
#pragma unroll
        for (int q = 0; q < NEWP; q++)
          if (k0 + q < 3) store_plane(vx[q], k0 + q);
      }
    } else {
      // steady state: the NEWP new planes (prefetched during the previous tile) replace the oldest ones
      s0 = (s0 + NEWP) % 3;
#pragma unroll
      for (int q = 0; q < NEWP; q++) store_plane(vx[q], (s0 + 3 - NEWP + q) % 3);
    }
    {
#pragma unroll
      for (int i = 0; i < PER; i++) {
        const int e = (tid + i * NT) * 4;
        float* d = &Ds[(e / VT) * PV + (e % VT)];  // PV is odd: scalar stores
        d[0] = vd[i].x, d[1] = vd[i].y, d[2] = vd[i].z, d[3] = vd[i].w;
      }
    }
    __syncthreads();
    const bool more = tile + 1 < t_end;
    if (more) {
      prep_next();
      fresh = c_oz == 0;   // the next tile starts a new column: its planes are loaded at its own store phase
    }

    // per-lane plane offsets of this tile's three slots
    int boff[CTW];
    {
      const int po0 = s0 * PZ, po1 = ((s0 + 1) % 3) * PZ, po2 = ((s0 + 2) % 3) * PZ;
#pragma unroll
      for (int t = 0; t < CTW; t++) boff[t] = b_base[t] + (kzsel[t] == 0 ? po0 : (kzsel[t] == 1 ? po1 : po2));
    }

    // MFMA loop, fragments double-buffered in registers.  The sched_barriers keep the LDS reads of k-step
    // s+1 ABOVE the MFMAs of k-step s (hipcc otherwise sinks them to just before their use).  The prefetch
    // of the next tile is cut into pieces issued between MFMA groups.
    auto frag = [&](int s, float& a, float (&bv)[CTW]) {
      const int row = s >> 4, xx = (s & 15) * 2;
      a = Ds[a_base + row * 32 + xx];
#pragma unroll
      for (int t = 0; t < CTW; t++) bv[t] = Xs[boff[t] + (row * S_IN) * RW + xx * S_IN];
    };
    {
      float a0, a1, b0[CTW], b1[CTW];
      frag(0, a0, b0);
#pragma unroll
      for (int g = 0; g < NG; g++) {
        const int s = 2 * g;
        frag(s + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < CTW; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < KS) frag(s + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < CTW; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more && g % GSTEP == 0) {
          const int piece = g / GSTEP;
          if (piece < NEWP) {
            if (!fresh) load_plane(vx[piece], c_oz * S_IN - 1 + (3 - NEWP) + piece);
          } else if (piece == NEWP) {
            load_dy();
          }
        }
      }
    }
  }

  // ---- partial block -> slab[split][n][j]
#pragma unroll
  for (int t = 0; t < CTW; t++) {
    const int j = j_begin + (wc * CTW + t) * 32 + (lane & 31);
    if (j >= p.Jpad) continue;
    float* dst = p.slab + ((long long)split * p.Npad + nb * ROWS + wn * 32) * p.Jpad + j;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      dst[(long long)n * p.Jpad] = acc[t][r];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient for layers with <= 4 output channels (the model's `last` conv, 69 -> 4): a 32-row MFMA
// tile would be 7/8 padding, so this one runs on the VALU.  One workgroup owns ONE input channel and a
// contiguous range of 4x8x32-voxel tiles; a thread owns one (y, x) column of 4 voxels and keeps all
// 27 taps x 4 rows = 108 partial sums in registers (sliding 3-plane window through the LDS halo tile);
// at the end the 256 threads are reduced (wave shuffles, then LDS) and written to slab[split][c][n][tap].
typedef float f32x2v __attribute__((ext_vector_type(2)));

struct SmallNParams {
  ChanCat x;
  const float* dy;     // (B, N, OZ, OY, OX), N <= 4
  int Cin, N;
  int Z, Y, X;         // stride 1: input grid = output grid
  int ntz, nty, ntx;
  long long ntiles, per_split;
  float* slab;         // [S][Cin][4][27]
};

__global__ __launch_bounds__(256) void wgrad_smalln_kernel(const SmallNParams p) {
  constexpr int TZ = 4, TY = 8, RW = 40, HY = TY + 2, HZ = TZ + 2;
  constexpr int PZ = HY * RW;   // plane pitch (floats)
  __shared__ __attribute__((aligned(16))) float Hs[HZ * PZ];
  __shared__ float red[4 * 108];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = tid & 31, ty = tid >> 5;
  const int split = blockIdx.x, c = blockIdx.y;
  const long long ZYX = (long long)p.Z * p.Y * p.X;
  const int si = cat_find(p.x, c);
  const gfloat_p xbase0 = (gfloat_p)cat_ptr(p.x, si) + (long long)(c - cat_cbeg(p.x, si)) * ZYX;
  const long long xbs = cat_bstride(p.x, si);
  const bool vec = (p.X % 4 == 0) && ((reinterpret_cast<uintptr_t>(cat_ptr(p.x, si)) & 15) == 0);

  f32x2v acc[2][27];   // rows (0,1) and (2,3) as pairs: one v_pk_fma_f32 per pair and tap
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int t = 0; t < 27; t++) acc[n][t] = f32x2v{0.f, 0.f};

  const long long t_begin = (long long)split * p.per_split;
  long long t_end = t_begin + p.per_split;
  if (t_end > p.ntiles) t_end = p.ntiles;

  // Tile loop with a register prefetch: the halo pieces and dY values of tile i+1 are loaded while tile i is
  // computed (without it every tile exposed a full global-memory round trip: 3 waves per SIMD cannot hide 2 us)
  constexpr int NHQ = (HZ * HY * (RW / 4) + 255) / 256;   // float4 halo pieces per thread
  auto decode = [&](long long tile, int& b, int& z0, int& y0, int& x0) {
    long long r = tile;
    const int tiz = (int)(r % p.ntz);
    r /= p.ntz;
    const int tix = (int)(r % p.ntx);
    r /= p.ntx;
    const int tiy = (int)(r % p.nty);
    b = (int)(r / p.nty);
    z0 = tiz * TZ, y0 = tiy * TY, x0 = tix * 32;
  };
  f32x4 hv[NHQ];
  float dv[TZ][4];
  auto prefetch = [&](long long tile) {
    int b, z0, y0, x0;
    decode(tile, b, z0, y0, x0);
    const gfloat_p xb = xbase0 + (long long)b * xbs;
#pragma unroll
    for (int i = 0; i < NHQ; i++) {   // halo tile [6][10][40]: rows start at x0 - 4 (16-byte aligned)
      const int e = tid + 256 * i;
      const int hz = e / (HY * (RW / 4)), r2 = e - hz * (HY * (RW / 4));
      const int hy = r2 / (RW / 4), q = r2 - hy * (RW / 4);
      const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, xs = x0 - 4 + 4 * q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (e < HZ * HY * (RW / 4) && (unsigned)gz < (unsigned)p.Z && (unsigned)gy < (unsigned)p.Y) {
        const gfloat_p row = xb + ((long long)gz * p.Y + gy) * p.X;
        if (vec) {
          if (xs >= 0 && xs + 3 < p.X) v = *(const __attribute__((address_space(1))) f32x4*)(row + xs);
        } else {
          if ((unsigned)(xs + 0) < (unsigned)p.X) v.x = row[xs + 0];
          if ((unsigned)(xs + 1) < (unsigned)p.X) v.y = row[xs + 1];
          if ((unsigned)(xs + 2) < (unsigned)p.X) v.z = row[xs + 2];
          if ((unsigned)(xs + 3) < (unsigned)p.X) v.w = row[xs + 3];
        }
      }
      hv[i] = v;
    }
  };
  auto load_dy = [&](long long tile) {
    int b, z0, y0, x0;
    decode(tile, b, z0, y0, x0);
    const int gy = y0 + ty, gx = x0 + tx;
    const bool inb = gy < p.Y && gx < p.X;
#pragma unroll
    for (int z = 0; z < TZ; z++)
#pragma unroll
      for (int n = 0; n < 4; n++) {
        dv[z][n] = 0.f;
        if (inb && z0 + z < p.Z && n < p.N)
          dv[z][n] = p.dy[(((long long)b * p.N + n) * p.Z + z0 + z) * p.Y * p.X + (long long)gy * p.X + gx];
      }
  };
  if (t_begin < t_end) prefetch(t_begin);
  if (t_begin < t_end) load_dy(t_begin);
  for (long long tile = t_begin; tile < t_end; tile++) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NHQ; i++) {
      const int e = tid + 256 * i;
      if (e < HZ * HY * (RW / 4)) {
        const int hz = e / (HY * (RW / 4)), r2 = e - hz * (HY * (RW / 4));
        const int hy = r2 / (RW / 4), q = r2 - hy * (RW / 4);
        *reinterpret_cast<f32x4*>(&Hs[hz * PZ + hy * RW + 4 * q]) = hv[i];
      }
    }
    float d[TZ][4];   // this tile's dY values; dv is refilled for the next tile
#pragma unroll
    for (int z = 0; z < TZ; z++)
#pragma unroll
      for (int n = 0; n < 4; n++) d[z][n] = dv[z][n];
    __syncthreads();
    if (tile + 1 < t_end) {
      prefetch(tile + 1);
      load_dy(tile + 1);
    }
    const float* hp = &Hs[ty * RW + tx + 3];  // neighbour (kz,ky,kx) of voxel z: hp[(z+kz)*PZ + ky*RW + kx]
#pragma unroll
    for (int z = 0; z < TZ; z++) {
#pragma unroll
      for (int kz = 0; kz < 3; kz++)
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
          for (int kx = 0; kx < 3; kx++) {
            const float xv = hp[(z + kz) * PZ + ky * RW + kx];
            const f32x2v xx = {xv, xv};
            acc[0][(kz * 3 + ky) * 3 + kx] += f32x2v{d[z][0], d[z][1]} * xx;
            acc[1][(kz * 3 + ky) * 3 + kx] += f32x2v{d[z][2], d[z][3]} * xx;
          }
    }
  }

  // reduce the 256 threads: wave64 shuffles, then the 4 wave partials through LDS
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 4; n++)
#pragma unroll
    for (int t = 0; t < 27; t++) {
      float v = acc[n >> 1][t][n & 1];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane == 0) red[wave * 108 + n * 27 + t] = v;
    }
  __syncthreads();
  if (tid < 108)
    p.slab[((long long)split * p.Cin + c) * 108 + tid] = red[tid] + red[108 + tid] + red[216 + tid] + red[324 + tid];
}

// dW[n][c][t] = sum_s slab[s][c][n][t]
__global__ void wgrad_smalln_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int N,
                                           int Cin) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * Cin * 27) return;
  const int n = e / (Cin * 27), r = e - n * Cin * 27, c = r / 27, t = r - c * 27;
  float s = 0.f;
  for (int k = 0; k < S; k++) s += slab[((long long)k * Cin + c) * 108 + n * 27 + t];
  dw[e] = s;
}

struct SmallNPlan {
  int ntz, nty, ntx, S;
  long long ntiles, per_split;
};

SmallNPlan smalln_plan(const sr3d_conv_desc_t* d) {
  SmallNPlan pl;
  pl.ntz = ceil_div(d->Z, 4), pl.nty = ceil_div(d->Y, 8), pl.ntx = ceil_div(d->X, 32);
  pl.ntiles = (long long)d->B * pl.ntz * pl.nty * pl.ntx;
  long long want = ceil_div(6144, d->Cin);  // ~24 workgroups per CU in total
  if (want > pl.ntiles) want = pl.ntiles;
  if (want < 1) want = 1;
  pl.per_split = (pl.ntiles + want - 1) / want;
  pl.S = (int)((pl.ntiles + pl.per_split - 1) / pl.per_split);
  return pl;
}

// Winograd-domain weight gradient (sr3d_wino_wgrad.hip) for the stride-1 layers; SR3D_WINOGRAD_WGRAD=0 selects
// the direct kernel.  The kernel feeds dY rows in groups of 4 with float2 loads: slice widths must be multiples of
// 4, X even and the dY pointers 8-byte aligned; anything else takes the direct kernel.
inline bool wino_wgrad_on() {
  static const bool on = getenv("SR3D_WINOGRAD_WGRAD") ? atoi(getenv("SR3D_WINOGRAD_WGRAD")) != 0 : true;
  return on && sr3d_wino_enabled();
}
inline bool use_wino_wgrad(const sr3d_conv_desc_t* d, int n_total) {
  // (the kernel's buffer descriptors span 4 dY rows of one sample: 16 * Z*Y*X bytes must fit 31 bits)
  return d->dtype != SR3D_DTYPE_BF16 && wino_wgrad_on() && d->stride == 1 && d->Cin >= 16 && n_total > 4 && n_total % 4 == 0 && d->X % 2 == 0 &&
         (long long)d->Z * d->Y * d->X < (1ll << 27);
}
// 1..4 input channels beyond a multiple of 32 (a mask concatenated to the features) would cost a whole 32-channel
// block: they go to the few-channel VALU kernel (sr3d_wgrad_few.hip) instead
inline int wino_wgrad_c_used(const sr3d_conv_desc_t* d) {
  const int rem = d->Cin % 32;
  return (d->Cin >= 32 && rem >= 1 && rem <= 4) ? d->Cin - rem : d->Cin;
}
inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
inline size_t wino_wgrad_total_ws(const sr3d_conv_desc_t* d, int n_total) {
  const int cu = wino_wgrad_c_used(d);
  size_t bytes = align256(sr3d_wino_wgrad_ws_bytes(d, n_total, cu));
  if (cu < d->Cin) bytes += sr3d_wgrad_few_ws_bytes(d, n_total, d->Cin - cu);
  return bytes;
}
// split-f16 weight gradient (sr3d_hwgrad.hip): stride-1 layers on grids that fill the chip (SR3D_SPLIT_F16, see
// sr3d_hconv.hip: 0 off, 1 auto, 2 always)
inline bool use_hwgrad(const sr3d_conv_desc_t* d, int n_total) {
  if (d->dtype == SR3D_DTYPE_BF16) return d->stride == 1 && d->X % 8 == 0;   // bf16 form of the kernel, any size
  const int mode = sr3d_hconv_mode();
  if (mode == 0 || d->stride != 1 || d->Cin < 32 || d->X % 8 != 0) return false;
  if (mode == 2) return true;
  // (the 4-row `last` layer, 69 -> 4 at full resolution: 3.7 ms here -- 4 of the 32 rows of a block used -- against 4.4 ms on the
  //  small-N VALU kernel, profiles/r03p_layers_last_wgrad.log)
  if (n_total < 16) return (long long)d->B * d->Z * d->Y * d->X >= 1000000;
  // measured against the Winograd-domain kernel (profiles/r03w_layers_small_grids_default_vs_forced.log, after the wave roles
  // and the split arithmetic of round 3): faster from U-Net level 3 up (16 k voxels: down3.1 0.80 -> 0.49 ms, up4.convs
  // 0.84 -> 0.55 / 0.48 -> 0.26 ms), equal on level 4
  const long long vox = (long long)d->B * d->Z * d->Y * d->X;
  return vox >= 10000;
}
// ... few input channels (conv0: 5) on its (channel, kx)-column form (sr3d_hwgrad_fc.hip), fp32 storage, grids that fill the chip
inline bool use_hwgrad_fc(const sr3d_conv_desc_t* d, int n_total) {
  if (d->stride != 1 || d->Cin > 5 || n_total < 16 || d->X % 8 != 0) return false;
  if (d->dtype == SR3D_DTYPE_BF16) return getenv("SR3D_NO_FC_BF16") == nullptr;   // (round 4: the bf16 form of the kernel, any size)
  const int mode = sr3d_hconv_mode();
  if (mode == 0) return false;
  if (mode == 2) return true;
  return (long long)d->B * d->Z * d->Y * d->X >= 100000;
}
// ... few OUTPUT rows (`last`: 69 -> 4) on the same kernel with the roles of x and dY exchanged (round 4; before: the split
// kernel with 4 of 32 rows used, 3.6 ms, plus the VALU kernel for the 5 channels beyond a multiple of 32)
inline bool use_hwgrad_fc_swapped(const sr3d_conv_desc_t* d, int n_total) {
  if (d->stride != 1 || n_total > 5 || d->Cin < 16 || d->X % 8 != 0) return false;
  if (getenv("SR3D_NO_FC_SWAPPED") != nullptr) return false;
  if (d->dtype == SR3D_DTYPE_BF16) return getenv("SR3D_NO_FC_BF16") == nullptr;
  const int mode = sr3d_hconv_mode();
  if (mode == 0) return false;
  if (mode == 2) return true;
  return (long long)d->B * d->Z * d->Y * d->X >= 1000000;
}
// ... and the stride-2 layers on its de-interleaving form (sr3d_hwgrad_s2.hip): bf16 always; fp32 where the grid fills the
// chip (U-Net levels 0-2; SR3D_SPLIT_F16 as above)
inline bool use_hwgrad_s2(const sr3d_conv_desc_t* d, int n_total) {
  if (d->stride != 2 || d->X % 16 != 0) return false;
  if (d->dtype == SR3D_DTYPE_BF16) return true;
  const int mode = sr3d_hconv_mode();
  if (mode == 0 || n_total < 16) return false;
  if (mode == 2) return true;
  return (long long)d->B * d->Z * d->Y * d->X >= 100000;   // (down3.0, 128 k voxels: 1.03 -> 0.49 ms; level 3: equal)
}
inline size_t hwgrad_total_ws(const sr3d_conv_desc_t* d, int n_total) {
  const int cu = wino_wgrad_c_used(d);
  size_t bytes = align256(sr3d_hwgrad_ws_bytes(d, n_total, cu));
  if (cu < d->Cin) bytes += sr3d_wgrad_few_ws_bytes(d, n_total, d->Cin - cu);
  return bytes;
}
inline bool wino_wgrad_slices_ok(const sr3d_slice_t* dy_srcs, int n_dy) {
  for (int i = 0; i < n_dy; i++)
    if (dy_srcs[i].channels % 4 || ((uintptr_t)dy_srcs[i].ptr & 7)) return false;
  return true;
}

inline bool use_smalln(const sr3d_conv_desc_t* d, int n_total, int n_dy) {
  if (use_hwgrad(d, n_total)) return false;
  return n_total <= 4 && n_dy == 1 && d->stride == 1 && d->Cin <= 65535 && d->dtype != SR3D_DTYPE_BF16;
}

// dW[n][j] = sum_s slab[s][n][j]  (fixed order: deterministic)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                          int S, int N, int J, int Npad, int Jpad) {
  const long long total = (long long)N * J;
  const long long plane = (long long)Npad * Jpad;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(e / J), j = (int)(e - (long long)n * J);
    const float* src = slab + (long long)n * Jpad + j;
    float s = 0.f;
    for (int k = 0; k < S; k++) s += src[(long long)k * plane];
    dw[e] = s;
  }
}

// db[c] = sum dy[b][c][:]  -- two deterministic stages
template <typename T>
__global__ __launch_bounds__(256) void bias_partial_kernel(const T* __restrict__ dy, float* __restrict__ part,
                                                          int B, int C, long long vox, int nsplit) {
  const int c = blockIdx.y, sp = blockIdx.x;
  const long long per = (((vox + nsplit - 1) / nsplit) + 3) & ~3ll;   // a multiple of 4: every split starts on a 16-byte piece
  const long long v0 = sp * per, v1 = (v0 + per < vox ? v0 + per : vox);
  float s = 0.f;
  // four elements per load and four running sums (one 4-byte load per thread and a single dependent chain read at
  // 3.3 TB/s); the order of the additions is fixed by the launch geometry: deterministic
  const bool vec = vox % 4 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
  for (int b = 0; b < B; b++) {
    const T* src = dy + ((long long)b * C + c) * vox;
    if (vec) {
      f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
      for (long long v = v0 + 4 * threadIdx.x; v < v1; v += 4 * 256) a4 += ActIo<T>::ld4(src + v);
      s += (a4.x + a4.y) + (a4.z + a4.w);
    } else {
      for (long long v = v0 + threadIdx.x; v < v1; v += 256) s += ActIo<T>::ld(src + v);
    }
  }
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(long long)c * nsplit + sp] = red[0];
}

__global__ void bias_final_kernel(const float* __restrict__ part, float* __restrict__ db, int C, int nsplit) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; k++) s += part[(long long)c * nsplit + k];
  db[c] = s;
}

inline int out_dim(int z, int s) { return (z - 1) / s + 1; }

struct Plan {
  int waves_n;   // 2, 4 or 8
  int ctw;       // column tiles per wave: 7, or 3 for very few columns (conv0: 5 channels)
  int Npad, Jpad, nblk, jblk, ty, nty, ntx;
  long long ntiles, per_split;
  int S;
};

Plan make_plan(const sr3d_conv_desc_t* d, int n_total) {
  Plan pl;
  const int J = d->Cin * 27;
  // few columns: let the waves split the rows instead (stride 2 always uses 2x2: its halo tile is large)
  pl.ctw = 7;
  if (d->stride == 2)
    pl.waves_n = 4;
  else if (J <= 192 && n_total > 64)
    pl.waves_n = 4, pl.ctw = 3;   // 128 rows x 192 columns
  else
    pl.waves_n = J <= 224 ? 8 : (J <= 448 ? 4 : 2);
  const int rows = 32 * pl.waves_n, cols = (8 / pl.waves_n) * pl.ctw * 32;
  pl.nblk = ceil_div(n_total, rows), pl.jblk = ceil_div(J, cols);
  pl.Npad = pl.nblk * rows, pl.Jpad = ceil_div(J, 32) * 32;
  const int OZ = out_dim(d->Z, d->stride), OY = out_dim(d->Y, d->stride), OX = out_dim(d->X, d->stride);
  pl.ty = d->stride == 1 ? 2 : 1;
  pl.nty = ceil_div(OY, pl.ty), pl.ntx = ceil_div(OX, 32);
  pl.ntiles = (long long)d->B * pl.nty * pl.ntx * OZ;
  // enough workgroups to fill the 256 CUs (one workgroup each) a few times over, slab capped at ~192 MiB
  long long want = ceil_div(1280, pl.nblk * pl.jblk);
  const long long slab_one = (long long)pl.Npad * pl.Jpad * 4;
  const long long cap = (192ll << 20) / slab_one;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  if (want > pl.ntiles) want = pl.ntiles;
  pl.per_split = (pl.ntiles + want - 1) / want;
  pl.S = (int)((pl.ntiles + pl.per_split - 1) / pl.per_split);
  return pl;
}

template <int S_IN, int TY, int WAVES_N, bool VEC, int CTW_, typename T>
int launch_wgrad_v(const WgradParams& p, dim3 grid, hipStream_t st) {
  auto kern = wgrad_kernel<S_IN, TY, WAVES_N, VEC, CTW_, T>;
  constexpr int kLds = (int)WgradCfg<S_IN, TY, WAVES_N, CTW_>::lds_bytes;
  static SrPerDevice setup;   // (the attribute is per device, not per thread)
  if (int rc = setup.once([&]() -> int {
        SR3D_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
        return SR3D_OK;
      }))
    return rc;
  hipLaunchKernelGGL(kern, grid, dim3(512), kLds, st, p);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

// 16-byte loads need rows that start and end on float4 boundaries and 16-byte aligned tensors
bool vec_ok(const WgradParams& p) {
  if (p.IX % 4 || p.OX % 4) return false;
  for (int i = 0; i < p.x.n; i++)
    if (reinterpret_cast<uintptr_t>(p.x.ptr[i]) & 15) return false;
  for (int i = 0; i < p.dy.n; i++)
    if (reinterpret_cast<uintptr_t>(p.dy.ptr[i]) & 15) return false;
  return true;
}

template <int S_IN, int TY, int WAVES_N, int CTW_ = 7>
int launch_wgrad(const WgradParams& p, dim3 grid, bool bf, hipStream_t st) {
  if (bf)
    return vec_ok(p) ? launch_wgrad_v<S_IN, TY, WAVES_N, true, CTW_, bf16raw>(p, grid, st)
                     : launch_wgrad_v<S_IN, TY, WAVES_N, false, CTW_, bf16raw>(p, grid, st);
  return vec_ok(p) ? launch_wgrad_v<S_IN, TY, WAVES_N, true, CTW_, float>(p, grid, st)
                   : launch_wgrad_v<S_IN, TY, WAVES_N, false, CTW_, float>(p, grid, st);
}

int bias_splits(long long vox) {
  long long s = vox / 16384;
  return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

}  // namespace

extern "C" {

size_t sr3d_conv3d_bwd_weight_workspace_bytes(const sr3d_conv_desc_t* d, int n_total) {
  if (!d || d->Cin <= 0 || n_total <= 0 || (d->stride != 1 && d->stride != 2)) return 0;
  // The query does not know into how many dy slices the rows are split, which decides between the small-N kernel
  // (one slice) and the MFMA kernels: return the largest slab any of the paths the call may take needs.
  const Plan pl = make_plan(d, n_total);
  size_t bytes = (size_t)pl.S * pl.Npad * pl.Jpad * 4;
  if (use_smalln(d, n_total, 1)) bytes = std::max(bytes, (size_t)smalln_plan(d).S * d->Cin * 108 * 4);
  if (use_wino_wgrad(d, n_total)) bytes = std::max(bytes, wino_wgrad_total_ws(d, n_total));
  if (use_hwgrad(d, n_total)) bytes = std::max(bytes, hwgrad_total_ws(d, n_total));
  if (use_hwgrad_s2(d, n_total)) bytes = std::max(bytes, sr3d_hwgrad_s2_ws_bytes(d, n_total));
  if (use_hwgrad_fc(d, n_total)) bytes = std::max(bytes, sr3d_hwgrad_fc_ws_bytes(d, n_total));
  if (use_hwgrad_fc_swapped(d, n_total)) bytes = std::max(bytes, sr3d_hwgrad_fc_ws_bytes(d, n_total, true));
  return bytes;
}

int sr3d_conv3d_bwd_weight(const sr3d_conv_desc_t* d, const sr3d_slice_t* x_srcs, int n_src,
                           const sr3d_slice_t* dy_srcs, int n_dy, void* dw, void* workspace, size_t workspace_bytes,
                           const void* x_absmax, const void* dy_absmax, void* stream) {
  SR3D_CHECK(d && dw && workspace, SR3D_E_ARG, "conv3d_bwd_weight: null pointer");
  SR3D_CHECK(d->stride == 1 || d->stride == 2, SR3D_E_ARG, "conv3d_bwd_weight: stride must be 1 or 2");
  SR3D_CHECK((long long)d->Z * d->Y * d->X < (1ll << 31), SR3D_E_ARG, "conv3d_bwd_weight: grid too large");
  int n_total = 0;
  for (int i = 0; i < n_dy && i < SR3D_MAX_SRC; i++) n_total += dy_srcs[i].channels;
  SR3D_CHECK(n_total > 0, SR3D_E_ARG, "conv3d_bwd_weight: dy_srcs hold no channels");
  SR3D_CHECK(d->dtype == SR3D_DTYPE_F32 || d->dtype == SR3D_DTYPE_BF16, SR3D_E_ARG, "conv3d_bwd_weight: unknown dtype %d", d->dtype);
  const bool bf = d->dtype == SR3D_DTYPE_BF16;
  if (use_smalln(d, n_total, n_dy)) {
    const SmallNPlan sp = smalln_plan(d);
    SR3D_CHECK(workspace_bytes >= (size_t)sp.S * d->Cin * 108 * 4, SR3D_E_WORKSPACE,
               "conv3d_bwd_weight: workspace of %zu bytes is too small", workspace_bytes);
    SmallNParams q{};
    if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &q.x, "x_srcs")) return rc;
    for (int i = 0; i < q.x.n; i++) SR3D_CHECK(q.x.ptr[i], SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
    SR3D_CHECK(dy_srcs[0].ptr, SR3D_E_ARG, "dy_srcs[0].ptr is null");
    q.dy = (const float*)dy_srcs[0].ptr, q.Cin = d->Cin, q.N = n_total;
    q.Z = d->Z, q.Y = d->Y, q.X = d->X;
    q.ntz = sp.ntz, q.nty = sp.nty, q.ntx = sp.ntx, q.ntiles = sp.ntiles, q.per_split = sp.per_split;
    q.slab = (float*)workspace;
    hipStream_t st2 = (hipStream_t)stream;
    void* tok2 = nullptr;
    if (sr3d_prof_active())
      sr3d_prof_begin(SR3D_PROF_WGRAD, 2.0 * 27 * d->Cin * (double)n_total * (double)d->Z * d->Y * d->X * d->B, st2,
                      &tok2);
    hipLaunchKernelGGL(wgrad_smalln_kernel, dim3(sp.S, d->Cin), dim3(256), 0, st2, q);
    sr3d_prof_end(tok2, st2);
    SR3D_HIP(hipGetLastError());
    const int total = n_total * d->Cin * 27;
    hipLaunchKernelGGL(wgrad_smalln_reduce_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st2,
                       (const float*)workspace, (float*)dw, sp.S, n_total, d->Cin);
    SR3D_HIP(hipGetLastError());
    return SR3D_OK;
  }
  if (use_hwgrad_s2(d, n_total)) {
    const int OZ = (d->Z - 1) / 2 + 1, OY = (d->Y - 1) / 2 + 1, OX = (d->X - 1) / 2 + 1;
    ChanCat xc, dc;
    if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &xc, "x_srcs")) return rc;
    if (int rc = sr3d_make_cat(dy_srcs, n_dy, (long long)OZ * OY * OX, n_total, &dc, "dy_srcs")) return rc;
    for (int i = 0; i < xc.n; i++) SR3D_CHECK(xc.ptr[i], SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
    for (int i = 0; i < dc.n; i++) SR3D_CHECK(dc.ptr[i], SR3D_E_ARG, "dy_srcs[%d].ptr is null", i);
    if (sr3d_hwgrad_s2_ok(d, xc, dc)) {
      SR3D_CHECK(workspace_bytes >= sr3d_hwgrad_s2_ws_bytes(d, n_total), SR3D_E_WORKSPACE,
                 "conv3d_bwd_weight: workspace of %zu bytes is too small", workspace_bytes);
      return sr3d_hwgrad_s2(d, xc, dc, n_total, (float*)dw, (float*)workspace, (hipStream_t)stream, (const unsigned*)x_absmax,
                            (const unsigned*)dy_absmax);
    }
  }
  if (use_hwgrad_fc_swapped(d, n_total)) {
    ChanCat xc, dc;
    if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &xc, "x_srcs")) return rc;
    if (int rc = sr3d_make_cat(dy_srcs, n_dy, (long long)d->Z * d->Y * d->X, n_total, &dc, "dy_srcs")) return rc;
    for (int i = 0; i < xc.n; i++) SR3D_CHECK(xc.ptr[i], SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
    for (int i = 0; i < dc.n; i++) SR3D_CHECK(dc.ptr[i], SR3D_E_ARG, "dy_srcs[%d].ptr is null", i);
    if (sr3d_hwgrad_fc_ok(d, xc, dc, n_total, true)) {
      SR3D_CHECK(workspace_bytes >= sr3d_hwgrad_fc_ws_bytes(d, n_total, true), SR3D_E_WORKSPACE,
                 "conv3d_bwd_weight: workspace of %zu bytes is too small", workspace_bytes);
      return sr3d_hwgrad_fc(d, xc, dc, n_total, (float*)dw, (float*)workspace, (hipStream_t)stream, (const unsigned*)x_absmax,
                            (const unsigned*)dy_absmax, true);
    }
  }
  if (use_hwgrad_fc(d, n_total)) {
    ChanCat xc, dc;
    if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &xc, "x_srcs")) return rc;
    if (int rc = sr3d_make_cat(dy_srcs, n_dy, (long long)d->Z * d->Y * d->X, n_total, &dc, "dy_srcs")) return rc;
    for (int i = 0; i < xc.n; i++) SR3D_CHECK(xc.ptr[i], SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
    for (int i = 0; i < dc.n; i++) SR3D_CHECK(dc.ptr[i], SR3D_E_ARG, "dy_srcs[%d].ptr is null", i);
    if (sr3d_hwgrad_fc_ok(d, xc, dc, n_total)) {
      SR3D_CHECK(workspace_bytes >= sr3d_hwgrad_fc_ws_bytes(d, n_total), SR3D_E_WORKSPACE,
                 "conv3d_bwd_weight: workspace of %zu bytes is too small", workspace_bytes);
      return sr3d_hwgrad_fc(d, xc, dc, n_total, (float*)dw, (float*)workspace, (hipStream_t)stream, (const unsigned*)x_absmax,
                            (const unsigned*)dy_absmax);
    }
  }
  if (use_hwgrad(d, n_total)) {
    ChanCat xc, dc;
    if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &xc, "x_srcs")) return rc;
    if (int rc = sr3d_make_cat(dy_srcs, n_dy, (long long)d->Z * d->Y * d->X, n_total, &dc, "dy_srcs")) return rc;
    for (int i = 0; i < xc.n; i++) SR3D_CHECK(xc.ptr[i], SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
    for (int i = 0; i < dc.n; i++) SR3D_CHECK(dc.ptr[i], SR3D_E_ARG, "dy_srcs[%d].ptr is null", i);
    if (sr3d_hwgrad_ok(d, xc, dc)) {
      SR3D_CHECK(workspace_bytes >= hwgrad_total_ws(d, n_total), SR3D_E_WORKSPACE,
                 "conv3d_bwd_weight: workspace of %zu bytes is too small", workspace_bytes);
      const int cu = wino_wgrad_c_used(d);
      if (int rc = sr3d_hwgrad(d, xc, dc, n_total, cu, (float*)dw, (float*)workspace, (hipStream_t)stream,
                               (const unsigned*)x_absmax, (const unsigned*)dy_absmax))
        return rc;
      if (cu < d->Cin) {
        float* ws2 = (float*)((char*)workspace + align256(sr3d_hwgrad_ws_bytes(d, n_total, cu)));
        return sr3d_wgrad_few(d, dc, n_total, xc, cu, d->Cin - cu, (float*)dw, (long long)d->Cin * 27, ws2,
                              (hipStream_t)stream);
      }
      return SR3D_OK;
    }
  }
  if (use_wino_wgrad(d, n_total) && wino_wgrad_slices_ok(dy_srcs, n_dy)) {
    SR3D_CHECK(workspace_bytes >= wino_wgrad_total_ws(d, n_total), SR3D_E_WORKSPACE,
               "conv3d_bwd_weight: workspace of %zu bytes is too small", workspace_bytes);
    ChanCat xc, dc;
    if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &xc, "x_srcs")) return rc;
    if (int rc = sr3d_make_cat(dy_srcs, n_dy, (long long)d->Z * d->Y * d->X, n_total, &dc, "dy_srcs")) return rc;
    for (int i = 0; i < xc.n; i++) SR3D_CHECK(xc.ptr[i], SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
    for (int i = 0; i < dc.n; i++) SR3D_CHECK(dc.ptr[i], SR3D_E_ARG, "dy_srcs[%d].ptr is null", i);
    const int cu = wino_wgrad_c_used(d);
    if (int rc = sr3d_wino_wgrad(d, xc, dc, n_total, cu, (float*)dw, (float*)workspace, (hipStream_t)stream)) return rc;
    if (cu < d->Cin) {
      float* ws2 = (float*)((char*)workspace + align256(sr3d_wino_wgrad_ws_bytes(d, n_total, cu)));
      return sr3d_wgrad_few(d, dc, n_total, xc, cu, d->Cin - cu, (float*)dw, (long long)d->Cin * 27, ws2,
                            (hipStream_t)stream);
    }
    return SR3D_OK;
  }
  const Plan pl = make_plan(d, n_total);
  SR3D_CHECK(workspace_bytes >= (size_t)pl.S * pl.Npad * pl.Jpad * 4, SR3D_E_WORKSPACE,
             "conv3d_bwd_weight: workspace of %zu bytes is too small", workspace_bytes);
  WgradParams p{};
  const int OZ = out_dim(d->Z, d->stride), OY = out_dim(d->Y, d->stride), OX = out_dim(d->X, d->stride);
  if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &p.x, "x_srcs")) return rc;
  if (int rc = sr3d_make_cat(dy_srcs, n_dy, (long long)OZ * OY * OX, n_total, &p.dy, "dy_srcs")) return rc;
  for (int i = 0; i < p.x.n; i++) SR3D_CHECK(p.x.ptr[i], SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
  for (int i = 0; i < p.dy.n; i++) SR3D_CHECK(p.dy.ptr[i], SR3D_E_ARG, "dy_srcs[%d].ptr is null", i);
  p.Cin = d->Cin, p.N = n_total, p.J = d->Cin * 27;
  p.IZ = d->Z, p.IY = d->Y, p.IX = d->X;
  p.OZ = OZ, p.OY = OY, p.OX = OX;
  p.nty = pl.nty, p.ntx = pl.ntx, p.ntiles = pl.ntiles, p.per_split = pl.per_split;
  p.slab = (float*)workspace, p.Npad = pl.Npad, p.Jpad = pl.Jpad;
  dim3 grid(pl.S, pl.jblk, pl.nblk);
  SR3D_CHECK(pl.jblk <= 65535 && pl.nblk <= 65535, SR3D_E_ARG, "conv3d_bwd_weight: too many blocks");
  hipStream_t st = (hipStream_t)stream;
  void* tok = nullptr;
  if (sr3d_prof_active())
    sr3d_prof_begin(SR3D_PROF_WGRAD, 2.0 * 27 * d->Cin * (double)n_total * (double)OZ * OY * OX * d->B, st, &tok);
  int rc;
  if (d->stride == 2)
    rc = launch_wgrad<2, 1, 4>(p, grid, bf, st);
  else if (pl.ctw == 3)
    rc = launch_wgrad<1, 2, 4, 3>(p, grid, bf, st);
  else if (pl.waves_n == 8)
    rc = launch_wgrad<1, 2, 8>(p, grid, bf, st);
  else if (pl.waves_n == 4)
    rc = launch_wgrad<1, 2, 4>(p, grid, bf, st);
  else
    rc = launch_wgrad<1, 2, 2>(p, grid, bf, st);
  sr3d_prof_end(tok, st);
  if (rc) return rc;
  const long long total = (long long)n_total * p.J;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  SrProfScope prof(SR3D_PROF_PACK, 4.0 * pl.S * (double)pl.Npad * pl.Jpad, st);   // split-K slabs read once
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, (float*)dw, pl.S,
                     n_total, p.J, pl.Npad, pl.Jpad);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

size_t sr3d_bias_grad_workspace_bytes(int B, int C, long long voxels) {
  (void)B;
  return (size_t)C * bias_splits(voxels) * 4;
}

int sr3d_bias_grad(const void* dy, int B, int C, long long voxels, void* db, void* workspace, int dtype, void* stream) {
  SR3D_CHECK(dy && db && workspace && B > 0 && C > 0 && voxels > 0, SR3D_E_ARG, "bias_grad: bad argument");
  SR3D_CHECK(C <= 65535, SR3D_E_ARG, "bias_grad: too many channels");
  SR3D_CHECK(dtype == SR3D_DTYPE_F32 || dtype == SR3D_DTYPE_BF16, SR3D_E_ARG, "bias_grad: unknown dtype %d", dtype);
  SrProfScope prof(SR3D_PROF_BIAS_GRAD, (dtype == SR3D_DTYPE_BF16 ? 2.0 : 4.0) * (double)B * C * (double)voxels, (hipStream_t)stream);
  const int ns = bias_splits(voxels);
  if (dtype == SR3D_DTYPE_BF16)
    hipLaunchKernelGGL(bias_partial_kernel<bf16raw>, dim3(ns, C), dim3(256), 0, (hipStream_t)stream, (const bf16raw*)dy,
                       (float*)workspace, B, C, voxels, ns);
  else
    hipLaunchKernelGGL(bias_partial_kernel<float>, dim3(ns, C), dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                     (float*)workspace, B, C, voxels, ns);
  SR3D_HIP(hipGetLastError());
  hipLaunchKernelGGL(bias_final_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, (float*)db, C, ns);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

}  // extern "C"
