// Implicit-GEMM 3x3x3 Conv3d for gfx950 on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   D[n][m] = sum_{tap t, channel c} Wp[t][c][n] * In[c][ m shifted by tap t ]
//
//   rows  n : output channels (A operand = packed weights, staged in LDS by LDS-DMA)
//   cols  m : output voxels   (B operand = an LDS halo tile of the input, x on the lanes)
//   K       : (tap, channel), channels in chunks of KC
//
// One kernel serves the forward conv (stride 1/2), the stride-1 input gradient
// (same loop, transposed weight image, mirrored tap list) and the stride-2
// input gradient (8 parity classes, each a small conv over the coarse grid with
// 1/2/4/8 taps whose result is written to every second voxel).  The input is a
// *virtual* channel concatenation of up to 4 tensors, so torch.cat
// (reference unet.py:255-293) is never materialised.  Epilogues: bias +
// activation, gate (sigmoid * act) with the tensors the backward needs, and
// the voxel-unshuffle scatter (voxel_shuffle.py:26-42).
#include "sr3d_common.h"

#include <limits.h>

namespace {

enum { EPI_PLAIN = 0, EPI_GATED = 1, EPI_UNSHUFFLE = 2 };

struct IgemmParams {
  ChanCat in;          // K side (channels of the virtual concat)
  int K;               // valid input channels
  int IZ, IY, IX;      // input grid
  int OZ, OY, OX;      // o-space (tile space) grid
  int ntz, nty, ntx;   // tiles per dim
  int ntaps;
  int tap_off[SR3D_MAX_TAPS];  // float offset of each tap inside one channel of the halo tile
  const float* wp;     // packed weights [nblk][chunk][tap][KC][BN]
  int nchunks;
  int N;               // valid rows
  int epi, act;
  const float* bias;   // plain/unshuffle: bias[n];  gated: feature bias (may be null)
  const float* bias2;  // gated: gate bias
  ChanCat out;         // plain: destination concat (rows n are concat channels)
  float* y;            // gated / unshuffle output
  float* save_f;
  float* save_s;
  int TZ_, TY_, TX_;   // destination tensor grid
  int s_out, pz, py, px;
  int unsh_C;          // unshuffle: channels of the result (= N / 8)
  int Cg;              // gated: width of one branch (channels of y / save_f / save_s)
  int n_off;           // first GEMM row of this launch (a layer's rows may be covered by two launches)
  int grid_nblk;       // row blocks of this launch
  // Stride-2 input gradient: the 8 output-parity classes run as ONE launch (blockIdx.z = class); each class has its
  // own tap list, weight image, o-space grid and output parity.  ncls == 0: the fields above describe the launch.
  int ncls;
  struct Cls {
    int ntaps;
    int tap_off[8];
    long long wp_off;    // floats from wp
    int OZ, OY, OX, pz, py, px;
  } cls[8];
};

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == SR3D_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == SR3D_ACT_LRELU) return v > 0.f ? v : 0.01f * v;
  return v;
}

// Workgroup = 4 waves that split the TZ*TY voxel rows of the tile; every wave holds RT row tiles
// (32 output channels each) x CT voxel rows of 32 x.
template <int S_IN, int LO, int HI, int TZ, int TY, int RT, int KC>
struct IgemmCfg {
  static constexpr int BN = 32 * RT;
  static constexpr int TX = 32;
  static constexpr int HZ = (TZ - 1) * S_IN + (HI - LO) + 1;
  static constexpr int HY = (TY - 1) * S_IN + (HI - LO) + 1;
  static constexpr int HX = (TX - 1) * S_IN + (HI - LO) + 1;
  static constexpr int HCH = HZ * HY * HX;
  static constexpr int HS = (KC * HCH + 3) & ~3;  // floats, keeps the weight image 16-B aligned
  static constexpr int WM = 4;
  static constexpr int CT = TZ * TY / WM;
  static constexpr size_t lds_bytes(int ntaps) { return (size_t)(HS + ntaps * KC * BN) * 4; }
  static_assert(TZ * TY % WM == 0, "col tiles must split over waves");
};

template <int S_IN, int LO, int HI, int TZ, int TY, int RT, int KC>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmParams p) {
  using C = IgemmCfg<S_IN, LO, HI, TZ, TY, RT, KC>;
  constexpr int HY = C::HY, HX = C::HX, HCH = C::HCH, CT = C::CT, BN = C::BN;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Hs = lds;
  float* Ws = lds + C::HS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave;

  // Workgroup -> (row block, tile): the row blocks of one tile read the same input halo, so they should run
  // at the same time on the same XCD (own L2).  Hardware deals blockIdx round-robin over the 8 XCDs, hence
  // ids that are congruent mod 8 are made consecutive in (tile, row block) order (bijective for any grid).
  int v;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // launch-wide or per-class (blockIdx.z) description
  int ntaps = p.ntaps, OZ = p.OZ, OY = p.OY, OX = p.OX, pz = p.pz, py = p.py, px = p.px;
  const int* tap_off = p.tap_off;
  const float* wp = p.wp;
  if (p.ncls > 0) {
    const IgemmParams::Cls& c = p.cls[blockIdx.z];
    ntaps = c.ntaps, OZ = c.OZ, OY = c.OY, OX = c.OX, pz = c.pz, py = c.py, px = c.px;
    tap_off = c.tap_off, wp = wp + c.wp_off;
  }
  const int nblk = v % p.grid_nblk;
  int tile = v / p.grid_nblk;
  const int tix = tile % p.ntx;
  tile /= p.ntx;
  const int tiy = tile % p.nty;
  const int tiz = tile / p.nty;
  const int b = blockIdx.y;
  const int oz0 = tiz * TZ, oy0 = tiy * TY, ox0 = tix * 32;
  if (p.ncls > 0 && (oz0 >= OZ || oy0 >= OY || ox0 >= OX)) return;   // the grid is sized for the largest class
  const int gz0 = oz0 * S_IN + LO, gy0 = oy0 * S_IN + LO, gx0 = ox0 * S_IN + LO;
  const long long IZYX = (long long)p.IZ * p.IY * p.IX;

  f32x16 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; i++)
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int a_lane = (lane >> 5) * BN + (lane & 31);
  const int b_lane = (lane >> 5) * HCH + (lane & 31) * S_IN;
  int b_ct[CT];
#pragma unroll
  for (int j = 0; j < CT; j++) {
    const int ct = wm * CT + j;
    b_ct[j] = ((ct / TY) * S_IN * HY + (ct % TY) * S_IN) * HX + b_lane;
  }
  // spatial offsets of this thread's halo elements (independent of the chunk); -1 = outside the grid
  constexpr int NI = (HCH + 255) / 256;
  int hoff[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int r = tid + i * 256;
    const int hz = r / (HY * HX);
    const int r2 = r - hz * (HY * HX);
    const int hy = r2 / HX;
    const int hx = r2 - hy * HX;
    const int gz = gz0 + hz, gy = gy0 + hy, gx = gx0 + hx;
    const bool ok = r < HCH && (unsigned)gz < (unsigned)p.IZ && (unsigned)gy < (unsigned)p.IY &&
                    (unsigned)gx < (unsigned)p.IX;
    hoff[i] = ok ? (gz * p.IY + gy) * p.IX + gx : -1;
  }
  const int wblock = ntaps * KC * BN;  // floats per (nblk, chunk)
  const int ninstr = (wblock + 255) / 256;  // 1 KiB LDS-DMA pieces (the last one may be partial: wblock % 128 == 0)

  for (int chunk = 0; chunk < p.nchunks; chunk++) {
    __syncthreads();  // everyone is done reading the previous chunk
    // ---- weights: contiguous block, asynchronous global -> LDS (no VGPRs)
    {
      const float* gw = wp + (size_t)(nblk * p.nchunks + chunk) * wblock;
      for (int i = wave; i < ninstr; i += 4)
        if (i * 256 + lane * 4 < wblock)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw + i * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(Ws + i * 256), 16, 0, 0);
    }
    // ---- input halo tile [KC][HZ][HY][HX], zero outside the grid / beyond K.
    // The channel is wave-uniform (scalar source select), the spatial offsets
    // were computed once before the chunk loop.
    {
      float v[KC][NI];
#pragma unroll
      for (int c = 0; c < KC; c++) {
        const int gc = chunk * KC + c;
        const float* base = nullptr;
        if (gc < p.K) {
          const int si = cat_find(p.in, gc);
          base = p.in.ptr[si] + (long long)b * p.in.bstride[si] + (long long)(gc - p.in.cbeg[si]) * IZYX;
        }
#pragma unroll
        for (int i = 0; i < NI; i++) v[c][i] = (base != nullptr && hoff[i] >= 0) ? base[hoff[i]] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < KC; c++)
#pragma unroll
        for (int i = 0; i < NI; i++)
          if (tid + i * 256 < HCH) Hs[c * HCH + tid + i * 256] = v[c][i];
    }
    __syncthreads();  // (hipcc drains vmcnt here: the LDS-DMA has landed)

    for (int t = 0; t < ntaps; t++) {
      const float* wt = Ws + t * (KC * BN) + a_lane;
      const float* ht = Hs + tap_off[t];
#pragma unroll
      for (int kk = 0; kk < KC / 2; kk++) {
        float a[RT], bb[CT];
#pragma unroll
        for (int i = 0; i < RT; i++) a[i] = wt[(2 * kk) * BN + i * 32];
#pragma unroll
        for (int j = 0; j < CT; j++) bb[j] = ht[(2 * kk) * HCH + b_ct[j]];
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
          for (int j = 0; j < CT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ------------------------------------------------------------------ epilogue
  const int ox = ox0 + (lane & 31);
  const int nrow0 = p.n_off + nblk * BN + 4 * (lane >> 5);
  const long long TZYX = (long long)p.TZ_ * p.TY_ * p.TX_;

  if (p.epi == EPI_GATED) {
    // row tiles come in (feature, gate) pairs of the same 32 channels
    if constexpr (RT % 2 == 0) {
#pragma unroll
      for (int j = 0; j < CT; j++) {
        const int ct = wm * CT + j;
        const int oz = oz0 + ct / TY, oy = oy0 + ct % TY;
        if (oz >= OZ || oy >= OY || ox >= OX) continue;
        const long long sp = ((long long)oz * p.TY_ + oy) * p.TX_ + ox;
#pragma unroll
        for (int i = 0; i < RT; i += 2) {
          const int cbase = (p.n_off + nblk * BN + i * 32) / 2 + 4 * (lane >> 5);
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int co = cbase + (r & 3) + 8 * (r >> 2);
            if (co < p.Cg) {
              float f = acc[i][j][r];
              if (p.bias) f += p.bias[co];
              const float g = acc[i + 1][j][r] + (p.bias2 ? p.bias2[co] : 0.f);
              const float s = 1.f / (1.f + expf(-g));
              f = act_apply(f, p.act);
              const long long o = ((long long)b * p.Cg + co) * TZYX + sp;
              p.y[o] = s * f;
              if (p.save_f) p.save_f[o] = f;
              if (p.save_s) p.save_s[o] = s;
            }
          }
        }
      }
    }
  } else if (p.epi == EPI_UNSHUFFLE) {
#pragma unroll
    for (int j = 0; j < CT; j++) {
      const int ct = wm * CT + j;
      const int oz = oz0 + ct / TY, oy = oy0 + ct % TY;
      if (oz >= OZ || oy >= OY || ox >= OX) continue;
#pragma unroll
      for (int i = 0; i < RT; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int n = nrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
          if (n < p.N) {
            const float v = act_apply(acc[i][j][r] + p.bias[n], p.act);
            const int f = n / p.unsh_C, c = n - f * p.unsh_C;
            const long long o = ((long long)b * p.unsh_C + c) * TZYX +
                                ((long long)(2 * oz + (f >> 2)) * p.TY_ + (2 * oy + ((f >> 1) & 1))) * p.TX_ +
                                (2 * ox + (f & 1));
            p.y[o] = v;
          }
        }
    }
  } else {
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int n = nrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
        if (n >= p.N) continue;
        const int si = cat_find(p.out, n);
        float* base = cat_ptr(p.out, si);
        if (base == nullptr) continue;
        base += (long long)b * cat_bstride(p.out, si) + (long long)(n - cat_cbeg(p.out, si)) * TZYX;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int j = 0; j < CT; j++) {
          const int ct = wm * CT + j;
          const int oz = oz0 + ct / TY, oy = oy0 + ct % TY;
          if (oz < OZ && oy < OY && ox < OX) {
            const long long sp = ((long long)(oz * p.s_out + pz) * p.TY_ + (oy * p.s_out + py)) * p.TX_ +
                                 (ox * p.s_out + px);
            base[sp] = act_apply(acc[i][j][r] + bv, p.act);
          }
        }
      }
  }
}

// ------------------------------------------------------------------- small-N forward
// Forward conv for layers with <= 4 output channels (the model's `last`, 69 -> 4): a 32-row MFMA tile
// would be 7/8 padding, so this runs on the VALU.  Tile = 4x8x32 voxels, a thread owns 4 consecutive x
// voxels x 4 output channels (16 accumulators); wave w stages input channel w of each 4-channel chunk
// (float4 rows starting at x0-4); the weights [c][tap][4] are wave-uniform and come in by scalar loads.
typedef const __attribute__((address_space(1))) float* gfloat_p;

struct SmallFwdParams {
  ChanCat in;
  int K, N;            // input channels, output channels (<= 4)
  int Z, Y, X;
  int ntz, nty, ntx;
  const float* w;      // [K][27][4]
  const float* bias;   // may be null
  float* y;            // (B, N, Z, Y, X), or N channels inside a wider tensor (y_bstride)
  long long y_bstride; // elements between samples of y
  int act;
  // remainder rows of an input gradient whose destination slice has the activation backward fused (sr3d_conv3d_bwd_data_act):
  // act_y = the producing layer's output at the first of these channels (same strides as y); the result is multiplied by
  // lrelu'(act_y), its maximum goes to act_amax[64]; unsh_C > 0: y is that slice's buffer and receives the values in the
  // producer's SHUFFLED layout, channel ((fz*2+fy)*2+fx) * unsh_C + unsh_c0 + n on the coarse grid (needs X % 4 == 0)
  const float* act_y;
  unsigned* act_amax;
  int unsh_C, unsh_c0;
};

// NN = output channels computed (1, 2 or 4): the remainder rows of the decoder's input gradients are ONE row each (193 = 3 x 64 + 1),
// and with four accumulator columns three quarters of that launch's FMAs multiplied zeros
template <int NN>
__global__ __launch_bounds__(256) void smalln_fwd_kernel(const SmallFwdParams p) {
  constexpr int TZ = 4, TY = 8, RW = 40, RQ = RW / 4, HY = TY + 2, HZ = TZ + 2, KC = 4;
  constexpr int PZ = HY * RW, PC = HZ * PZ;
  __shared__ __attribute__((aligned(16))) float Hs[KC * PC];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xq = tid & 7, ty = (tid >> 3) & 7, tz = tid >> 6;
  // neighbouring tiles share halo rows: consecutive tile ids run on ONE XCD (block ids are dealt round-robin over the 8 XCDs, each
  // with its own L2), as in sr3d_hconv.hip -- with the plain order `last`'s forward fetched 11 GB for 2.3 GB of operands
  // (profiles/r04_pmc_hbm_traffic.json) at 5.2 TB/s: it was HBM-bound on its own over-fetch
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tix = tile % p.ntx;
  tile /= p.ntx;
  const int tiy = tile % p.nty;
  const int tiz = tile / p.nty;
  const int b = blockIdx.y;
  const int z0 = tiz * TZ, y0 = tiy * TY, x0 = tix * 32;
  const long long ZYX = (long long)p.Z * p.Y * p.X;
  bool vec = p.X % 4 == 0;
  for (int i = 0; i < p.in.n; i++) vec = vec && (reinterpret_cast<uintptr_t>(p.in.ptr[i]) & 15) == 0;

  float acc[4][NN];  // [voxel i][channel n]
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int n = 0; n < NN; n++) acc[i][n] = 0.f;

  const int nchunks = (p.K + KC - 1) / KC;
  for (int chunk = 0; chunk < nchunks; chunk++) {
    __syncthreads();
    {  // wave w stages channel chunk*4 + w
      const int gc = chunk * KC + wave;
      gfloat_p base = nullptr;
      if (gc < p.K) {
        const int si = cat_find(p.in, gc);
        base = (gfloat_p)cat_ptr(p.in, si) + ((long long)(gc - cat_cbeg(p.in, si)) * ZYX + (long long)b * cat_bstride(p.in, si));
      }
      for (int e = lane; e < HZ * HY * RQ; e += 64) {
        const int hz = e / (HY * RQ), r2 = e - hz * (HY * RQ);
        const int hy = r2 / RQ, q = r2 - hy * RQ;
        const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, xs = x0 - 4 + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (base != nullptr && (unsigned)gz < (unsigned)p.Z && (unsigned)gy < (unsigned)p.Y) {
          const gfloat_p row = base + ((long long)gz * p.Y + gy) * p.X;
          if (vec) {
            if (xs >= 0 && xs + 3 < p.X) v = *(const __attribute__((address_space(1))) f32x4*)(row + xs);
          } else {
            if ((unsigned)(xs + 0) < (unsigned)p.X) v.x = row[xs + 0];
            if ((unsigned)(xs + 1) < (unsigned)p.X) v.y = row[xs + 1];
            if ((unsigned)(xs + 2) < (unsigned)p.X) v.z = row[xs + 2];
            if ((unsigned)(xs + 3) < (unsigned)p.X) v.w = row[xs + 3];
          }
        }
        *reinterpret_cast<f32x4*>(&Hs[wave * PC + hz * PZ + hy * RW + 4 * q]) = v;
      }
    }
    __syncthreads();
    const int cmax = p.K - chunk * KC < KC ? p.K - chunk * KC : KC;
    for (int c = 0; c < cmax; c++) {
      const float* wc = p.w + (long long)(chunk * KC + c) * 108;   // wave-uniform: scalar loads
      const float* hc = &Hs[c * PC + tz * PZ + ty * RW + 4 * xq];
#pragma unroll
      for (int kz = 0; kz < 3; kz++)
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
          const float* hr = hc + kz * PZ + ky * RW;
          const f32x4 h0 = *reinterpret_cast<const f32x4*>(hr);       // x0-4+4xq .. +3
          const f32x4 h1 = *reinterpret_cast<const f32x4*>(hr + 4);   // .. +7
          const float h8 = hr[8];
          const float h[6] = {h0.w, h1.x, h1.y, h1.z, h1.w, h8};      // voxels x-1 .. x+4
#pragma unroll
          for (int kx = 0; kx < 3; kx++) {
            const float* wt = wc + ((kz * 3 + ky) * 3 + kx) * 4;
#pragma unroll
            for (int n = 0; n < NN; n++) {
              const float wv = wt[n];
#pragma unroll
              for (int i = 0; i < 4; i++) acc[i][n] += h[kx + i] * wv;
            }
          }
        }
    }
  }

  const int gz = z0 + tz, gy = y0 + ty, gx = x0 + 4 * xq;
  float amax_act = 0.f;
  if (gz < p.Z && gy < p.Y) {
#pragma unroll
    for (int n = 0; n < NN; n++) {
      if (n >= p.N) break;
      const float bv = p.bias ? p.bias[n] : 0.f;
      const long long fine = (long long)b * p.y_bstride + ((long long)n * p.Z + gz) * p.Y * p.X + (long long)gy * p.X + gx;
      float* o = p.y + fine;
      float r[4];
#pragma unroll
      for (int i = 0; i < 4; i++) r[i] = act_apply(acc[i][n] + bv, p.act);
      if (p.act_y != nullptr) {      // (fused activation backward: X % 4 == 0 and aligned tensors, checked by the caller)
        if (gx + 3 >= p.X) continue;
        const f32x4 yv = *reinterpret_cast<const f32x4*>(p.act_y + fine);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          r[i] = yv[i] > 0.f ? r[i] : 0.01f * r[i];
          amax_act = fmaxf(amax_act, fabsf(r[i]));
        }
        if (p.unsh_C > 0) {
          const long long cvox = ((long long)p.Z * p.Y * p.X) >> 3;
          const int f0 = ((gz & 1) * 2 + (gy & 1)) * 2;
          float* d0 = p.y + (long long)b * p.y_bstride + ((long long)f0 * p.unsh_C + p.unsh_c0 + n) * cvox +
                      ((long long)(gz >> 1) * (p.Y >> 1) + (gy >> 1)) * (p.X >> 1) + (gx >> 1);
          *reinterpret_cast<float2*>(d0) = float2{r[0], r[2]};
          *reinterpret_cast<float2*>(d0 + (long long)p.unsh_C * cvox) = float2{r[1], r[3]};
          continue;
        }
      }
      if (vec && gx + 3 < p.X) {
        *reinterpret_cast<f32x4*>(o) = f32x4{r[0], r[1], r[2], r[3]};
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (gx + i < p.X) o[i] = r[i];
      }
    }
  }
  if (p.act_amax != nullptr) {   // (kernel-argument condition: uniform)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax_act = fmaxf(amax_act, __shfl_xor(amax_act, off, 64));
    if (lane == 0 && amax_act > 0.f) atomicMax(p.act_amax + (blockIdx.x & 63), __float_as_uint(amax_act));
  }
}

inline void launch_smalln_fwd(const SmallFwdParams& q, int B, hipStream_t st) {
  const dim3 grid(q.ntz * q.nty * q.ntx, B);
  if (q.N == 1)
    hipLaunchKernelGGL(smalln_fwd_kernel<1>, grid, dim3(256), 0, st, q);
  else if (q.N == 2)
    hipLaunchKernelGGL(smalln_fwd_kernel<2>, grid, dim3(256), 0, st, q);
  else
    hipLaunchKernelGGL(smalln_fwd_kernel<4>, grid, dim3(256), 0, st, q);
}

// w[(c*27 + t)*4 + n] = W[n][c][t]
__global__ void pack_smalln_kernel(const float* __restrict__ w, float* __restrict__ wp, int N, int K) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= K * 108) return;
  const int n = e & 3, r = e >> 2;  // r = c*27 + t
  wp[e] = n < N ? w[(long long)n * K * 27 + r] : 0.f;
}

// input-gradient rows on the VALU path: wp[(k*27 + t)*4 + r] = W[k][ci0 + r][26 - t]  (k over feat then gate)
__global__ void pack_smalln_bwd_kernel(const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ wp,
                                       int Cout, int Cin, int K, int ci0, int nrows) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= K * 108) return;
  const int r = e & 3, kt = e >> 2, k = kt / 27, tp = kt - k * 27;
  const float* w = k < Cout ? w1 + (long long)k * Cin * 27 : w2 + (long long)(k - Cout) * Cin * 27;
  wp[e] = r < nrows ? w[(ci0 + r) * 27 + (26 - tp)] : 0.f;
}

// bfloat16 activations (sr3d_conv_desc_t.dtype): every convolution runs on the bf16 form of the split kernels
// (sr3d_hconv.hip / sr3d_hconv_s2.hip, one bf16 MFMA per product), whatever its size
inline bool is_bf(const sr3d_conv_desc_t* d) { return d->dtype == SR3D_DTYPE_BF16; }

inline bool use_smalln_fwd(const sr3d_conv_desc_t* d, int kind) {
  return kind == SR3D_PACK_FWD && d->Cout <= 4 && d->stride == 1 && !is_bf(d);
}

// stride-1 convolutions (forward and input gradient) run on the Winograd kernel (sr3d_wino.hip)
// (the Winograd kernel keeps a table of channel pointers in LDS: at most 4096 channels on the K side, which is Cin
// forward and n_dy * Cout for the input gradient - checked again there)
inline bool use_wino(const sr3d_conv_desc_t* d) {
  return d->stride == 1 && sr3d_wino_enabled() && d->Cin <= SR3D_WINO_MAX_K && d->Cout <= SR3D_WINO_MAX_K;
}
// Stride-1 convolutions with >= 32 GEMM-K channels run on the split-f16 kernel (sr3d_hconv.hip; SR3D_SPLIT_F16=0 turns
// it off, =2 forces it) when the launch fills the chip: its 2 x 4 x 32-voxel workgroups come two per CU, and on the small grids of
// U-Net levels 3-4 the Winograd kernel with its one-tile workgroups is the faster one (measured: up4.convs 0.68 vs 0.89 ms).
inline bool use_hconv(const sr3d_conv_desc_t* d, int K, int rows) {
  if (is_bf(d)) return d->stride == 1;
  const int mode = sr3d_hconv_mode();
  // (K <= 8 -- conv0, the input gradient of `last` -- is ONE half chunk of the kernel: 7 phases; 9 .. 31 channels stay with Winograd)
  if (d->stride != 1 || (K < 32 && K > 8) || rows < 16 || mode == 0) return false;
  if (mode == 2) return true;
  // (re-measured at the end of round 3, after the kernel's refill was restructured: faster than the Winograd kernel from U-Net
  //  level 3 up whatever the launch size -- up3.up input gradient, 400 workgroups: 2.61 -> 2.04 ms; up4.convs: 0.68 -> 0.56 ms --
  //  and equal on level 4, profiles/r03w_layers_small_grids_default_vs_forced.log; the threshold was 448)
  const long long wgs = (long long)d->B * ceil_div(d->Z, 2) * ceil_div(d->Y, 4) * ceil_div(d->X, 32) * ceil_div(rows, 64);
  return wgs >= 100;
}
// ... and the stride-2 layers on its parity-class form (sr3d_hconv_s2.hip); bwd: 8 class launches over the coarse grid
inline bool use_hconv_s2(const sr3d_conv_desc_t* d, int K, int rows, bool bwd = false) {
  if (is_bf(d)) return d->stride == 2;
  const int mode = sr3d_hconv_mode();
  if (d->stride != 2 || K < 32 || rows < 16 || mode == 0) return false;
  if (mode == 2) return true;
  const int oz = (d->Z - 1) / 2 + 1, oy = (d->Y - 1) / 2 + 1, ox = (d->X - 1) / 2 + 1;
  // (re-measured at the end of round 3, profiles/r03w_layers_small_grids_default_vs_forced.log: the input gradient is faster than
  //  the 8-class fp32 kernel down to level 3 -- down3.0 1.32 -> 0.57 ms, down4.0 0.59 -> 0.24 ms --, the forward loses on
  //  level 3's 120 workgroups: 0.46 -> 0.59 ms)
  return (long long)d->B * ceil_div(oz, 2) * ceil_div(oy, 4) * ceil_div(ox, 32) * ceil_div(rows, 64) >= (bwd ? 50 : 200);
}
inline int hconv_fwd_rows(const sr3d_conv_desc_t* d, int kind) {
  return kind == SR3D_PACK_FWD_GATED ? 64 * ((d->Cout + 31) / 32) : d->Cout;
}
inline int wino_fwd_rows(const sr3d_conv_desc_t* d, int kind) {
  return kind == SR3D_PACK_FWD_GATED ? 32 * ((d->Cout + 15) / 16) : d->Cout;
}

// --------------------------------------------------------------------- packing
// Packed image of one launch region: [nblk][chunk][tap][KC][BN] with BN = 32*RT.
// A layer's rows are covered by (U / 4) blocks of 128 rows plus one block with the
// remaining U % 4 row tiles, so no more than 31 padded rows are ever multiplied.
struct PackParams {
  const float* w1;
  const float* w2;
  float* wp;
  int Cout, Cin;   // of the physical (Cout, Cin, 27) tensors
  int kind;
  int K, N;        // logical GEMM dims (N counts compacted rows for the backward kinds)
  int nchunks, nblk, BN, KC;
  int n_off;       // first logical row of this region
  int ntaps;
  int tap[SR3D_MAX_TAPS];  // original tap index (kz*3+ky)*3+kx of packed tap t
  // backward kinds: logical row n -> input channel, rows of slices that need no gradient are skipped
  int rbeg[SR3D_MAX_SRC + 1];  // first logical row of needed slice i
  int cbeg[SR3D_MAX_SRC];      // its first input channel
};

__global__ void pack_kernel(const PackParams p) {
  const long long total = (long long)p.nblk * p.nchunks * p.ntaps * p.KC * p.BN;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    long long r = e;
    const int nn = r % p.BN;
    r /= p.BN;
    const int kc = r % p.KC;
    r /= p.KC;
    const int t = r % p.ntaps;
    r /= p.ntaps;
    const int chunk = r % p.nchunks;
    const int nb = r / p.nchunks;
    const int n = p.n_off + nb * p.BN + nn, k = chunk * p.KC + kc;
    float v = 0.f;
    if (n < p.N && k < p.K) {
      const int tap = p.tap[t];
      if (p.kind == SR3D_PACK_FWD) {
        v = p.w1[((long long)n * p.Cin + k) * 27 + tap];
      } else if (p.kind == SR3D_PACK_FWD_GATED) {
        const int blk = n >> 5;
        const int co = (blk >> 1) * 32 + (n & 31);
        if (co < p.Cout) v = ((blk & 1) ? p.w2 : p.w1)[((long long)co * p.Cin + k) * 27 + tap];
      } else {
        const int si = (n >= p.rbeg[1]) + (n >= p.rbeg[2]) + (n >= p.rbeg[3]);
        const int ci = p.cbeg[si] + (n - p.rbeg[si]);
        const float* w = k < p.Cout ? p.w1 + (long long)k * p.Cin * 27 : p.w2 + (long long)(k - p.Cout) * p.Cin * 27;
        v = w[ci * 27 + tap];
      }
    }
    p.wp[e] = v;
  }
}

// ------------------------------------------------------------------ host side
constexpr int kKC = 4;

inline int out_dim(int z, int s) { return (z - 1) / s + 1; }

// How the N rows of a GEMM are cut into launches: `nblk` blocks of `rt` row tiles (32 rows each) plus one
// block with the remaining `rem` tiles.  rt = 4 gives the most operand reuse; when the spatial grid is so
// small that 4-tile blocks would leave CUs idle (levels 2-4 of the U-Net at batch 1), finer blocks are used.
struct RowPlan {
  int units;       // 32-row tiles
  int rt;          // tiles per main block: 4, 2 or 1
  int nblk;        // main blocks
  int rem;         // tiles in the last block (0 .. rt-1)
};

RowPlan row_plan(int n_rows, long long ntiles, bool pairs, int rt_max = 4) {
  RowPlan r;
  r.units = ceil_div(n_rows, 32);
  r.rt = rt_max;
  while (r.rt > (pairs ? 2 : 1) && ntiles * ceil_div(r.units, r.rt) < 1024) r.rt /= 2;
  r.nblk = r.units / r.rt;
  r.rem = r.units % r.rt;
  return r;
}

inline size_t region_floats(int nblk, int nchunks, int ntaps, int rt) {
  return (size_t)nblk * nchunks * ntaps * kKC * 32 * rt;
}

inline size_t image_floats(int units, int nchunks, int ntaps) { return region_floats(units, nchunks, ntaps, 1); }

inline long long tiles_of(int B, int oz, int oy, int ox, int tz, int ty) {
  return (long long)B * ceil_div(oz, tz) * ceil_div(oy, ty) * ceil_div(ox, 32);
}

// stride-2 backward: taps of parity class (pz,py,px), in (kz,ky,kx) order
int class_taps(int cls, int* orig, int* dz, int* dy, int* dx) {
  const int par[3] = {(cls >> 2) & 1, (cls >> 1) & 1, cls & 1};
  int k[3][2], dd[3][2], cnt[3];
  for (int a = 0; a < 3; a++) {
    if (par[a] == 0) {
      cnt[a] = 1, k[a][0] = 1, dd[a][0] = 0;
    } else {
      cnt[a] = 2, k[a][0] = 0, dd[a][0] = 1, k[a][1] = 2, dd[a][1] = 0;
    }
  }
  int n = 0;
  for (int a = 0; a < cnt[0]; a++)
    for (int bb = 0; bb < cnt[1]; bb++)
      for (int c = 0; c < cnt[2]; c++) {
        orig[n] = (k[0][a] * 3 + k[1][bb]) * 3 + k[2][c];
        dz[n] = dd[0][a], dy[n] = dd[1][bb], dx[n] = dd[2][c];
        n++;
      }
  return n;
}

int run_pack(PackParams p, const RowPlan& rp, float* image, hipStream_t st) {
  // region A: nblk blocks of 32*rt rows; region B: one block of 32*rem rows
  for (int region = 0; region < 2; region++) {
    const int rt = region == 0 ? rp.rt : rp.rem;
    const int nblk = region == 0 ? rp.nblk : (rp.rem ? 1 : 0);
    if (nblk == 0) continue;
    p.nblk = nblk, p.BN = 32 * rt, p.KC = kKC;
    p.n_off = region == 0 ? 0 : rp.nblk * rp.rt * 32;
    p.wp = image + (region == 0 ? 0 : region_floats(rp.nblk, p.nchunks, p.ntaps, rp.rt));
    const long long total = (long long)p.nblk * p.nchunks * p.ntaps * p.KC * p.BN;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    SrProfScope prof(SR3D_PROF_PACK, 8.0 * (double)total, st);
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, st, p);
    SR3D_HIP(hipGetLastError());
  }
  return SR3D_OK;
}

template <int S_IN, int LO, int HI, int TZ, int TY, int RT, int KC>
int launch_one(IgemmParams& p, int B, int nblk, hipStream_t st) {
  using C = IgemmCfg<S_IN, LO, HI, TZ, TY, RT, KC>;
  p.ntz = ceil_div(p.OZ, TZ), p.nty = ceil_div(p.OY, TY), p.ntx = ceil_div(p.OX, 32);
  auto kern = igemm_kernel<S_IN, LO, HI, TZ, TY, RT, KC>;
  const size_t lds = C::lds_bytes(p.ncls > 0 ? 8 : p.ntaps);
  {   // the attribute is per device; this instantiation is launched with several LDS sizes: keep the largest per device
    static std::mutex mu;
    static size_t configured[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (lds > configured[dev & 63]) {
      SR3D_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured[dev & 63] = lds;
    }
  }
  p.grid_nblk = nblk;
  SR3D_CHECK((long long)p.ntz * p.nty * p.ntx * nblk < (1ll << 31) && B <= 65535, SR3D_E_ARG, "igemm: grid too large");
  dim3 grid(p.ntz * p.nty * p.ntx * nblk, B, p.ncls > 0 ? p.ncls : 1);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

// all launches of one GEMM: region A with RT = 4, region B with RT = rem
template <int S_IN, int LO, int HI, int TZ, int TY>
int launch(IgemmParams p, int B, const RowPlan& rp, const float* image, hipStream_t st) {
  void* tok = nullptr;
  if (sr3d_prof_active()) {
    // algorithmic FLOPs: 2 * taps * K * (valid rows) * output voxels (no padding counted)
    const double rows = p.epi == EPI_GATED ? 2.0 * p.Cg : (double)p.N;
    double flops = 2.0 * p.ntaps * p.K * rows * (double)p.OZ * p.OY * p.OX * B;
    if (p.ncls > 0) {
      flops = 0;
      for (int c = 0; c < p.ncls; c++)
        flops += 2.0 * p.cls[c].ntaps * p.K * rows * (double)p.cls[c].OZ * p.cls[c].OY * p.cls[c].OX * B;
    }
    const int id = S_IN == 2 ? SR3D_PROF_IGEMM_S2 : (LO == 0 ? SR3D_PROF_IGEMM_BWD_S2 : SR3D_PROF_IGEMM_S1);
    sr3d_prof_begin(id, flops, st, &tok);
  }
  int rc = SR3D_OK;
  if (rp.nblk > 0) {
    p.wp = image, p.n_off = 0;
    switch (rp.rt) {
      case 4: rc = launch_one<S_IN, LO, HI, TZ, TY, 4, kKC>(p, B, rp.nblk, st); break;
      case 2: rc = launch_one<S_IN, LO, HI, TZ, TY, 2, kKC>(p, B, rp.nblk, st); break;
      default: rc = launch_one<S_IN, LO, HI, TZ, TY, 1, kKC>(p, B, rp.nblk, st); break;
    }
  }
  if (rc == SR3D_OK && rp.rem > 0) {
    p.wp = image + region_floats(rp.nblk, p.nchunks, p.ntaps, rp.rt), p.n_off = rp.nblk * rp.rt * 32;
    if (p.ncls > 0) {   // per class: region B follows region A inside the class's own image
      p.wp = image;
      for (int c = 0; c < p.ncls; c++) p.cls[c].wp_off += (long long)region_floats(rp.nblk, p.nchunks, p.cls[c].ntaps, rp.rt);
    }
    switch (rp.rem) {
      case 1: rc = launch_one<S_IN, LO, HI, TZ, TY, 1, kKC>(p, B, 1, st); break;
      case 2: rc = launch_one<S_IN, LO, HI, TZ, TY, 2, kKC>(p, B, 1, st); break;
      default: rc = launch_one<S_IN, LO, HI, TZ, TY, 3, kKC>(p, B, 1, st); break;
    }
  }
  sr3d_prof_end(tok, st);
  return rc;
}

void full_taps(IgemmParams& p, int HY, int HX, bool mirrored) {
  p.ntaps = 27;
  for (int kz = 0; kz < 3; kz++)
    for (int ky = 0; ky < 3; ky++)
      for (int kx = 0; kx < 3; kx++) {
        const int t = (kz * 3 + ky) * 3 + kx;
        // forward: input offset d = k-1 -> (d-LO) = k ; stride-1 backward: d = 1-k -> (d-LO) = 2-k
        const int a = mirrored ? 2 - kz : kz, b = mirrored ? 2 - ky : ky, c = mirrored ? 2 - kx : kx;
        p.tap_off[t] = (a * HY + b) * HX + c;
      }
}

int check_desc(const sr3d_conv_desc_t* d) {
  SR3D_CHECK(d != nullptr, SR3D_E_ARG, "conv desc is null");
  SR3D_CHECK(d->B > 0 && d->Cin > 0 && d->Cout > 0 && d->Z > 0 && d->Y > 0 && d->X > 0, SR3D_E_ARG,
             "conv desc: non-positive dimension");
  SR3D_CHECK(d->stride == 1 || d->stride == 2, SR3D_E_ARG, "conv desc: stride must be 1 or 2 (got %d)", d->stride);
  SR3D_CHECK((long long)d->Z * d->Y * d->X < (1ll << 31), SR3D_E_ARG, "conv desc: grid has >= 2^31 voxels");
  SR3D_CHECK(d->dtype == SR3D_DTYPE_F32 || d->dtype == SR3D_DTYPE_BF16, SR3D_E_ARG, "conv desc: unknown dtype %d", d->dtype);
  return SR3D_OK;
}

inline int fwd_rows(const sr3d_conv_desc_t* d, int kind) {
  return kind == SR3D_PACK_FWD_GATED ? 2 * ((d->Cout + 31) / 32) * 32 : d->Cout;
}

// the forward launch plan depends only on the descriptor, so sr3d_pack_weights and sr3d_*_fwd agree on it
RowPlan fwd_plan(const sr3d_conv_desc_t* d, int rows, bool gated) {
  const int oz = out_dim(d->Z, d->stride), oy = out_dim(d->Y, d->stride), ox = out_dim(d->X, d->stride);
  // stride 2: the halo tile is 2x larger per voxel; 64-row blocks keep two workgroups resident per CU
  return row_plan(rows, tiles_of(d->B, oz, oy, ox, d->stride == 1 ? 2 : 1, 4), gated, d->stride == 1 ? 4 : 2);
}

int forward_common(const sr3d_conv_desc_t* d, const sr3d_slice_t* x_srcs, int n_src, IgemmParams& p, int dst_scale,
                   const float* image, hipStream_t st, void* x_absmax = nullptr) {
  const long long vox = (long long)d->Z * d->Y * d->X;
  if (int rc = sr3d_make_cat(x_srcs, n_src, vox, d->Cin, &p.in, "x_srcs")) return rc;
  for (int i = 0; i < p.in.n; i++) SR3D_CHECK(p.in.ptr[i] != nullptr, SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
  p.K = d->Cin;
  p.IZ = d->Z, p.IY = d->Y, p.IX = d->X;
  p.OZ = out_dim(d->Z, d->stride), p.OY = out_dim(d->Y, d->stride), p.OX = out_dim(d->X, d->stride);
  p.TZ_ = p.OZ * dst_scale, p.TY_ = p.OY * dst_scale, p.TX_ = p.OX * dst_scale;
  p.s_out = 1, p.pz = p.py = p.px = 0;
  if (use_hconv_s2(d, d->Cin, p.N) && p.epi != EPI_UNSHUFFLE) {
    SrHconvS2Params q{};
    q.in = p.in, q.K = p.K;
    q.IZ = d->Z, q.IY = d->Y, q.IX = d->X;
    q.Z = p.OZ, q.Y = p.OY, q.X = p.OX;
    q.N = p.N, q.n_off = 0, q.epi = p.epi, q.act = p.act, q.bias = p.bias, q.bias2 = p.bias2;
    q.out = p.out, q.y = p.y, q.save_f = p.save_f, q.save_s = p.save_s, q.Cg = p.Cg;
    q.TZ_ = p.TZ_, q.TY_ = p.TY_, q.TX_ = p.TX_;
    q.amax_out = is_bf(d) ? nullptr : (unsigned*)x_absmax;
    return sr3d_hconv_s2_launch(1, q, image, d->B, is_bf(d), st);
  }
  SR3D_CHECK(!is_bf(d), SR3D_E_ARG, "bf16 activations: stride-2 convolution with the unshuffle epilogue is not implemented");
  if (x_absmax != nullptr && d->stride == 2)   // (sr3d_conv3d_fwd_exports_absmax promised the maxima, this path has no by-product: sweep)
    for (int i = 0; i < p.in.n; i++)
      if (int rc = sr3d_absmax_launch(p.in.ptr[i], (long long)d->B * p.in.bstride[i], (unsigned*)x_absmax + 64 * i, st)) return rc;
  p.nchunks = ceil_div(p.K, kKC);
  const RowPlan rp = fwd_plan(d, p.N, p.epi == EPI_GATED);
  if (d->stride == 1) {
    using C = IgemmCfg<1, -1, 1, 2, 4, 4, kKC>;
    full_taps(p, C::HY, C::HX, false);
    return launch<1, -1, 1, 2, 4>(p, d->B, rp, image, st);
  }
  using C = IgemmCfg<2, -1, 1, 1, 4, 4, kKC>;
  full_taps(p, C::HY, C::HX, false);
  return launch<2, -1, 1, 1, 4>(p, d->B, rp, image, st);
}

}  // namespace

extern "C" {

size_t sr3d_packed_weight_bytes(const sr3d_conv_desc_t* d, int kind) {
  if (kind == SR3D_PACK_FWD_UNSHUFFLE) kind = SR3D_PACK_FWD;   // same size, other row order
  if (check_desc(d) != SR3D_OK || (kind != SR3D_PACK_FWD && kind != SR3D_PACK_FWD_GATED)) return 0;
  if (use_smalln_fwd(d, kind)) return (size_t)d->Cin * 108 * 4;
  if (use_hconv(d, d->Cin, hconv_fwd_rows(d, kind))) return sr3d_hconv_image_bytes(hconv_fwd_rows(d, kind), d->Cin, is_bf(d));
  if (use_hconv_s2(d, d->Cin, hconv_fwd_rows(d, kind))) return sr3d_hconv_s2_image_bytes(hconv_fwd_rows(d, kind), d->Cin, is_bf(d));
  if (use_wino(d)) return sr3d_wino_image_floats(wino_fwd_rows(d, kind), d->Cin) * 4;
  return image_floats(ceil_div(fwd_rows(d, kind), 32), ceil_div(d->Cin, kKC), 27) * 4;
}

int sr3d_pack_weights(const sr3d_conv_desc_t* d, int kind, const void* w_feat, const void* w_gate, void* w_packed,
                      void* stream) {
  if (int rc = check_desc(d)) return rc;
  // SR3D_PACK_FWD_UNSHUFFLE: the image of a layer whose epilogue is the voxel unshuffle.  The split-f16 / bf16 kernel
  // takes its rows in unshuffle order (paired stores); the other kernels keep channel order.
  const bool unsh = kind == SR3D_PACK_FWD_UNSHUFFLE;
  if (unsh) kind = SR3D_PACK_FWD;
  SR3D_CHECK(!unsh || d->Cout % 8 == 0, SR3D_E_ARG, "pack: unshuffle needs Cout %% 8 == 0 (got %d)", d->Cout);
  SR3D_CHECK(kind == SR3D_PACK_FWD || kind == SR3D_PACK_FWD_GATED, SR3D_E_ARG, "pack: unknown kind %d", kind);
  SR3D_CHECK(w_feat && w_packed, SR3D_E_ARG, "pack: null pointer");
  SR3D_CHECK(kind != SR3D_PACK_FWD_GATED || w_gate, SR3D_E_ARG, "pack: gated kind needs w_gate");
  if (use_smalln_fwd(d, kind)) {
    hipLaunchKernelGGL(pack_smalln_kernel, dim3(ceil_div(d->Cin * 108, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)w_feat, (float*)w_packed, d->Cout, d->Cin);
    SR3D_HIP(hipGetLastError());
    return SR3D_OK;
  }
  if (use_hconv(d, d->Cin, hconv_fwd_rows(d, kind)))
    return sr3d_hconv_pack(kind, d->Cout, d->Cin, hconv_fwd_rows(d, kind), d->Cin, (const float*)w_feat,
                           (const float*)w_gate, nullptr, nullptr, w_packed, is_bf(d), (hipStream_t)stream, unsh ? d->Cout / 8 : 0);
  if (use_hconv_s2(d, d->Cin, hconv_fwd_rows(d, kind)))
    return sr3d_hconv_s2_pack(sr3d_hconv_s2_fwd_paired(d->X) ? 3 : 1, kind, d->Cout, d->Cin, hconv_fwd_rows(d, kind), d->Cin, (const float*)w_feat,
                              (const float*)w_gate, nullptr, nullptr, w_packed, is_bf(d), (hipStream_t)stream);
  if (use_wino(d))
    return sr3d_wino_pack(kind, d->Cout, d->Cin, wino_fwd_rows(d, kind), d->Cin, (const float*)w_feat,
                          (const float*)w_gate, nullptr, nullptr, (float*)w_packed, (hipStream_t)stream);
  PackParams p{};
  p.w1 = (const float*)w_feat, p.w2 = (const float*)w_gate;
  p.Cout = d->Cout, p.Cin = d->Cin, p.kind = kind, p.K = d->Cin, p.N = fwd_rows(d, kind);
  p.nchunks = ceil_div(p.K, kKC);
  p.ntaps = 27;
  for (int t = 0; t < 27; t++) p.tap[t] = t;
  return run_pack(p, fwd_plan(d, p.N, kind == SR3D_PACK_FWD_GATED), (float*)w_packed, (hipStream_t)stream);
}

int sr3d_conv3d_fwd_exports_absmax(const sr3d_conv_desc_t* d, int gated) {
  if (check_desc(d) != SR3D_OK || is_bf(d)) return 0;
  const int kind = gated ? SR3D_PACK_FWD_GATED : SR3D_PACK_FWD;
  if (use_smalln_fwd(d, kind)) return 0;
  return (use_hconv(d, d->Cin, hconv_fwd_rows(d, kind)) || use_hconv_s2(d, d->Cin, hconv_fwd_rows(d, kind))) ? 1 : 0;
}

int sr3d_conv3d_fwd(const sr3d_conv_desc_t* d, const sr3d_slice_t* x_srcs, int n_src, const void* w_packed,
                    const void* bias, void* y, int act, int unshuffle, void* x_absmax, void* stream) {
  if (int rc = check_desc(d)) return rc;
  SR3D_CHECK(w_packed && y, SR3D_E_ARG, "conv3d_fwd: null pointer");
  const bool out_f32 = (act & SR3D_ACT_OUT_F32) != 0;
  act &= ~SR3D_ACT_OUT_F32;
  SR3D_CHECK(act >= 0 && act <= 2, SR3D_E_ARG, "conv3d_fwd: unknown activation %d", act);
  SR3D_CHECK(!out_f32 || (is_bf(d) && d->stride == 1 && !unshuffle), SR3D_E_ARG,
             "conv3d_fwd: SR3D_ACT_OUT_F32 is the fp32 output of a bf16-storage, stride-1, plain layer");
  if (use_smalln_fwd(d, SR3D_PACK_FWD) && !unshuffle) {
    SmallFwdParams q{};
    if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &q.in, "x_srcs")) return rc;
    for (int i = 0; i < q.in.n; i++) SR3D_CHECK(q.in.ptr[i] != nullptr, SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
    q.K = d->Cin, q.N = d->Cout, q.Z = d->Z, q.Y = d->Y, q.X = d->X;
    q.ntz = ceil_div(d->Z, 4), q.nty = ceil_div(d->Y, 8), q.ntx = ceil_div(d->X, 32);
    q.w = (const float*)w_packed, q.bias = (const float*)bias, q.y = (float*)y, q.act = act;
    q.y_bstride = (long long)d->Cout * d->Z * d->Y * d->X;
    SR3D_CHECK(d->B <= 65535, SR3D_E_ARG, "conv3d_fwd: batch too large");
    launch_smalln_fwd(q, d->B, (hipStream_t)stream);
    SR3D_HIP(hipGetLastError());
    return SR3D_OK;
  }
  if (use_hconv(d, d->Cin, d->Cout)) {
    SrHconvParams q{};
    if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &q.in, "x_srcs")) return rc;
    for (int i = 0; i < q.in.n; i++) SR3D_CHECK(q.in.ptr[i] != nullptr, SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
    q.K = d->Cin, q.Z = d->Z, q.Y = d->Y, q.X = d->X, q.N = d->Cout;
    q.act = act, q.bias = (const float*)bias;
    q.amax_out = is_bf(d) ? nullptr : (unsigned*)x_absmax;
    q.out_f32 = out_f32 ? 1 : 0;
    if (unshuffle) {
      SR3D_CHECK(d->Cout % 8 == 0 && bias != nullptr, SR3D_E_ARG,
                 "conv3d_fwd: unshuffle needs Cout %% 8 == 0 and a bias (Cout = %d)", d->Cout);
      q.epi = SR3D_EPI_UNSHUFFLE, q.y = (float*)y, q.unsh_C = d->Cout / 8;
      q.TZ_ = 2 * d->Z, q.TY_ = 2 * d->Y, q.TX_ = 2 * d->X;
    } else {
      q.epi = SR3D_EPI_PLAIN;
      sr3d_slice_t ys{y, d->Cout};
      if (int rc = sr3d_make_cat(&ys, 1, (long long)d->Z * d->Y * d->X, d->Cout, &q.out, "y")) return rc;
      q.TZ_ = d->Z, q.TY_ = d->Y, q.TX_ = d->X;
    }
    return sr3d_hconv_launch(q, w_packed, d->B, is_bf(d), (hipStream_t)stream);
  }
  if (use_wino(d)) {
    SrWinoParams q{};
    if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &q.in, "x_srcs")) return rc;
    for (int i = 0; i < q.in.n; i++) SR3D_CHECK(q.in.ptr[i] != nullptr, SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
    q.K = d->Cin, q.Z = d->Z, q.Y = d->Y, q.X = d->X, q.N = d->Cout;
    q.up = (const float*)w_packed, q.act = act, q.bias = (const float*)bias;
    if (unshuffle) {
      SR3D_CHECK(d->Cout % 8 == 0 && bias != nullptr, SR3D_E_ARG,
                 "conv3d_fwd: unshuffle needs Cout %% 8 == 0 and a bias (Cout = %d)", d->Cout);
      q.epi = SR3D_EPI_UNSHUFFLE, q.y = (float*)y, q.unsh_C = d->Cout / 8;
      q.TZ_ = 2 * d->Z, q.TY_ = 2 * d->Y, q.TX_ = 2 * d->X;
    } else {
      q.epi = SR3D_EPI_PLAIN;
      sr3d_slice_t ys{y, d->Cout};
      if (int rc = sr3d_make_cat(&ys, 1, (long long)d->Z * d->Y * d->X, d->Cout, &q.out, "y")) return rc;
      q.TZ_ = d->Z, q.TY_ = d->Y, q.TX_ = d->X;
    }
    return sr3d_wino_launch(q, d->B, (hipStream_t)stream);
  }
  IgemmParams p{};
  p.N = d->Cout;
  p.act = act;
  p.bias = (const float*)bias;
  const int OZ = out_dim(d->Z, d->stride), OY = out_dim(d->Y, d->stride), OX = out_dim(d->X, d->stride);
  if (unshuffle) {
    SR3D_CHECK(d->Cout % 8 == 0, SR3D_E_ARG, "conv3d_fwd: unshuffle needs Cout %% 8 == 0 (got %d)", d->Cout);
    SR3D_CHECK(bias != nullptr, SR3D_E_ARG, "conv3d_fwd: unshuffle epilogue expects a bias");
    SR3D_CHECK(d->stride == 1, SR3D_E_ARG, "conv3d_fwd: unshuffle epilogue is stride-1 only");
    p.epi = EPI_UNSHUFFLE;
    p.y = (float*)y;
    p.unsh_C = d->Cout / 8;
  } else {
    p.epi = EPI_PLAIN;
    sr3d_slice_t ys{y, d->Cout};
    if (int rc = sr3d_make_cat(&ys, 1, (long long)OZ * OY * OX, d->Cout, &p.out, "y")) return rc;
  }
  return forward_common(d, x_srcs, n_src, p, unshuffle ? 2 : 1, (const float*)w_packed, (hipStream_t)stream, x_absmax);
}

int sr3d_gated_conv3d_fwd(const sr3d_conv_desc_t* d, const sr3d_slice_t* x_srcs, int n_src, const void* w_packed,
                          const void* bias_f, const void* bias_g, void* y, void* save_f, void* save_s, int act,
                          void* x_absmax, void* stream) {
  if (int rc = check_desc(d)) return rc;
  SR3D_CHECK(w_packed && y, SR3D_E_ARG, "gated_conv3d_fwd: null pointer");
  SR3D_CHECK(save_f == nullptr || save_s != nullptr, SR3D_E_ARG, "gated_conv3d_fwd: save_f without save_s");
  SR3D_CHECK(act >= 0 && act <= 2, SR3D_E_ARG, "gated_conv3d_fwd: unknown activation %d", act);
  if (use_hconv(d, d->Cin, hconv_fwd_rows(d, SR3D_PACK_FWD_GATED))) {
    SrHconvParams q{};
    if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &q.in, "x_srcs")) return rc;
    for (int i = 0; i < q.in.n; i++) SR3D_CHECK(q.in.ptr[i] != nullptr, SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
    q.K = d->Cin, q.Z = d->Z, q.Y = d->Y, q.X = d->X;
    q.N = hconv_fwd_rows(d, SR3D_PACK_FWD_GATED), q.Cg = d->Cout;
    q.act = act, q.epi = SR3D_EPI_GATED;
    q.bias = (const float*)bias_f, q.bias2 = (const float*)bias_g;
    q.y = (float*)y, q.save_f = (float*)save_f, q.save_s = (float*)save_s;
    q.TZ_ = d->Z, q.TY_ = d->Y, q.TX_ = d->X;
    q.amax_out = is_bf(d) ? nullptr : (unsigned*)x_absmax;
    return sr3d_hconv_launch(q, w_packed, d->B, is_bf(d), (hipStream_t)stream);
  }
  if (use_wino(d)) {
    SrWinoParams q{};
    if (int rc = sr3d_make_cat(x_srcs, n_src, (long long)d->Z * d->Y * d->X, d->Cin, &q.in, "x_srcs")) return rc;
    for (int i = 0; i < q.in.n; i++) SR3D_CHECK(q.in.ptr[i] != nullptr, SR3D_E_ARG, "x_srcs[%d].ptr is null", i);
    q.K = d->Cin, q.Z = d->Z, q.Y = d->Y, q.X = d->X;
    q.N = wino_fwd_rows(d, SR3D_PACK_FWD_GATED), q.Cg = d->Cout;
    q.up = (const float*)w_packed, q.act = act, q.epi = SR3D_EPI_GATED;
    q.bias = (const float*)bias_f, q.bias2 = (const float*)bias_g;
    q.y = (float*)y, q.save_f = (float*)save_f, q.save_s = (float*)save_s;
    q.TZ_ = d->Z, q.TY_ = d->Y, q.TX_ = d->X;
    return sr3d_wino_launch(q, d->B, (hipStream_t)stream);
  }
  IgemmParams p{};
  p.epi = EPI_GATED;
  p.act = act;
  p.bias = (const float*)bias_f, p.bias2 = (const float*)bias_g;
  p.y = (float*)y, p.save_f = (float*)save_f, p.save_s = (float*)save_s;
  p.N = fwd_rows(d, SR3D_PACK_FWD_GATED);  // GEMM rows: feature/gate interleaved in blocks of 32
  p.Cg = d->Cout;
  return forward_common(d, x_srcs, n_src, p, 1, (const float*)w_packed, (hipStream_t)stream, x_absmax);
}

// rows of the backward GEMM = input channels that need a gradient (slices with a null ptr are skipped)
static int bwd_rows(const sr3d_slice_t* dx_dsts, int n_dst) {
  int n = 0;
  for (int i = 0; i < n_dst; i++)
    if (dx_dsts[i].ptr) n += dx_dsts[i].channels;
  return n;
}
// 1..4 gradient rows beyond a multiple of 64 that take the small-N VALU kernel instead of a padded MFMA tile (they are the
// last channels of the last destination slice)
static int bwd_rem_rows(const sr3d_conv_desc_t* d, int rows, int last_channels) {
  const int rem = rows % 64;
  if (!(rows > 64 && rem >= 1 && rem <= 4 && last_channels >= rem && (long long)d->Z * d->Y * d->X >= 500000) || is_bf(d)) return 0;
  return rem;
}
// the fused activation epilogue exists in the split-f16 / bf16 stride-1 kernel only, and not for a slice whose last
// channels leave for the VALU remainder kernel
static bool bwd_data_fuses_act(const sr3d_conv_desc_t* d, int n_dy, const sr3d_slice_t* dx_dsts, int n_dst, int act_slice, bool unsh = false) {
  if (check_desc(d) != SR3D_OK || d->stride != 1 || act_slice < 0 || act_slice >= n_dst || dx_dsts == nullptr) return false;
  if (dx_dsts[act_slice].ptr == nullptr) return false;
  if (unsh) {   // the producer's shuffled layout is written by the 16-byte epilogues only (MFMA kernel and VALU remainder kernel)
    if (d->X % 4 != 0 || d->Y % 2 != 0 || d->Z % 2 != 0 || getenv("SR3D_HCONV_SCALAR_EPILOGUE") != nullptr) return false;
    for (int i = 0; i < n_dst; i++)
      if (reinterpret_cast<uintptr_t>(dx_dsts[i].ptr) & 15) return false;
    return use_hconv(d, n_dy * d->Cout, bwd_rows(dx_dsts, n_dst));
  }
  const int rows = bwd_rows(dx_dsts, n_dst);
  if (!use_hconv(d, n_dy * d->Cout, rows)) return false;
  int last = -1;
  for (int i = 0; i < n_dst; i++)
    if (dx_dsts[i].ptr) last = i;
  // (remainder rows of the fused slice take the VALU kernel's fused epilogue: 4 x-neighbours per thread)
  return !(last == act_slice && bwd_rem_rows(d, rows, dx_dsts[last].channels) > 0 && d->X % 4 != 0);
}

size_t sr3d_conv3d_bwd_data_workspace_bytes(const sr3d_conv_desc_t* d, int n_dy) {
  if (check_desc(d) != SR3D_OK || (n_dy != 1 && n_dy != 2)) return 0;
  // upper bound: every input channel needs a gradient; stride 2 stores the 8 parity-class images (27 taps in total)
  const size_t direct = image_floats(ceil_div(d->Cin, 32), ceil_div(n_dy * d->Cout, kKC), 27) * 4;
  // (+ the [K][27][4] image of up to 4 remainder rows that take the VALU kernel)
  const size_t wino = use_wino(d) ? sr3d_wino_image_floats(d->Cin, n_dy * d->Cout) * 4 + (size_t)n_dy * d->Cout * 108 * 4 : 0;
  const size_t hc = use_hconv(d, n_dy * d->Cout, d->Cin)
                        ? ((sr3d_hconv_image_bytes(d->Cin, n_dy * d->Cout, is_bf(d)) + 255) & ~(size_t)255) + (size_t)n_dy * d->Cout * 108 * 4
                        : 0;
  const size_t hs2 = use_hconv_s2(d, n_dy * d->Cout, d->Cin, true) ? sr3d_hconv_s2_image_bytes(d->Cin, n_dy * d->Cout, is_bf(d)) : 0;
  size_t m = direct > wino ? direct : wino;
  m = m > hc ? m : hc;
  return m > hs2 ? m : hs2;
}

// the input gradient; act_slice >= 0: destination slice `act_slice` (index into dx_dsts) receives result * lrelu'(act_y)
// (sr3d_conv3d_bwd_data_act) -- only the split-f16 / bf16 stride-1 kernel has that epilogue
static int bwd_data_impl(const sr3d_conv_desc_t* d, const sr3d_slice_t* dy_srcs, int n_dy, const void* w_feat,
                         const void* w_gate, const sr3d_slice_t* dx_dsts, int n_dst, void* workspace,
                         size_t workspace_bytes, void* stream, int act_slice, const void* act_y, void* act_absmax, bool act_unsh = false) {
  if (int rc = check_desc(d)) return rc;
  SR3D_CHECK(w_feat && workspace && dx_dsts, SR3D_E_ARG, "conv3d_bwd_data: null pointer");
  SR3D_CHECK(n_dy == 1 || (n_dy == 2 && w_gate), SR3D_E_ARG, "conv3d_bwd_data: n_dy must be 1, or 2 with w_gate");
  SR3D_CHECK(n_dst >= 1 && n_dst <= SR3D_MAX_SRC, SR3D_E_ARG, "conv3d_bwd_data: need 1..4 destination slices");
  SR3D_CHECK(workspace_bytes >= sr3d_conv3d_bwd_data_workspace_bytes(d, n_dy), SR3D_E_WORKSPACE,
             "conv3d_bwd_data: workspace of %zu bytes is too small", workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  const int OZ = out_dim(d->Z, d->stride), OY = out_dim(d->Y, d->stride), OX = out_dim(d->X, d->stride);
  const int K = n_dy * d->Cout;
  // compacted destination list + row -> channel map
  sr3d_slice_t need[SR3D_MAX_SRC];
  PackParams pk{};
  int nn = 0, ch = 0, rows = 0, act_nn = -1;
  for (int i = 0; i <= SR3D_MAX_SRC; i++) pk.rbeg[i] = INT_MAX;
  for (int i = 0; i < n_dst; i++) {
    SR3D_CHECK(dx_dsts[i].channels > 0, SR3D_E_ARG, "dx_dsts[%d]: channels must be positive", i);
    if (dx_dsts[i].ptr) {
      if (i == act_slice) act_nn = nn;
      need[nn] = dx_dsts[i];
      pk.rbeg[nn] = rows, pk.cbeg[nn] = ch;
      rows += dx_dsts[i].channels;
      nn++;
    }
    ch += dx_dsts[i].channels;
  }
  SR3D_CHECK(ch == d->Cin, SR3D_E_ARG, "dx_dsts: slices hold %d channels, the layer has %d", ch, d->Cin);
  if (rows == 0) return SR3D_OK;
  SR3D_CHECK(rows == bwd_rows(dx_dsts, n_dst), SR3D_E_ARG, "internal: row count");

  IgemmParams p{};
  if (int rc = sr3d_make_cat(dy_srcs, n_dy, (long long)OZ * OY * OX, K, &p.in, "dy_srcs")) return rc;
  for (int i = 0; i < p.in.n; i++) SR3D_CHECK(p.in.ptr[i] != nullptr, SR3D_E_ARG, "dy_srcs[%d].ptr is null", i);
  if (int rc = sr3d_make_cat(need, nn, (long long)d->Z * d->Y * d->X, rows, &p.out, "dx_dsts")) return rc;
  p.K = K, p.N = rows, p.nchunks = ceil_div(K, kKC);
  p.IZ = OZ, p.IY = OY, p.IX = OX;
  p.TZ_ = d->Z, p.TY_ = d->Y, p.TX_ = d->X;
  p.epi = EPI_PLAIN, p.act = SR3D_ACT_NONE;
  const RowPlan rp = row_plan(rows, d->stride == 1 ? tiles_of(d->B, d->Z, d->Y, d->X, 2, 4)
                                                   : tiles_of(d->B, (d->Z + 1) / 2, (d->Y + 1) / 2, (d->X + 1) / 2, 2, 4),
                              false);

  pk.w1 = (const float*)w_feat, pk.w2 = (const float*)w_gate;
  pk.Cout = d->Cout, pk.Cin = d->Cin, pk.kind = n_dy == 2 ? SR3D_PACK_BWD_GATED : SR3D_PACK_BWD;
  pk.K = K, pk.N = rows, pk.nchunks = p.nchunks;
  float* image = (float*)workspace;

  const bool hconv = use_hconv(d, K, rows);
  SR3D_CHECK(act_slice < 0 || (act_nn >= 0 && act_y != nullptr && bwd_data_fuses_act(d, n_dy, dx_dsts, n_dst, act_slice, act_unsh)), SR3D_E_ARG,
             "conv3d_bwd_data_act: this launch has no fused activation epilogue (ask sr3d_conv3d_bwd_data_fuses_act first)");
  if (hconv || (use_wino(d) && K <= SR3D_WINO_MAX_K)) {
    // 1..4 gradient rows beyond a multiple of 64 (e.g. 193 = 3 * 64 + 1) would cost a whole 32-row Winograd tile per
    // voxel block: when they are the last channels of the last destination slice they take the small-N VALU kernel
    const sr3d_slice_t& last = need[nn - 1];
    // (on the small grids of the deep levels the extra launches cost more than the padded tile)
    const int rem = bwd_rem_rows(d, rows, last.channels);
    const int main_rows = rows - rem;
    float* wsm = nullptr;   // image of the remainder rows
    if (hconv) {
      SrHconvParams q{};
      q.in = p.in, q.out = p.out;
      q.K = K, q.N = main_rows, q.Z = d->Z, q.Y = d->Y, q.X = d->X;
      q.TZ_ = d->Z, q.TY_ = d->Y, q.TX_ = d->X;
      q.epi = SR3D_EPI_PLAIN, q.act = SR3D_ACT_NONE;
      if (act_nn >= 0) q.act_slice1 = act_nn + 1, q.act_y = act_y, q.act_amax = is_bf(d) ? nullptr : (unsigned*)act_absmax, q.act_unsh = act_unsh ? 1 : 0;
      if (int rc = sr3d_hconv_pack(pk.kind, d->Cout, d->Cin, main_rows, K, pk.w1, pk.w2, pk.rbeg, pk.cbeg, image, is_bf(d), st)) return rc;
      if (int rc = sr3d_hconv_launch(q, image, d->B, is_bf(d), st)) return rc;
      wsm = (float*)((unsigned char*)image + ((sr3d_hconv_image_bytes(main_rows, K, is_bf(d)) + 255) & ~(size_t)255));
    } else {
      SrWinoParams q{};
      q.in = p.in, q.out = p.out;
      q.K = K, q.N = main_rows, q.Z = d->Z, q.Y = d->Y, q.X = d->X;
      q.TZ_ = d->Z, q.TY_ = d->Y, q.TX_ = d->X;
      q.epi = SR3D_EPI_PLAIN, q.act = SR3D_ACT_NONE, q.up = image;
      if (int rc = sr3d_wino_pack(pk.kind, d->Cout, d->Cin, main_rows, K, pk.w1, pk.w2, pk.rbeg, pk.cbeg, image, st)) return rc;
      if (int rc = sr3d_wino_launch(q, d->B, st)) return rc;
      wsm = image + sr3d_wino_image_floats(main_rows, K);
    }
    if (rem > 0) {
      const int ci0 = pk.cbeg[nn - 1] + last.channels - rem;   // input channel of the first remainder row
      hipLaunchKernelGGL(pack_smalln_bwd_kernel, dim3(ceil_div(K * 108, 256)), dim3(256), 0, st, pk.w1, pk.w2, wsm, d->Cout,
                         d->Cin, K, ci0, rem);
      SR3D_HIP(hipGetLastError());
      SmallFwdParams sq{};
      sq.in = p.in, sq.K = K, sq.N = rem, sq.Z = d->Z, sq.Y = d->Y, sq.X = d->X;
      sq.ntz = ceil_div(d->Z, 4), sq.nty = ceil_div(d->Y, 8), sq.ntx = ceil_div(d->X, 32);
      const long long vox = (long long)d->Z * d->Y * d->X;
      sq.w = wsm, sq.bias = nullptr, sq.act = SR3D_ACT_NONE;
      sq.y = (float*)last.ptr + (long long)(last.channels - rem) * vox, sq.y_bstride = (long long)last.channels * vox;
      if (act_nn == nn - 1) {   // the remainder rows belong to the activation-fused slice
        sq.act_y = (const float*)act_y + (long long)(last.channels - rem) * vox;
        sq.act_amax = (unsigned*)act_absmax;
        if (act_unsh) sq.y = (float*)last.ptr, sq.unsh_C = last.channels, sq.unsh_c0 = last.channels - rem;
      }
      SR3D_CHECK(d->B <= 65535, SR3D_E_ARG, "conv3d_bwd_data: batch too large");
      launch_smalln_fwd(sq, d->B, st);
      SR3D_HIP(hipGetLastError());
    }
    return SR3D_OK;
  }
  if (d->stride == 1) {
    pk.ntaps = 27;
    for (int t = 0; t < 27; t++) pk.tap[t] = t;
    if (int rc = run_pack(pk, rp, image, st)) return rc;
    p.OZ = d->Z, p.OY = d->Y, p.OX = d->X;
    p.s_out = 1;
    using C = IgemmCfg<1, -1, 1, 2, 4, 4, kKC>;
    full_taps(p, C::HY, C::HX, true);
    return launch<1, -1, 1, 2, 4>(p, d->B, rp, image, st);
  }
  if (use_hconv_s2(d, K, rows, true)) {
    SrHconvS2Params q{};
    q.in = p.in, q.out = p.out, q.K = K, q.N = rows, q.n_off = 0;
    q.IZ = OZ, q.IY = OY, q.IX = OX;
    q.Z = d->Z, q.Y = d->Y, q.X = d->X;
    q.TZ_ = d->Z, q.TY_ = d->Y, q.TX_ = d->X;
    q.epi = SR3D_EPI_PLAIN, q.act = SR3D_ACT_NONE;
    const int s2mode = sr3d_hconv_s2_bwd_paired(q, is_bf(d)) ? 4 : 2;   // 4: both x classes per workgroup (quad loads possible)
    if (int rc = sr3d_hconv_s2_pack(s2mode, pk.kind, d->Cout, d->Cin, rows, K, pk.w1, pk.w2, pk.rbeg, pk.cbeg, image, is_bf(d), st)) return rc;
    return sr3d_hconv_s2_launch(s2mode, q, image, d->B, is_bf(d), st);
  }
  using C = IgemmCfg<1, 0, 1, 2, 4, 4, kKC>;
  // the 8 output-parity classes: packed one after the other, then ONE launch with blockIdx.z = class
  p.ncls = 8, p.s_out = 2;
  long long off = 0;
  for (int cls = 0; cls < 8; cls++) {
    IgemmParams::Cls& c = p.cls[cls];
    int dz[27], dy[27], dx[27];
    pk.ntaps = c.ntaps = class_taps(cls, pk.tap, dz, dy, dx);
    for (int t = 0; t < c.ntaps; t++) c.tap_off[t] = (dz[t] * C::HY + dy[t]) * C::HX + dx[t];
    c.pz = (cls >> 2) & 1, c.py = (cls >> 1) & 1, c.px = cls & 1;
    c.OZ = (d->Z - c.pz + 1) / 2, c.OY = (d->Y - c.py + 1) / 2, c.OX = (d->X - c.px + 1) / 2;
    c.wp_off = off;
    if (c.OZ > 0 && c.OY > 0 && c.OX > 0)
      if (int rc = run_pack(pk, rp, image + off, st)) return rc;
    off += (long long)image_floats(rp.units, p.nchunks, c.ntaps);
  }
  // grid of class 0 (even positions: the largest); ntaps of the launch-wide fields is only used for sizing
  p.OZ = p.cls[0].OZ, p.OY = p.cls[0].OY, p.OX = p.cls[0].OX, p.ntaps = 8;
  return launch<1, 0, 1, 2, 4>(p, d->B, rp, image, st);
}

int sr3d_conv3d_bwd_data(const sr3d_conv_desc_t* d, const sr3d_slice_t* dy_srcs, int n_dy, const void* w_feat,
                         const void* w_gate, const sr3d_slice_t* dx_dsts, int n_dst, void* workspace,
                         size_t workspace_bytes, void* stream) {
  return bwd_data_impl(d, dy_srcs, n_dy, w_feat, w_gate, dx_dsts, n_dst, workspace, workspace_bytes, stream, -1, nullptr, nullptr);
}

int sr3d_conv3d_bwd_data_fuses_act(const sr3d_conv_desc_t* d, int n_dy, const sr3d_slice_t* dx_dsts, int n_dst, int act_slice, int act) {
  if ((act & ~SR3D_ACT_UNSHUFFLE) != SR3D_ACT_LRELU) return 0;
  return bwd_data_fuses_act(d, n_dy, dx_dsts, n_dst, act_slice, (act & SR3D_ACT_UNSHUFFLE) != 0) ? 1 : 0;
}

int sr3d_conv3d_bwd_data_act(const sr3d_conv_desc_t* d, const sr3d_slice_t* dy_srcs, int n_dy, const void* w_feat,
                             const void* w_gate, const sr3d_slice_t* dx_dsts, int n_dst, int act_slice, const void* act_y,
                             int act, void* act_absmax, void* workspace, size_t workspace_bytes, void* stream) {
  const bool unsh = (act & SR3D_ACT_UNSHUFFLE) != 0;
  act &= ~SR3D_ACT_UNSHUFFLE;
  SR3D_CHECK(act == SR3D_ACT_LRELU, SR3D_E_ARG, "conv3d_bwd_data_act: the fused epilogue is LeakyReLU'(y) (got act %d)", act);
  SR3D_CHECK(act_slice >= 0 && act_slice < n_dst && act_y != nullptr, SR3D_E_ARG, "conv3d_bwd_data_act: bad act_slice / act_y");
  return bwd_data_impl(d, dy_srcs, n_dy, w_feat, w_gate, dx_dsts, n_dst, workspace, workspace_bytes, stream, act_slice, act_y, act_absmax, unsh);
}

}  // extern "C"
