// Shared device/host declarations for libsr3d (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sr3d.h"

#define SR3D_MAX_SRC 4
// most GEMM-K channels the Winograd stride-1 kernel takes (its per-workgroup channel-pointer table lives in LDS);
// the model's largest is 2056 (input gradient of up3.up / up4.up).  Beyond it the direct kernel runs.
#define SR3D_WINO_MAX_K 2560
#define SR3D_MAX_TAPS 27

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A virtual channel-concatenation of up to 4 NCDHW fp32 tensors that share B and
// the spatial grid.  cbeg[i] .. cbeg[i+1] are the concat channels of tensor i;
// unused entries have cbeg = INT_MAX.  ptr may be null for a write destination
// that does not need a gradient.
struct ChanCat {
  float* ptr[SR3D_MAX_SRC];
  long long bstride[SR3D_MAX_SRC];  // elements between batch samples (= channels * Z*Y*X)
  int cbeg[SR3D_MAX_SRC + 1];
  int n;
};

void sr3d_set_error(const char* fmt, ...);
// p[0 .. nwords) = 0 by a kernel launch on `st` (never a memset node inside a captured graph: sr3d_api.hip)
int sr3d_zero_words(void* p, int nwords, hipStream_t st);



#define SR3D_CHECK(cond, code, ...)        \
  do {                                     \
    if (!(cond)) {                         \
      sr3d_set_error(__VA_ARGS__);         \
      return (code);                       \
    }                                      \
  } while (0)

#define SR3D_HIP(call)                                                        \
  do {                                                                        \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess) {                                                   \
      sr3d_set_error("%s failed: %s", #call, hipGetErrorString(e_));          \
      return SR3D_E_HIP;                                                      \
    }                                                                         \
  } while (0)

// conv epilogues shared by the direct (sr3d_igemm.hip) and Winograd (sr3d_wino.hip) kernels
enum { SR3D_EPI_PLAIN = 0, SR3D_EPI_GATED = 1, SR3D_EPI_UNSHUFFLE = 2 };

// launch description of the Winograd stride-1 conv (sr3d_wino.hip); filled by the entry points in sr3d_igemm.hip
struct SrWinoParams {
  ChanCat in;          // K side (virtual concat)
  int K;               // input channels
  int Z, Y, X;         // grid (stride 1: input grid = output grid)
  int ntz, nty, ntx, nblk, nchunks;   // set by sr3d_wino_launch
  int nb_off, nimg;    // first 64-row block of this launch; blocks in the packed image
  int rt_split;        // one-tile launch with one workgroup per 32-row tile
  const float* up;     // transformed + packed weights
  int N;               // GEMM rows (gated: 32 per 16 channels)
  int n_off;
  int epi, act;
  const float* bias;
  const float* bias2;
  ChanCat out;         // plain epilogue destinations
  float* y;
  float* save_f;
  float* save_s;
  int TZ_, TY_, TX_;   // destination grid (2x for the unshuffle epilogue)
  int unsh_C, Cg;
  int pair_aligned;    // every destination pointer is 8-byte aligned
};
bool sr3d_wino_enabled();
size_t sr3d_wino_image_floats(int rows, int K);
int sr3d_wino_pack(int kind, int Cout, int Cin, int rows, int K, const float* w1, const float* w2, const int* rbeg,
                   const int* cbeg, float* image, hipStream_t st);
int sr3d_wino_launch(SrWinoParams& p, int B, hipStream_t st);
// launch description of the split-f16 stride-1 conv (sr3d_hconv.hip)
struct SrHconvParams {
  ChanCat in;          // K side (virtual concat)
  int K;               // input channels
  int Z, Y, X;
  int ntz, nty, ntx, nblk, nchunks;   // set by sr3d_hconv_launch
  int nb_off;          // first row block of this launch inside its image region
  const void* wimg;    // split + packed weights (region base)
  const float* absmax_w;   // device: max |w| (header of the packed image)
  int N;               // GEMM rows (gated: 64 per 32 channels)
  int n_off;
  int epi, act;
  const float* bias;
  const float* bias2;
  ChanCat out;
  float* y;
  float* save_f;
  float* save_s;
  int TZ_, TY_, TX_;
  int unsh_C, Cg;
  unsigned* amax_out;  // optional [SR3D_MAX_SRC][64]: max |x| per K-side slice, a by-product of the block scaling (fp32 only)
  // plain epilogue, input gradient: destination slice `act_slice1 - 1` (index into `out`) is the output y of a LeakyReLU layer;
  // what is stored there is result * lrelu'(y) = that layer's dL/dpre (sr3d_conv3d_bwd_data_act), and max |stored| goes to
  // act_amax[64] (optional, fp32: the scale of that layer's split-f16 weight gradient)
  int vec_epi;         // set by sr3d_hconv_launch: X % 4 == 0 and every destination / y / save pointer 16-byte aligned -> the plain
                       // and gated epilogues transpose their 16 x 16 tiles through LDS and store 4 x-neighbours at a time
  int itail;           // set by sr3d_hconv_launch: the last chunk (1 .. 5 channels) runs in im2col form (sr3d_hconv.hip)
  int out_f32;         // bf16 storage, plain epilogue: the destinations are fp32 tensors (the network's prediction: `last`)
  const void* act_y;
  int act_unsh;        // the fused slice is stored in the producer's SHUFFLED layout (8 C channels on the coarse grid); vec_epi only
  int act_slice1;      // 1 + slice index; 0 (a zero-initialised launch description): none
  unsigned* act_amax;
};
int sr3d_hconv_mode();   // SR3D_SPLIT_F16: 0 off, 1 auto (default), 2 always
// (bf: activations stored as bfloat16, one bf16 MFMA per product; sr3d_conv_desc_t.dtype == SR3D_DTYPE_BF16)
size_t sr3d_hconv_image_bytes(int rows, int K, bool bf = false);
// (unsh_C > 0, plain forward only: rows in voxel-unshuffle order -- the image header tells the kernel)
int sr3d_hconv_pack(int kind, int Cout, int Cin, int rows, int K, const float* w1, const float* w2, const int* rbeg,
                    const int* cbeg, void* image, bool bf, hipStream_t st, int unsh_C = 0);
int sr3d_hconv_launch(SrHconvParams& p, const void* image, int B, bool bf, hipStream_t st);
// split-f16 stride-2 conv over parity classes (sr3d_hconv_s2.hip): mode 1 = forward, mode 2 = input gradient
struct SrHconvS2Params {
  ChanCat in;          // K side (virtual concat): x (mode 1) or dy (mode 2)
  int K;
  int IZ, IY, IX;      // grid of the K-side tensors
  int Z, Y, X;         // mode 1: output grid; mode 2: grid of dx (the per-class tile spaces are cZ/cY/cX)
  int ntz, nty, ntx, nblk, nchunks;   // set by sr3d_hconv_s2_launch
  int nb_off;
  const void* wimg;
  const float* absmax_w;
  int N;
  int n_off;
  int epi, act;
  const float* bias;
  const float* bias2;
  ChanCat out;
  float* y;
  float* save_f;
  float* save_s;
  int TZ_, TY_, TX_;   // destination tensor grid
  int Cg;
  long long blk_stride;    // mode 1: bytes of one row block of the image
  long long cls_off[8];    // mode 2: byte offset of the class image, bytes of one of its row blocks
  long long cls_blk[8];
  int cZ[8], cY[8], cX[8];
  unsigned* amax_out;  // mode 1, optional [SR3D_MAX_SRC][64]: max |x| per K-side slice (as SrHconvParams.amax_out; fp32 only)
  int itail;           // paired forward, set by the launch: the last chunk (1 - 2 channels) in im2col form
};
int sr3d_absmax_launch(const float* x, long long n, unsigned* slot, hipStream_t st);   // max |x| into *slot (sr3d_hconv.hip)
size_t sr3d_hconv_s2_image_bytes(int rows, int K, bool bf = false);
bool sr3d_hconv_s2_bwd_paired(const SrHconvS2Params& p, bool bf);   // input gradient: pack order / launch mode 4 instead of 2
bool sr3d_hconv_s2_fwd_paired(int IX);   // forward image in pair order (pack mode 3) / paired kernel: X % 4 == 0
int sr3d_hconv_s2_pack(int mode, int kind, int Cout, int Cin, int rows, int K, const float* w1, const float* w2,
                       const int* rbeg, const int* cbeg, void* image, bool bf, hipStream_t st);
int sr3d_hconv_s2_launch(int mode, SrHconvS2Params& p, const void* image, int B, bool bf, hipStream_t st);
// split-f16 weight gradient of the stride-1 layers (sr3d_hwgrad.hip); same contract as sr3d_wino_wgrad
size_t sr3d_hwgrad_ws_bytes(const sr3d_conv_desc_t* d, int n_total, int c_used);
bool sr3d_hwgrad_ok(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy);
// x_absmax / dy_absmax (optional): [slice][64] maxima exported by the forward kernel / the activation-backward kernels
int sr3d_hwgrad(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy, int n_total, int c_used, float* dw, float* ws,
                hipStream_t st, const unsigned* x_absmax = nullptr, const unsigned* dy_absmax = nullptr);
// amax[i] (x slices 0..3) / amax[4 + i] (dy slices) = max over the 64 hashed slots another kernel exported (sr3d_hwgrad.hip)
int sr3d_gather_absmax(const unsigned* x_absmax, int nx, const unsigned* dy_absmax, int nd, unsigned* amax, hipStream_t st);
// few-channel (Cin <= 5) stride-1 weight gradient on the f16 MFMA (sr3d_hwgrad_fc.hip), fp32 storage
size_t sr3d_hwgrad_fc_ws_bytes(const sr3d_conv_desc_t* d, int n_total, bool swapped = false);
bool sr3d_hwgrad_fc_ok(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy, int n_total, bool swapped = false);
// swapped: few OUTPUT rows (n_total <= 5, `last`) instead of few input channels: the roles of x and dY exchanged, taps mirrored
int sr3d_hwgrad_fc(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy, int n_total, float* dw, float* ws, hipStream_t st,
                   const unsigned* x_absmax, const unsigned* dy_absmax, bool swapped = false);
// stride-2 weight gradient on the f16 / bf16 MFMA (sr3d_hwgrad_s2.hip); all d->Cin channels
size_t sr3d_hwgrad_s2_ws_bytes(const sr3d_conv_desc_t* d, int n_total);
bool sr3d_hwgrad_s2_ok(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy);
int sr3d_hwgrad_s2(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy, int n_total, float* dw, float* ws, hipStream_t st,
                   const unsigned* x_absmax = nullptr, const unsigned* dy_absmax = nullptr);
// Winograd-domain weight gradient (sr3d_wino_wgrad.hip)
// (the first `c_used` input channels; dW rows keep their full length d->Cin * 27)
size_t sr3d_wino_wgrad_ws_bytes(const sr3d_conv_desc_t* d, int n_total, int c_used);
int sr3d_wino_wgrad(const sr3d_conv_desc_t* d, const ChanCat& x, const ChanCat& dy, int n_total, int c_used, float* dw,
                    float* ws, hipStream_t st);
// VALU weight gradient for a few channels on one side (sr3d_wgrad_few.hip)
size_t sr3d_wgrad_few_ws_bytes(const sr3d_conv_desc_t* d, int M, int few_n);
int sr3d_wgrad_few(const sr3d_conv_desc_t* d, const ChanCat& many, int M, const ChanCat& few, int few_c0, int few_n,
                   float* dw, long long ldw, float* ws, hipStream_t st);

// per-kernel HIP-event timing (off unless sr3d_profile_enable(1)); ids are SR3D_PROF_*
bool sr3d_prof_active();
void sr3d_prof_begin(int id, double flops, hipStream_t st, void** token);
void sr3d_prof_end(void* token, hipStream_t st);
// brackets everything launched on `st` during its lifetime (work: FLOPs or bytes, see include/sr3d.h)
struct SrProfScope {
  void* tok;
  hipStream_t st;
  SrProfScope(int id, double work, hipStream_t s) : tok(nullptr), st(s) {
    if (sr3d_prof_active()) sr3d_prof_begin(id, work, s, &tok);
  }
  ~SrProfScope() { sr3d_prof_end(tok, st); }
  SrProfScope(const SrProfScope&) = delete;
  SrProfScope& operator=(const SrProfScope&) = delete;
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Activations are stored as fp32 or as bfloat16 (sr3d_conv_desc_t.dtype).  Kernels that only move or reduce them, and
// the fp32-MFMA fallback kernels, are templated on the STORAGE type and compute in fp32: bf16raw is the 16-bit pattern.
typedef unsigned short bf16raw;
template <typename T> struct ActIo;
template <> struct ActIo<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }   // 16-byte aligned
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct ActIo<bf16raw> {
  static __device__ __forceinline__ float ld(const bf16raw* p) { return __builtin_bit_cast(float, (unsigned)*p << 16); }
  static __device__ __forceinline__ f32x4 ld4(const bf16raw* p) {   // 8-byte aligned
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return f32x4{__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                 __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
  }
  static __device__ __forceinline__ bf16raw rne(float v) { return __builtin_bit_cast(bf16raw, (__bf16)v); }
  static __device__ __forceinline__ void st(bf16raw* p, float v) { *p = rne(v); }
  static __device__ __forceinline__ void st4(bf16raw* p, f32x4 v) {
    uint2 u;
    u.x = (unsigned)rne(v.x) | ((unsigned)rne(v.y) << 16);
    u.y = (unsigned)rne(v.z) | ((unsigned)rne(v.w) << 16);
    *reinterpret_cast<uint2*>(p) = u;
  }
};

// One-time launch setup that is a property of the DEVICE (hipFuncSetAttribute: dynamic LDS above 64 KB), keyed on the
// current device: a process may run the engine on cuda:1 after cuda:0 from the same thread.
#include <atomic>
#include <mutex>
class SrPerDevice {
 public:
  template <typename F>
  int once(F&& setup) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done_.load(std::memory_order_acquire) & bit) return SR3D_OK;
    std::lock_guard<std::mutex> lk(mu_);
    if (done_.load(std::memory_order_relaxed) & bit) return SR3D_OK;
    if (int rc = setup()) return rc;
    done_.fetch_or(bit, std::memory_order_release);
    return SR3D_OK;
  }

 private:
  std::mutex mu_;
  std::atomic<unsigned long long> done_{0};
};

int sr3d_make_cat(const sr3d_slice_t* s, int n, long long vox, int expect_channels, ChanCat* out, const char* what);

__device__ __forceinline__ int cat_find(const ChanCat& c, int ch) {
  return (ch >= c.cbeg[1]) + (ch >= c.cbeg[2]) + (ch >= c.cbeg[3]);
}
__device__ __forceinline__ float* cat_ptr(const ChanCat& c, int si) {
  float* p = c.ptr[0];
  p = si == 1 ? c.ptr[1] : p;
  p = si == 2 ? c.ptr[2] : p;
  p = si == 3 ? c.ptr[3] : p;
  return p;
}
__device__ __forceinline__ long long cat_bstride(const ChanCat& c, int si) {
  long long p = c.bstride[0];
  p = si == 1 ? c.bstride[1] : p;
  p = si == 2 ? c.bstride[2] : p;
  p = si == 3 ? c.bstride[3] : p;
  return p;
}
__device__ __forceinline__ int cat_cbeg(const ChanCat& c, int si) {
  int p = c.cbeg[0];
  p = si == 1 ? c.cbeg[1] : p;
  p = si == 2 ? c.cbeg[2] : p;
  p = si == 3 ? c.cbeg[3] : p;
  return p;
}
