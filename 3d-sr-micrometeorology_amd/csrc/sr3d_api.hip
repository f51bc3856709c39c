// libsr3d: error plumbing and argument helpers shared by all entry points.
#include "sr3d_common.h"

#include <limits.h>
#include <stdarg.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void sr3d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int sr3d_make_cat(const sr3d_slice_t* s, int n, long long vox, int expect_channels, ChanCat* out, const char* what) {
  SR3D_CHECK(s != nullptr && n >= 1 && n <= SR3D_MAX_SRC, SR3D_E_ARG, "%s: need 1..%d slices (got %d)", what,
             SR3D_MAX_SRC, n);
  int c = 0;
  for (int i = 0; i <= SR3D_MAX_SRC; i++) out->cbeg[i] = INT_MAX;
  for (int i = 0; i < SR3D_MAX_SRC; i++) out->ptr[i] = nullptr, out->bstride[i] = 0;
  for (int i = 0; i < n; i++) {
    SR3D_CHECK(s[i].channels > 0, SR3D_E_ARG, "%s[%d]: channels must be positive", what, i);
    out->ptr[i] = (float*)s[i].ptr;
    out->bstride[i] = (long long)s[i].channels * vox;
    out->cbeg[i] = c;
    c += s[i].channels;
  }
  out->n = n;
  SR3D_CHECK(c == expect_channels, SR3D_E_ARG, "%s: slices hold %d channels, the layer expects %d", what, c,
             expect_channels);
  return SR3D_OK;
}

// ---- zeroing of the few control words (maxima, scale headers) in front of a launch ---------------------------------
// A KERNEL, not hipMemsetAsync: inside a captured training step (src/graph.py) a memset becomes a memset NODE of the
// hipGraph, and on this runtime (ROCm 7.2) the replay does not keep such a node ordered against the kernel nodes around
// it -- the maxima words of the split-f16 kernels were cleared too early or too late.  Kernel nodes
// of one captured stream replay in order.  SR3D_DEBUG_MEMSET_NODE=1 brings the memset back (tests/test_gpu_graph.py shows
// the divergence with it).
namespace {
__global__ void zero_words_kernel(unsigned* __restrict__ p, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u;
}
}  // namespace

int sr3d_zero_words(void* p, int nwords, hipStream_t st) {
  static const bool memset_node = getenv("SR3D_DEBUG_MEMSET_NODE") != nullptr && atoi(getenv("SR3D_DEBUG_MEMSET_NODE")) != 0;
  if (memset_node) {
    SR3D_HIP(hipMemsetAsync(p, 0, (size_t)nwords * 4, st));
    return SR3D_OK;
  }
  hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(64), 0, st, (unsigned*)p, nwords);
  SR3D_HIP(hipGetLastError());
  return SR3D_OK;
}

// ---- optional per-kernel timing with HIP events (bench.py roofline leg) ------------
// The event pairs are created by sr3d_profile_enable(1), i.e. OUTSIDE the timed region; a launch only records two
// of them.  When the pool is used up further launches are counted as dropped (bench.py reports it).
#include <mutex>
#include <vector>
namespace {
struct ProfRec { hipEvent_t a, b; double work; int id; bool used; };
constexpr int kProfPool = 16384;
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
int g_prof_next = 0;
long long g_prof_dropped = 0;
bool g_prof_on = false;
bool g_prof_dominant_only = false;   // sr3d_profile_enable(2): only the stride-1 conv families are bracketed
}  // namespace

bool sr3d_prof_active() { return g_prof_on; }

void sr3d_prof_begin(int id, double work, hipStream_t st, void** token) {
  *token = nullptr;
  if (!g_prof_on) return;
  if (g_prof_dominant_only && id != SR3D_PROF_HCONV && id != SR3D_PROF_HCONV_SMALL && id != SR3D_PROF_IGEMM_S1) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof_next >= (int)g_prof.size()) { g_prof_dropped++; return; }
  ProfRec* r = &g_prof[g_prof_next++];
  r->work = work, r->id = id, r->used = false;
  if (hipEventRecord(r->a, st) != hipSuccess) return;
  *token = r;
}

void sr3d_prof_end(void* token, hipStream_t st) {
  if (!token) return;
  ProfRec* r = (ProfRec*)token;
  r->used = hipEventRecord(r->b, st) == hipSuccess;
}

extern "C" {
int sr3d_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_next = 0, g_prof_dropped = 0;
  if (on && g_prof.empty()) {
    g_prof.reserve(kProfPool);
    for (int i = 0; i < kProfPool; i++) {
      ProfRec r{nullptr, nullptr, 0.0, -1, false};
      if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) break;
      g_prof.push_back(r);
    }
  }
  g_prof_on = on != 0;
  g_prof_dominant_only = on == 2;
  return SR3D_OK;
}

int sr3d_profile_read(int kernel_id, double* ms, double* work, long long* launches) {
  SR3D_CHECK(ms && work && launches, SR3D_E_ARG, "profile_read: null pointer");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double t = 0, f = 0; long long n = 0;
  if (kernel_id == SR3D_PROF_DROPPED) { *ms = 0, *work = 0, *launches = g_prof_dropped; return SR3D_OK; }
  for (int i = 0; i < g_prof_next; i++) {
    ProfRec& r = g_prof[i];
    if (r.id != kernel_id || !r.used) continue;
    SR3D_HIP(hipEventSynchronize(r.b));
    float e = 0.f;
    SR3D_HIP(hipEventElapsedTime(&e, r.a, r.b));
    t += e; f += r.work; n++;
  }
  *ms = t; *work = f; *launches = n;
  return SR3D_OK;
}

int sr3d_version(void) { return SR3D_VERSION; }
const char* sr3d_last_error(void) { return g_err; }
}
