// libsr3d: error plumbing and argument helpers shared by all entry points.
#include "sr3d_common.h"

#include <limits.h>
#include <stdarg.h>

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

extern "C" {
int sr3d_version(void) { return SR3D_VERSION; }
const char* sr3d_last_error(void) { return g_err; }
}
