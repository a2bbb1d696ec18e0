// Error reporting + library identification for the C-ABI (include/openpvsg_hip.h).
#include "common.h"
#include <stdarg.h>

namespace pvsg {
static thread_local char g_err[512] = {0};
char* err_buf() { return g_err; }
int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace pvsg

extern "C" const char* pvsg_last_error(void) { return pvsg::err_buf(); }
extern "C" const char* pvsg_version(void) { return "openpvsg_amd-hip 0.1 (gfx950)"; }
extern "C" int pvsg_abi_version(void) { return 11; }

// ---- host-side codec of the result formats (no device work) ----------------------------------------------------------------
// COCO compressed run-length strings of MANY masks at once ([3P] pycocotools rleToString: the 4th and later counts delta-coded
// against counts[i-2], 5 bits per character least significant first, 0x20 = more follow, offset 48).  counts: the run lengths of
// all masks back to back, seg[j] of them belong to mask j; out: the characters of all masks back to back (13 bytes per count are
// always enough), out_len[j] of them belong to mask j.  Returns the total number of characters, -1 on bad arguments.
extern "C" long long pvsg_rle_counts_to_chars(const long long* counts, const long long* seg, int nseg, unsigned char* out,
                                              long long* out_len) {
  if (nseg < 0 || (nseg > 0 && (!counts || !seg || !out || !out_len))) {
    pvsg::set_err(PVSG_ERR_INVALID_ARG, "pvsg_rle_counts_to_chars: null pointer or negative mask count");
    return -1;
  }
  long long p = 0, o = 0;
  for (int j = 0; j < nseg; ++j) {
    const long long n = seg[j], o0 = o;
    for (long long i = 0; i < n; ++i) {
      long long x = counts[p + i];
      if (i > 2) x -= counts[p + i - 2];
      bool more = true;
      while (more) {
        unsigned char c = (unsigned char)(x & 0x1f);
        x >>= 5;                                             // arithmetic: negative deltas end at -1
        more = (c & 0x10) ? x != -1 : x != 0;
        if (more) c |= 0x20;
        out[o++] = (unsigned char)(c + 48);
      }
    }
    out_len[j] = o - o0;
    p += n;
  }
  return o;
}
