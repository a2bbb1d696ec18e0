/* ORACLE (test infrastructure, never shipped): plain-C scalar restatement of the two kernels whose
 * arithmetic is spelled out element by element in the reference path.
 *
 *  oracle_msda_forward  -- [3P] mmcv-full 1.4.0 ms_deform_attn forward semantics (SURVEY.md Appendix
 *      A1 step 6): x = loc_x*W - 0.5, y = loc_y*H - 0.5, window (-1, size), per-corner zero padding,
 *      out += weight * bilinear.  "parity unpinned" like the rest of the third-party blocks; checked
 *      against F.grid_sample in tests/test_oracle_c.py.
 *  oracle_pair_score    -- models/relation_head/base.py:49-62: tokens = max over T, then for every
 *      i != j  pair[i][j] = w2 . relu(W1 [s_i ; o_j] + b1) + b2, diagonal 0.  Pinned by
 *      tests/golden/rel_*.npz (pred_matrix of the reference module).
 * Build: gcc -O2 -shared -fPIC oracle/c/ref_kernels.c -o oracle/_build/libpvsg_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

void oracle_msda_forward(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                         const float* w, float* out, int B, int S, int M, int D, int Lq, int L, int P) {
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Lq; ++q)
      for (int m = 0; m < M; ++m) {
        float* o = out + (((int64_t)b * Lq + q) * M + m) * D;
        for (int d = 0; d < D; ++d) o[d] = 0.f;
        for (int l = 0; l < L; ++l) {
          const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
          for (int p = 0; p < P; ++p) {
            const int64_t li = ((((int64_t)b * Lq + q) * M + m) * L + l) * P + p;
            const float x = loc[2 * li] * W - 0.5f, y = loc[2 * li + 1] * H - 0.5f;
            if (!(y > -1 && x > -1 && y < H && x < W)) continue;
            const int y0 = (int)floorf(y), x0 = (int)floorf(x);
            const float ly = y - y0, lx = x - x0, hy = 1 - ly, hx = 1 - lx;
            for (int d = 0; d < D; ++d) {
              float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
#define VAL(yy, xx) value[((((int64_t)b * S + lsi[l] + (int64_t)(yy) * W + (xx)) * M) + m) * D + d]
              if (y0 >= 0 && x0 >= 0) v1 = VAL(y0, x0);
              if (y0 >= 0 && x0 + 1 <= W - 1) v2 = VAL(y0, x0 + 1);
              if (y0 + 1 <= H - 1 && x0 >= 0) v3 = VAL(y0 + 1, x0);
              if (y0 + 1 <= H - 1 && x0 + 1 <= W - 1) v4 = VAL(y0 + 1, x0 + 1);
#undef VAL
              o[d] += w[li] * (hy * hx * v1 + hy * lx * v2 + ly * hx * v3 + ly * lx * v4);
            }
          }
        }
      }
}

void oracle_pair_score(const float* sub, const float* obj, const float* W1, const float* b1,
                       const float* w2, const float* b2, float* out, int N, int T, int C, int Hd) {
  float* st = (float*)malloc(sizeof(float) * (size_t)N * C * 2);
  float* ot = st + (size_t)N * C;
  for (int i = 0; i < N; ++i)
    for (int c = 0; c < C; ++c) {
      float ms = -INFINITY, mo = -INFINITY;
      for (int t = 0; t < T; ++t) {
        const float a = sub[((int64_t)i * T + t) * C + c], b = obj[((int64_t)i * T + t) * C + c];
        if (a > ms) ms = a;
        if (b > mo) mo = b;
      }
      st[(size_t)i * C + c] = ms;
      ot[(size_t)i * C + c] = mo;
    }
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      if (i == j) { out[(size_t)i * N + j] = 0.f; continue; }
      float acc = b2[0];
      for (int k = 0; k < Hd; ++k) {
        float h = b1[k];
        const float* wr = W1 + (size_t)k * 2 * C;
        for (int c = 0; c < C; ++c) h += wr[c] * st[(size_t)i * C + c];
        for (int c = 0; c < C; ++c) h += wr[C + c] * ot[(size_t)j * C + c];
        if (h > 0) acc += w2[k] * h;
      }
      out[(size_t)i * N + j] = acc;
    }
  free(st);
}

/* ---------------------------------------------------------------------------------------------------------------
 * COCO compressed run-length strings -- [3P] pycocotools 2.0.x `mask.encode` / `mask.decode` (cocoapi
 * common/maskApi.c: rleEncode, rleToString, rleFrString), used by models/mask2former_vps/utils.py:8,48 and
 * models/unitrack/utils/io.py:31.  pycocotools is not installed here ("parity unpinned" against the package itself);
 * this is a scalar restatement of the PUBLISHED algorithm, independent of the vectorised numpy codec the product
 * ships (openpvsg_amd/tubes.py), and tests/golden/rle_known_answers.json holds hand-derived strings for it.
 *   runs:    column-major scan of the (h, w) mask, counts alternate 0-run, 1-run, ... starting with zeros
 *            (a mask that starts with a one gets a leading 0 count)
 *   string:  count i >= 3 is stored as the difference to count i-2; each value in 5-bit groups, least significant
 *            first, bit 0x20 = "more groups follow", a value ends when the rest is 0 with bit 0x10 clear or -1 with
 *            bit 0x10 set (sign extension); every group + 48 is one ASCII character
 * oracle_rle_encode: mask row-major (h, w) uint8 -> string (returns its length, or -1 if `cap` is too small)
 * oracle_rle_decode: string -> row-major mask (returns the number of counts, -1 on overrun)
 */
static long rle_counts(const uint8_t* mask, int h, int w, uint32_t* cnts) {
  long k = 0;
  uint32_t c = 0;
  uint8_t p = 0;
  for (long j = 0; j < (long)h * w; ++j) {
    const uint8_t v = mask[(j % h) * (long)w + (j / h)] != 0;     /* element j of the column-major order */
    if (v != p) {
      cnts[k++] = c;
      c = 0;
      p = v;
    }
    ++c;
  }
  cnts[k++] = c;
  return k;
}

long oracle_rle_encode(const uint8_t* mask, int h, int w, char* s, long cap) {
  uint32_t* cnts = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)h * w + 1));
  const long m = rle_counts(mask, h, w, cnts);
  long p = 0;
  for (long i = 0; i < m; ++i) {
    long x = (long)cnts[i];
    if (i > 2) x -= (long)cnts[i - 2];
    int more = 1;
    while (more) {
      char c = (char)(x & 0x1f);
      x >>= 5;
      more = (c & 0x10) ? x != -1 : x != 0;
      if (more) c |= 0x20;
      c += 48;
      if (p + 1 >= cap) {
        free(cnts);
        return -1;
      }
      s[p++] = c;
    }
  }
  s[p] = 0;
  free(cnts);
  return p;
}

long oracle_rle_decode(const char* s, int h, int w, uint8_t* mask) {
  long m = 0, p = 0, pos = 0;
  long prev1 = 0, prev2 = 0;                                    /* counts m-1 and m-2 */
  uint8_t v = 0;
  for (long j = 0; j < (long)h * w; ++j) mask[j] = 0;
  while (s[p]) {
    long x = 0;
    int k = 0, more = 1;
    while (more) {
      const char c = (char)(s[p] - 48);
      x |= (long)(c & 0x1f) << (5 * k);
      more = c & 0x20;
      ++p;
      ++k;
      if (!more && (c & 0x10)) x |= -1L << (5 * k);
    }
    if (m > 2) x += prev2;
    if (pos + x > (long)h * w) return -1;
    if (v)
      for (long j = pos; j < pos + x; ++j) mask[(j % h) * (long)w + (j / h)] = 1;
    pos += x;
    v ^= 1;
    prev2 = prev1;
    prev1 = x;
    ++m;
  }
  return m;
}
