// The tail of a clip step on the device: class decision + kept-query compaction, first-appearance tube bookkeeping and the
// tube feature scatter -- so that the host waits ONCE per clip (for the tube count the relation head's shapes need) instead of
// after the class decision (torch.nonzero) and again inside tube assembly (ids to the host, numpy, indices back).
//
// Replaces (reference):
//   keep = labels.ne(num_classes) & (scores > object_mask_thr); cur_scores = scores[keep] ...
//                                                      models/mask2former/mask2former_fusion_head.py:117-124
//   concat_seq: tubes keyed by segment id in order of first appearance, feat[0] of an id's list per frame,
//   absent frames = None -> zeros                      models/mask2former_vps/utils.py:20-89,
//                                                      utils/relation_matching.py:431-444
// (SURVEY.md section 8f rows 1-2).  Host mirror: openpvsg_amd/pipeline.py (PVSGPipeline._tail_device); the numpy form
// (pipeline.assemble_tubes) stays as the checker of these kernels (tests/test_tubes.py) and as the path of ragged inputs.
#include "common.h"

#include "../../include/openpvsg_hip.h"

namespace pvsg {

constexpr int SEL_MAXK = PVSG_SEL_MAXK;            // 128: stride of the kept-query tables and of seg_id rows
static_assert(PVSG_SEL_WORDS == 4 + 3 * PVSG_SEL_MAXK, "selection record layout");

// one workgroup of 128 threads: thread q decides query q; kept queries are compacted in query order by ballots
__global__ __launch_bounds__(128) void panoptic_select_kernel(const float* __restrict__ scores,
                                                              const long long* __restrict__ labels, int Q,
                                                              int num_classes, float thr, int* __restrict__ sel) {
  __shared__ int s_cnt[2];
  const int q = threadIdx.x, lane = q & 63, wv = q >> 6;
  float sc = 0.f;
  int lb = num_classes;
  if (q < Q) { sc = scores[q]; lb = (int)labels[q]; }
  const bool keep = q < Q && lb != num_classes && sc > thr;
  const unsigned long long bal = __ballot(keep);
  if (lane == 0) s_cnt[wv] = __popcll(bal);
  __syncthreads();
  const int before = (wv ? s_cnt[0] : 0) + __popcll(bal & ((1ull << lane) - 1ull));
  const int total = s_cnt[0] + s_cnt[1];
  if (q == 0) {
    sel[0] = total > SEL_MAXK - 1 ? SEL_MAXK - 1 : total;      // what the fused kernels walk (their tables hold 127)
    sel[1] = total;                                            // > 127: the caller must take the un-fused path
    sel[2] = 0;
    sel[3] = 0;
  }
  if (keep && before < SEL_MAXK) {
    sel[4 + before] = q;
    sel[4 + SEL_MAXK + before] = lb;
    sel[4 + 2 * SEL_MAXK + before] = __float_as_int(sc);
  }
}

// One workgroup.  seg: per-frame rows of SEL_MAXK ids (-1 = dropped / unused), frame t at physical row
// (t / fpb) * rpb + t % fpb  (fpb = rpb = T for a local clip; an all-gathered frame shard carries one extra row per rank whose
// first word is that rank's f16x2 overflow count: fpb = T_local, rpb = T_local + 1).
//   table    (K+1)*1000 + 1 ints of scratch (any contents): first entry index of every id
//   rec      [N tubes, K, K before clamping, overflow count (sum over ranks / the local counter), 0, 0, 0, 0]
//   tube_ids (T*K) int64, first N valid: ids in order of first appearance (frame-major, then query order)
//   rowmap   (T, SEL_MAXK) int32: tube row of the FIRST query carrying its id in that frame, else -1
__global__ __launch_bounds__(1024) void tube_index_kernel(const int* __restrict__ seg, const int* __restrict__ sel, int T,
                                                          int fpb, int rpb, const unsigned* __restrict__ overflow,
                                                          int* __restrict__ table, int* __restrict__ rec,
                                                          long long* __restrict__ tube_ids, int* __restrict__ rowmap) {
  __shared__ int s_scan[1024];
  __shared__ int s_total;
  const int K = sel[0];
  const int tid = threadIdx.x;
  const int n = T * K;                                         // entries e = t * K + k, the order of first appearance
  const int tsize = (K + 1) * 1000 + 1;
  for (int i = tid; i < tsize; i += 1024) table[i] = 0x7fffffff;
  for (int i = tid; i < T * SEL_MAXK; i += 1024) rowmap[i] = -1;
  __syncthreads();
  auto id_at = [&](int e) {
    const int t = e / K, k = e - t * K;
    return seg[((long long)(t / fpb) * rpb + t % fpb) * SEL_MAXK + k];
  };
  for (int e = tid; e < n; e += 1024) {
    const int id = id_at(e);
    if (id >= 0 && id < tsize) atomicMin(table + id, e);
  }
  __syncthreads();
  // rank of every first entry = number of first entries before it: each thread owns a contiguous chunk
  const int per = (n + 1023) / 1024;
  const int e0 = tid * per, e1 = min(n, e0 + per);
  int cnt = 0;
  for (int e = e0; e < e1; ++e) {
    const int id = id_at(e);
    cnt += (id >= 0 && id < tsize && table[id] == e) ? 1 : 0;
  }
  s_scan[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {                   // inclusive Hillis-Steele scan over the 1024 chunk counts
    const int v = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  if (tid == 1023) s_total = s_scan[1023];
  int rank = s_scan[tid] - cnt;
  __syncthreads();                                             // every chunk has read its counts before ranks overwrite ids
  for (int e = e0; e < e1; ++e) {
    const int id = id_at(e);
    if (id >= 0 && id < tsize && table[id] == e) {
      tube_ids[rank] = id;
      table[id] = -(rank + 1);                                 // first-entry index -> tube row, in place (negative = row)
      ++rank;
    }
  }
  __syncthreads();
  for (int e = tid; e < n; e += 1024) {
    const int id = id_at(e);
    if (id < 0 || id >= tsize) continue;
    const int t = e / K, k = e - t * K;
    bool first = true;                                         // feat[0] of the id's list in this frame = smallest k
    for (int k2 = 0; k2 < k; ++k2) first = first && id_at(t * K + k2) != id;
    if (first) rowmap[t * SEL_MAXK + k] = -table[id] - 1;
  }
  if (tid == 0) {
    unsigned ovf = 0u;
    if (overflow) ovf = overflow[0];
    if (rpb > fpb)                                             // gathered shard: the ranks' counters ride in the extra rows
      for (int r = 0; r * fpb < T; ++r) ovf += (unsigned)seg[((long long)r * rpb + fpb) * SEL_MAXK];
    rec[0] = s_total; rec[1] = K; rec[2] = sel[1]; rec[3] = (int)ovf;
    rec[4] = 0; rec[5] = 0; rec[6] = 0; rec[7] = 0;
  }
}

// feats (N, T, C): row (n, t) = the query feature of the frame-first query of tube n in frame t, zeros where the tube is absent
__global__ __launch_bounds__(128) void tube_scatter_kernel(const float* __restrict__ query, long long qstride,
                                                           const int* __restrict__ sel, const int* __restrict__ rowmap,
                                                           float* __restrict__ feats, int T, int C) {
  __shared__ int s_k;
  const int n = blockIdx.x / T, t = blockIdx.x - n * T;
  if (threadIdx.x == 0) s_k = -1;
  __syncthreads();
  if (rowmap[t * SEL_MAXK + threadIdx.x] == n) s_k = threadIdx.x;      // at most one query per (tube, frame)
  __syncthreads();
  float* dst = feats + ((long long)n * T + t) * C;
  if (s_k < 0) {
    for (int c = threadIdx.x; c < C; c += 128) dst[c] = 0.f;
    return;
  }
  const float* src = query + (long long)sel[4 + s_k] * qstride;
  for (int c = threadIdx.x; c < C; c += 128) dst[c] = src[c];
}

}  // namespace pvsg

extern "C" int pvsg_panoptic_select(const float* scores, const long long* labels, int Q, int num_classes,
                                    float score_thr, int* sel, void* stream_) {
  using namespace pvsg;
  PVSG_REQUIRE(scores && labels && sel, "panoptic_select: null pointer argument");
  PVSG_REQUIRE(Q > 0 && Q <= SEL_MAXK, "panoptic_select: 1..%d queries (got %d)", SEL_MAXK, Q);
  hipLaunchKernelGGL(panoptic_select_kernel, dim3(1), dim3(128), 0, static_cast<hipStream_t>(stream_), scores, labels, Q,
                     num_classes, score_thr, sel);
  PVSG_LAUNCH_CHECK("panoptic_select");
  return PVSG_OK;
}

extern "C" long long pvsg_tube_index_table_words(void) { return (long long)pvsg::SEL_MAXK * 1000 + 1; }

extern "C" int pvsg_tube_index(const int* seg_id, const int* sel, int T, int frames_per_block, int rows_per_block,
                               const uint32_t* overflow, int* table_ws, int* rec, long long* tube_ids, int* rowmap,
                               void* stream_) {
  using namespace pvsg;
  PVSG_REQUIRE(seg_id && sel && table_ws && rec && tube_ids && rowmap, "tube_index: null pointer argument");
  PVSG_REQUIRE(T > 0 && frames_per_block > 0 && rows_per_block >= frames_per_block && T % frames_per_block == 0,
               "tube_index: bad frame layout (T=%d, %d frames in blocks of %d rows)", T, frames_per_block, rows_per_block);
  hipLaunchKernelGGL(tube_index_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream_), seg_id, sel, T,
                     frames_per_block, rows_per_block, overflow, table_ws, rec, tube_ids, rowmap);
  PVSG_LAUNCH_CHECK("tube_index");
  return PVSG_OK;
}

extern "C" int pvsg_tube_scatter(const float* query, long long query_row_stride, const int* sel, const int* rowmap,
                                 float* feats, int N, int T, int C, void* stream_) {
  using namespace pvsg;
  PVSG_REQUIRE(query && sel && rowmap && feats, "tube_scatter: null pointer argument");
  PVSG_REQUIRE(N >= 0 && T > 0 && C > 0 && (long long)N * T <= 0x7fffffffLL, "tube_scatter: bad shape (N=%d T=%d C=%d)", N, T, C);
  if (N == 0) return PVSG_OK;
  hipLaunchKernelGGL(tube_scatter_kernel, dim3((unsigned)(N * T)), dim3(128), 0, static_cast<hipStream_t>(stream_), query,
                     query_row_stride, sel, rowmap, feats, T, C);
  PVSG_LAUNCH_CHECK("tube_scatter");
  return PVSG_OK;
}

// ---- run boundaries of instance masks in COCO's (column-major) scan order -------------------------------------------------------
// [3P] mmdet `encode_mask_results` (pycocotools mask.encode) run-length codes every (H, W) mask of `ins_results` column by column
// (tools/test.py -> single_gpu_test).  The masks are blobs: a few hundred boundaries per mask against 0.9 MB of mask bytes, so only
// the boundaries travel.  Position p = x * H + y; a change AT (x, y) means mask(x, y) != mask(pred(x, y)), pred = the element before
// it in scan order ((x, y - 1), or (x - 1, H - 1) for y == 0); it is recorded as p - 1, the index numpy's
// `flat[1:] != flat[:-1]` would report.  A thread owns 4 columns x RLE_SEG rows (4-byte loads, coalesced along x);
//   pass 1 (pvsg_rle_count):      changes per (mask, column, row segment), laid out in scan order -> the caller's exclusive scan
//   pass 2 (pvsg_rle_positions):  the positions, written at the scanned offsets: sorted by (mask, position) by construction.
// The tensor-op form this replaces (transpose copy, shifted compare, sum, nonzero, cat) made five passes over the 92 MB of one
// 720p image x 100 masks: 1.0 ms of the 1.28 ms `rles()` took; these two read the masks twice.
namespace pvsg {
constexpr int RLE_SEG = 48;

template <bool WRITE>
__global__ __launch_bounds__(256) void rle_boundaries_kernel(const unsigned char* __restrict__ masks, int n, int H, int W, int nseg,
                                                             int* __restrict__ counts, const int* __restrict__ offsets,
                                                             int* __restrict__ positions) {
  const int W4 = W >> 2;
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)n * nseg * W4;
  if (tid >= total) return;
  const int cg = (int)(tid % W4);
  const int seg = (int)((tid / W4) % nseg), i = (int)(tid / ((long long)W4 * nseg));
  const unsigned char* m = masks + (long long)i * H * W;
  const int x0 = 4 * cg, y0 = seg * RLE_SEG, y1 = min(H, y0 + RLE_SEG);
  unsigned prev;                                                 // the 4 predecessors of row y0, one byte per column (0 / 1)
  if (y0 > 0) {
    prev = *reinterpret_cast<const unsigned*>(m + (long long)(y0 - 1) * W + x0);
  } else {
    const unsigned char* last = m + (long long)(H - 1) * W;
    const unsigned b0 = x0 > 0 ? last[x0 - 1] : m[0];            // column 0 has no predecessor: compare (0, 0) with itself
    prev = b0 | ((unsigned)last[x0] << 8) | ((unsigned)last[x0 + 1] << 16) | ((unsigned)last[x0 + 2] << 24);
  }
  int cnt[4] = {0, 0, 0, 0};
  int* dst[4];
  if (WRITE) {
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = positions + offsets[((long long)i * W + x0 + c) * nseg + seg];
  }
  for (int y = y0; y < y1; ++y) {
    const unsigned cur = *reinterpret_cast<const unsigned*>(m + (long long)y * W + x0);
    const unsigned diff = (cur ^ prev) & 0x01010101u;            // masks are 0 / 1 bytes (torch.bool)
    if (diff) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if ((diff >> (8 * c)) & 1u) {
          if (WRITE) dst[c][cnt[c]] = (x0 + c) * H + y - 1;
          ++cnt[c];
        }
    }
    prev = cur;
  }
  if (!WRITE) {
#pragma unroll
    for (int c = 0; c < 4; ++c) counts[((long long)i * W + x0 + c) * nseg + seg] = cnt[c];
  }
}
}  // namespace pvsg

extern "C" int pvsg_rle_segments(int H) { return (H + pvsg::RLE_SEG - 1) / pvsg::RLE_SEG; }

extern "C" int pvsg_rle_count(const unsigned char* masks, int n, int H, int W, int* counts, void* stream_) {
  using namespace pvsg;
  PVSG_REQUIRE(masks && counts, "rle_count: null pointer argument");
  PVSG_REQUIRE(n > 0 && H > 0 && W > 0, "rle_count: bad shape");
  if (W % 4 || (long long)H * W >= (1LL << 31) || (reinterpret_cast<uintptr_t>(masks) & 3u))
    return set_err(PVSG_ERR_UNSUPPORTED, "rle_count: built for W %% 4 == 0, H * W < 2^31, 4-byte aligned masks (got H=%d W=%d)", H, W);
  const int nseg = pvsg_rle_segments(H);
  const long long total = (long long)n * nseg * (W / 4);
  PVSG_REQUIRE((total + 255) / 256 < (1LL << 31), "rle_count: too many blocks");
  hipLaunchKernelGGL(rle_boundaries_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     masks, n, H, W, nseg, counts, (const int*)nullptr, (int*)nullptr);
  PVSG_LAUNCH_CHECK("rle_count");
  return PVSG_OK;
}

extern "C" int pvsg_rle_positions(const unsigned char* masks, int n, int H, int W, const int* offsets, int* positions, void* stream_) {
  using namespace pvsg;
  PVSG_REQUIRE(masks && offsets && positions, "rle_positions: null pointer argument");
  PVSG_REQUIRE(n > 0 && H > 0 && W > 0, "rle_positions: bad shape");
  if (W % 4 || (long long)H * W >= (1LL << 31) || (reinterpret_cast<uintptr_t>(masks) & 3u))
    return set_err(PVSG_ERR_UNSUPPORTED, "rle_positions: built for W %% 4 == 0, H * W < 2^31, 4-byte aligned masks (got H=%d W=%d)", H, W);
  const int nseg = pvsg_rle_segments(H);
  const long long total = (long long)n * nseg * (W / 4);
  hipLaunchKernelGGL(rle_boundaries_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     masks, n, H, W, nseg, (int*)nullptr, offsets, positions);
  PVSG_LAUNCH_CHECK("rle_positions");
  return PVSG_OK;
}
