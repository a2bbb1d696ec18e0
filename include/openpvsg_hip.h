/*
 * openpvsg_hip.h -- C ABI of the MI355X (gfx950) backend for the OpenPVSG inference hot path.
 *
 * Library: openpvsg_amd/lib/libopenpvsg_hip.so  (built by __graft_entry__.build()).
 *
 * Conventions (same as the op boundary the reference binds):
 *   - every pointer is a DEVICE pointer owned by the caller (the caller keeps it alive until the
 *     stream has drained); the library allocates nothing and never synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - all tensors are dense, row-major ("contiguous" in torch terms), float32 unless noted;
 *   - return value: 0 = PVSG_OK, otherwise an error code; pvsg_last_error() gives the text for the
 *     calling thread (the Python host raises RuntimeError, as TORCH_CHECK does in the reference's
 *     extension).
 *
 * Each entry point names the reference interface it replaces.  Paths are relative to the
 * reference tree (LilyDaytoy/OpenPVSG); "[3P]" marks interfaces of mmcv-full 1.4.0 / mmdet 2.25.0
 * that the reference selects by config but does not vendor.
 */
#ifndef OPENPVSG_HIP_H_
#define OPENPVSG_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVSG_OK 0
#define PVSG_ERR_INVALID_ARG 1
#define PVSG_ERR_UNSUPPORTED 2
#define PVSG_ERR_HIP 3

/* ---- library ----------------------------------------------------------------------------- */
const char* pvsg_last_error(void);
const char* pvsg_version(void);
int pvsg_abi_version(void);

/* Host-side codec of the result formats (no device work): COCO compressed run-length strings of many masks at once, what
 * [3P] pycocotools `mask.encode` / rleToString produce for models/unitrack/utils/io.py:14-36 (MOTS lines) and mmdet
 * `encode_mask_results`.  counts: run lengths of all masks back to back (zeros first), seg[j] of them belong to mask j;
 * out: >= 13 bytes per count; out_len[j] characters belong to mask j.  Returns the total number of characters (-1: bad args). */
long long pvsg_rle_counts_to_chars(const long long* counts, const long long* seg, int nseg, unsigned char* out,
                                   long long* out_len);

/* ---- a1: multi-scale deformable attention sampling core ----------------------------------
 * Replaces [3P] mmcv.ops.multi_scale_deform_attn: ext_module.ms_deform_attn_forward(value,
 * spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step)
 * selected by configs/mask2former/mask2former_r50_lsj_8x2_50e_coco-panoptic_custom_single_video_test.py:46-56
 * and reached from models/mask2former/mask2former_head.py:417.
 *   value            (B, S, M, D)
 *   spatial_shapes   (L, 2) int64, (H_l, W_l), low->high resolution
 *   level_start_index(L)    int64
 *   sampling_loc     (B, Lq, M, L, P, 2) normalised (x, y) in [0,1]
 *   attn_weight      (B, Lq, M, L, P)
 *   out              (B, Lq, M*D)
 */
int pvsg_ms_deform_attn_forward(const float* value, const int64_t* spatial_shapes,
                                const int64_t* level_start_index, const float* sampling_loc,
                                const float* attn_weight, float* out, int B, int S, int M, int D,
                                int Lq, int L, int P, int im2col_step, void* stream);

/* ---- a3: per-query mask-logit projection (fp32 matrix cores) --------------------------------
 * Replaces torch.einsum('bqc,bchw->bqhw') models/mask2former/mask2former_head.py:382 and
 * torch.einsum('bqc,btchw->btqhw') models/mask2former_vps/mask2former_video_head.py:344.
 *   mask_embed   (B, Q, C)      output of the 3-layer mask MLP
 *   mask_feature (B, T, C, N)   N = h*w of the stride-4 map (T = 1 for the image head)
 *   out          (B, T, Q, N)
 * Supported: Q <= 112, C % 16 == 0, C <= 320 (else PVSG_ERR_UNSUPPORTED); N % 4 == 0 takes the
 * vectorised path. */
int pvsg_mask_logits_forward(const float* mask_embed, const float* mask_feature, float* out, int B,
                             int T, int Q, int C, int N, void* stream);

/* Same contraction (mask2former_head.py:382, mask2former_video_head.py:344) on the bf16 matrix cores from the exact
 * three-limb split (csrc/gemm_bf16x3.hip: the mask embeddings are packed on the fly as the weight of a 1x1 convolution over
 * the (T, C, N) feature planes).  f32-class result (same error class as the library's f32 contraction).
 *   w_scratch  B * pvsg_gemm_bf16x3_packed_elems(Q, C) bf16 elements of device scratch, 16-byte aligned
 * Supported: C % 16 == 0, Q % 4 == 0, C*N and Q*N < 2^29 (else PVSG_ERR_UNSUPPORTED). */
int pvsg_mask_logits_bf16x3(const float* mask_embed, const float* mask_feature, void* w_scratch, float* out, int B, int T,
                            int Q, int C, long long N, void* stream);

/* ---- a3: attention-mask bits straight from low-resolution features --------------------------
 * Replaces F.interpolate(mask_pred, level size) -> flatten -> repeat(num_heads) -> sigmoid() < 0.5
 * (mask2former_head.py:383-393, video_head.py:346-357) and the all-masked-row test that feeds the
 * reset at mask2former_head.py:453-454, for levels whose size divides the stride-4 map by 2/4/8
 * (bilinear resize is then linear and commutes with the projection).
 *   feature_lowres (B, T, C, N_l)  = pvsg_center_downsample output for that level
 *   bits           (B, T*N_l, 4) uint32: bit q of a key's 128-bit word = 1 <=> query q is BLOCKED
 *   flags          (B, 4) uint32:        bit q = 1 <=> query q has at least one unblocked key */
int pvsg_attn_mask_bits_forward(const float* mask_embed, const float* feature_lowres, uint32_t* bits,
                                uint32_t* flags, int B, int T, int Q, int C, int N, void* stream);

/* Same bits and flag words (mask2former_head.py:383-393,453-454; video_head.py:346-357) from the split-bf16 kernel
 * (csrc/gemm_bf16x3.hip: logits on the bf16 matrix cores from the exact three-limb split, thresholded in registers).
 *   w_scratch  B * pvsg_gemm_bf16x3_packed_elems(Q, C) bf16 elements of device scratch, 16-byte aligned
 * Supported: C % 16 == 0, Q <= 128, C*N < 2^29 (else PVSG_ERR_UNSUPPORTED). */
int pvsg_attn_mask_bits_bf16x3(const float* mask_embed, const float* feature_lowres, void* w_scratch, uint32_t* bits,
                               uint32_t* flags, int B, int T, int Q, int C, long long N, void* stream);

/* Same bits/flags from already resized logits (general sizes; exact for any interpolate factor).
 *   logits_lowres (B, T, Q, HW) */
int pvsg_attn_mask_pack(const float* logits_lowres, uint32_t* bits, uint32_t* flags, int B, int T,
                        int Q, int HW, void* stream);

/* Centre-tap (== bilinear, align_corners=False) down-sampling of (planes, H, W) by 2, 4 and 8 in one
 * pass; H % 8 == 0 and W % 8 == 0.  Replaces the three F.interpolate calls per decoder layer. */
int pvsg_center_downsample(const float* feature, float* d2, float* d4, float* d8, long long planes,
                           int H, int W, void* stream);

/* ---- a4/a5: masked cross-attention (streaming, split over key ranges) ------------------------
 * Replaces [3P] mmcv MultiheadAttention -> nn.MultiheadAttention with a (B*heads, Q, K) bool mask as
 * driven by mask2former_head.py:457-468 / video_head.py:435-446.
 *   q_proj (B, Q, 256)  = ((query + query_pos) Wq + bq) / sqrt(32)
 *   k_proj (B, K, 256)  = (key + key_pos) Wk + bk ;  v_proj (B, K, 256) = value Wv + bv
 *   mask_bits / mask_flags as above, or both NULL for unmasked attention
 *   part_o (B, NS, 8, Q, 32), part_ml (B, NS, 8, Q, 2): un-normalised partial output, running max
 *   and sum per key range.  pvsg_xattn_combine merges NS ranges (also the ranges gathered from other
 *   GPUs) into out (B, Q, 256); the caller applies Wo and the residual. */
int pvsg_xattn_num_splits(int B, long long K);
int pvsg_masked_xattn_partial(const float* q_proj, const float* k_proj, const float* v_proj,
                              const uint32_t* mask_bits, const uint32_t* mask_flags, float* part_o,
                              float* part_ml, int B, int Q, long long K, int M, int D, int NS,
                              void* stream);
/* The same with `kv_row_stride` floats (>= M*D, multiple of 4) between consecutive rows of k_proj / v_proj, batch stride
 * K * kv_row_stride: the key (value) projections of the three decoder layers that attend over one pyramid level
 * (mask2former_head.py:457-468: level = layer % 3) are columns [256 j, 256 j + 256) of ONE (K, 768) GEMM output. */
int pvsg_masked_xattn_partial_strided(const float* q_proj, const float* k_proj, const float* v_proj,
                                      const uint32_t* mask_bits, const uint32_t* mask_flags, float* part_o,
                                      float* part_ml, int B, int Q, long long K, int M, int D, int NS,
                                      long long kv_row_stride, void* stream);
int pvsg_xattn_combine(const float* part_o, const float* part_ml, float* out, int B, int Q, int M,
                       int D, int NS, void* stream);
/* Frame-sharded clip (BASELINE config 4): ONE message per decoder layer and rank.
 * pvsg_xattn_merge_local: a rank's NS key-range partials -> one packed record per batch element,
 *   packed (B, REC) floats, REC = M*Q*(D+2) + 4 = [o (M,Q,D) un-normalised | (m, l) (M,Q,2) | the rank's 128 flag bits];
 *   mask_flags (B,4) = "query has an unblocked key among this rank's keys" (NULL: no mask).
 * pvsg_xattn_combine_packed: packed (R, B, REC) = the all-gathered records of R ranks -> out (B, Q, M*D).  A rank whose
 *   keys are all blocked for a query attended unmasked (the reset of mask2former_head.py:453-454 is decided by the whole
 *   clip); its contribution counts only if every rank reported the query blocked -- no flag exchange before the attention. */
int pvsg_xattn_merge_local(const float* part_o, const float* part_ml, const uint32_t* mask_flags, float* packed,
                           int B, int Q, int M, int D, int NS, void* stream);
int pvsg_xattn_combine_packed(const float* packed, float* out, int R, int B, int Q, int M, int D, void* stream);

/* ---- a4/a5 + query side of a3: the Q-row part of a decoder layer in two launches -----------------
 * Replaces, per layer of [3P] mmdet DetrTransformerDecoderLayer (operation_order cross_attn, norm, self_attn,
 * norm, ffn, norm) as driven by models/mask2former/mask2former_head.py:457-470 and
 * models/mask2former_vps/mask2former_video_head.py:435-452: the cross-attention out_proj + identity + LayerNorm,
 * the self-attention over the queries ([3P] mmcv MultiheadAttention -> nn.MultiheadAttention), the FFN + LayerNorms,
 * and the query side of forward_head (mask2former_head.py:375-381: post_norm, cls_embed, mask_embed), plus the
 * next layer's cross-attention query projection -- about 35 library launches on 100 rows per layer.
 * Weights are given in MFMA-fragment order: packed = pvsg_pack_rows_weight(W (N, K) row-major), size
 * roundup16(N) * K floats, prepared once per checkpoint.  Built for 256 dims, 8 heads, Q <= 128,
 * FFN width a multiple of 512, <= 128 class outputs (PVSG_ERR_UNSUPPORTED otherwise). */
typedef struct pvsg_decoder_layer {
  const float *xo_w, *xo_b;         /* attentions.0.attn.out_proj  packed (256,256), bias */
  const float *n0_g, *n0_b;         /* norms.0 */
  const float *sa_in_w, *sa_in_b;   /* attentions.1.attn.in_proj   packed (768,256), bias (768) */
  const float *sa_out_w, *sa_out_b; /* attentions.1.attn.out_proj */
  const float *n1_g, *n1_b;         /* norms.1 */
  const float *f1_w, *f1_b;         /* ffns.0.layers.0.0           packed (F,256), bias (F) */
  const float *f2_w, *f2_b;         /* ffns.0.layers.1             packed (256,F), bias (256) */
  const float *n2_g, *n2_b;         /* norms.2 */
  int embed_dims, num_heads, ffn_dim;
} pvsg_decoder_layer;
typedef struct pvsg_decoder_head {
  const float *pn_g, *pn_b;         /* transformer_decoder.post_norm */
  const float *cls_w, *cls_b;       /* cls_embed                   packed (num_cls_out,256), bias */
  const float *m0_w, *m0_b, *m1_w, *m1_b, *m2_w, *m2_b;   /* mask_embed.0 / .2 / .4 */
  int num_cls_out;
} pvsg_decoder_head;
int pvsg_pack_rows_weight(const float* W, float* packed, int N, int K, void* stream);
/* x1 = LN0(attn_core Wo^T + bo + query);  qkv (B*Q, 768) = [((x1+pos) Wq^T + bq)/sqrt(32) | (x1+pos) Wk^T + bk | x1 Wv^T + bv]
 *   attn_core (B, Q, 256) = pvsg_xattn_combine output; query (B, Q, 256) layer input; query_pos (Q, 256) */
int pvsg_decoder_rows_pre(const pvsg_decoder_layer* layer, const float* attn_core, const float* query,
                          const float* query_pos, float* x1, float* qkv, int B, int Q, void* stream);
/* query_out = LN2(FFN(x2) + x2), x2 = LN1(selfattn(qkv) Wo^T + bo + x1);  cls_out (B*Q, num_cls_out) and
 * mask_embed_out (B*Q, 256) from post_norm(query_out);  next_q_out = ((query_out + pos) Wq'^T + bq')/sqrt(32) for the
 * next layer's cross-attention (next_q_w packed (256,256); NULL = last layer).
 * layer == NULL: head part only on x1 = the initial queries (the forward_head call before layer 0).
 * workspace: NULL, or pvsg_decoder_rows_post_workspace_bytes(B, Q) bytes ZEROED ONCE by the caller and reused from call to
 *   call (one stream at a time): with it, and at most 32 row tiles (B * ceil(Q/16)), the FFN of a tile is split over eight
 *   workgroups that meet through it (a clip's 100 query rows would otherwise occupy 7 CUs); every launch leaves its
 *   arrival counters at zero.  The byte count is 0 when the split does not apply.
 * emb_pack_f16x2: NULL, or B * pvsg_gemm_f16x2_packed_elems(Q, 256) 16-bit elements, ZEROED ONCE by the caller: the kernel also
 *   writes the mask embeddings as the packed row operand of pvsg_attn_mask_bits_packed_f16x2 (one exact power-of-two scale per
 *   query row: the bits are signs), which saves the amax / pack / zero launches of pvsg_attn_mask_bits_f16x2 in every layer.
 * flags_zero: NULL, or (B, 4) words the kernel sets to 0 -- the flag words the bits kernel ORs into. */
long long pvsg_decoder_rows_post_workspace_bytes(int B, int Q);
int pvsg_decoder_rows_post(const pvsg_decoder_layer* layer, const pvsg_decoder_head* head, const float* next_q_w,
                           const float* next_q_b, const float* x1, const float* qkv, const float* query_pos,
                           float* query_out, float* cls_out, float* mask_embed_out, float* next_q_out, void* workspace,
                           void* emb_pack_f16x2, uint32_t* flags_zero, int B, int Q, void* stream);

/* The same two launches with every row GEMM on the 16-bit matrix pipe (f16x2 split arithmetic, the form of pvsg_gemm_f16x2:
 * f32 operands split into two f16 limbs per lane, three limb products, f32 accumulation -- an f32-class result at 1/5 of the
 * f32-MFMA time; a decoder layer chains ~11 of these 256 x 256 GEMMs on 16 rows whatever the clip length).  Every weight pointer
 * of pvsg_decoder_layer / pvsg_decoder_head / next_q_w is then packed by pvsg_pack_rows_weight_f16x2 (W (N, K) row-major, K % 32
 * == 0) into pvsg_rows_f16x2_packed_floats(N, K) floats, 16-byte aligned: [max|w|, 2^-e, 0, 0] + per (16-column tile, 32-wide k
 * block) the w_h and w_l fragments of w 2^e.  overflow: NULL or the caller's device counter, incremented per lane that split an
 * activation with |a| > 65504 (results invalid: re-run on the f32 entries above -- openpvsg_amd/ops.py does). */
long long pvsg_rows_f16x2_packed_floats(int N, int K);
int pvsg_pack_rows_weight_f16x2(const float* W, float* packed, int N, int K, void* stream);
int pvsg_decoder_rows_pre_f16x2(const pvsg_decoder_layer* layer, const float* attn_core, const float* query,
                                const float* query_pos, float* x1, float* qkv, int B, int Q, uint32_t* overflow, void* stream);
int pvsg_decoder_rows_post_f16x2(const pvsg_decoder_layer* layer, const pvsg_decoder_head* head, const float* next_q_w,
                                 const float* next_q_b, const float* x1, const float* qkv, const float* query_pos,
                                 float* query_out, float* cls_out, float* mask_embed_out, float* next_q_out, void* workspace,
                                 void* emb_pack_f16x2, uint32_t* flags_zero, int B, int Q, uint32_t* overflow, void* stream);

/* ---- a10 + a13: the relation head's encoders and temporal models as fused row kernels (csrc/relation_rows.hip) ----------
 * Replace the nn.TransformerEncoder / nn.Linear / F.conv1d library calls of
 *   models/relation_head/base.py:26-40        ObjectEncoder (2 x TransformerEncoderLayer(256, nhead 8, ff 512), attention across
 *                                              the N objects of a frame: feats (N, T, 256), batch_first=False)
 *   models/relation_head/transformer.py:7-56  TemporalTransformer (+ pe, TransformerEncoderLayer(512, nhead 4, ff 512) over the T
 *                                              frames of a pair, LayerNorm, fc1 / fc2 / span_head / pred_head, max over frames)
 *   models/relation_head/convolution.py:6-75  HandcraftedFilter, Learnable1DConv;  base.py:6-23 VanillaModel
 * torch.nn.TransformerEncoderLayer semantics: post-norm, ReLU, eval mode (dropout off).  Weights in MFMA-fragment order
 * (pvsg_pack_rows_weight); built for (d_model 256, 8 heads) and (d_model 512, 4 heads), dim_feedforward 512, any sequence
 * length (PVSG_ERR_UNSUPPORTED otherwise).  Rows of a (S sequences x L positions) token set live at
 * row = s * seq_stride + pos * pos_stride of (rows, d_model) tensors: ObjectEncoder on feats (N, T, 256): S = T, L = N,
 * seq_stride 1, pos_stride T;  TemporalTransformer on (P, T, 512): S = P, L = T, seq_stride T, pos_stride 1. */
typedef struct pvsg_encoder_layer {
  const float *in_w, *in_b;         /* self_attn.in_proj_weight packed (3D, D), in_proj_bias (3D) */
  const float *out_w, *out_b;       /* self_attn.out_proj       packed (D, D), bias */
  const float *n1_g, *n1_b;         /* norm1 */
  const float *f1_w, *f1_b;         /* linear1                  packed (F, D), bias (F) */
  const float *f2_w, *f2_b;         /* linear2                  packed (D, F), bias (D) */
  const float *n2_g, *n2_b;         /* norm2 */
  int d_model, num_heads, ffn_dim;
  float eps1, eps2;
} pvsg_encoder_layer;
typedef struct pvsg_relation_tail {
  const float *ln_g, *ln_b;         /* NULL, or TemporalTransformer.layer_norm applied to the input rows */
  const float *fc1_w, *fc1_b;       /* fc1 packed (256, 512), bias */
  const float *fc2_w, *fc2_b;       /* fc2 packed (128, 256), bias */
  const float *head_w, *head_b;     /* packed (128, 128): rows 0..R-1 span_head, rows 64..64+R-1 pred_head, the rest zero; bias (128) */
  const float *filter;              /* NULL, or the 5 taps of HandcraftedFilter (F.conv1d along T, padding 2, per channel) */
  int dim, num_relations;           /* 512, R <= 64 */
  float eps;
} pvsg_relation_tail;
/* in_proj of the FIRST layer of E (1 or 2) encoders that share their input: qkv (E, rows, 3D), q columns scaled by 1/sqrt(D/heads).
 *   x (rows, D), or NULL with the gather form: row (p, t) = [gather_sub[pairs[p][0], t, :] | gather_obj[pairs[p][1], t, :]]
 *   (concatenate_sub_obj, train_utils.py:67-81; sources (N, L, D/2), pairs (rows / L, 2) int64);
 *   pe: NULL or (>= L, D), added to the row at position `row % L` (PositionalEncoding, transformer.py:59-81);
 *   x0_out: NULL or (rows, D): the rows after gather / + pe = the first layer's residual input. */
int pvsg_rel_qkv(const pvsg_encoder_layer* layers, int E, const float* x, const float* gather_sub, const float* gather_obj,
                 const long long* gather_pairs, const float* pe, float* x0_out, float* qkv, long long rows, int L, void* stream);
/* one encoder layer of E encoders: y (E, S*L rows, D) = norm2(x1 + FFN(x1)), x1 = norm1(x + out_proj(softmax(q k^T) v));
 *   x: residual input, encoder e at x + e * x_encoder_stride floats (0: both encoders read the same rows);
 *   qkv (E, rows, 3D) from pvsg_rel_qkv / the previous layer; qkv_next (E, rows, 3D) != qkv: in_proj of next_layers on y, or both NULL. */
int pvsg_rel_encoder_layer(const pvsg_encoder_layer* layers, const pvsg_encoder_layer* next_layers, int E, const float* x,
                           long long x_encoder_stride, const float* qkv, float* y, float* qkv_next, int S, int L,
                           long long seq_stride, long long pos_stride, void* stream);
/* the attention of such a layer alone: out (rows, D) = concat_heads(softmax(q k^T / sqrt(D / heads)) v) from qkv (rows, 3 D) with
 * UN-scaled q -- for long videos, where the layer's linear parts run as token GEMMs (pvsg_gemm_bf16x3 / pvsg_add_layernorm) and
 * only the attention stays a row kernel.  Same (S, L, seq_stride, pos_stride) row addressing. */
int pvsg_rel_attention(const float* qkv, float* out, int S, int L, long long seq_stride, long long pos_stride, int d_model,
                       int num_heads, void* stream);
/* Learnable1DConv layer: y (P, T, C) = relu(Conv1d(C, C, 5, padding 2)(x along T));  w_packed = 5 x pvsg_pack_rows_weight(W[:, :, k]) */
int pvsg_rel_conv5(const float* w_packed, const float* bias, const float* x, float* y, int P, int T, int C, void* stream);
/* tail of every relation model on x (P, T, 512): span_pred (P, T, R), relation_pred (P, R) = max over T of pred_head.
 *   workspace: NULL (one workgroup per pair walks all its frames), or pvsg_rel_tail_workspace_bytes(P, T) bytes of scratch (no
 *   initialisation): videos of more than 64 frames are then cut into chunks of frames, one workgroup per (pair, chunk), and a
 *   second small launch folds the chunks' maxima.  The byte count is 0 where one workgroup per pair is used anyway. */
long long pvsg_rel_tail_workspace_bytes(int P, int T);
int pvsg_rel_tail(const pvsg_relation_tail* tail, const float* x, float* span_pred, float* relation_pred, void* workspace,
                  int P, int T, void* stream);

/* ---- a11: pairwise relation proposal scorer -------------------------------------------------
 * Replaces models/relation_head/base.py:49-62 PairProposalNetwork.forward (N^2 Python loop).
 *   sub_feats, obj_feats (N, T, 256)  encoder outputs; tokens = max over T (base.py:50-51)
 *   W1 (1024, 512), b1 (1024), w2 (1024), b2 (1)   = pair_ffn.0 / pair_ffn.2 parameters
 *   W1T (2, 256, 1024) = pvsg_pair_prepare_weights(W1), computed once per checkpoint
 *   work_uv (2, N, 1024) scratch; tokens_out (2, N, 256) or NULL; pair_matrix (N, N), diagonal 0 */
int pvsg_pair_prepare_weights(const float* W1, float* W1T, int C, int Hd, void* stream);
int pvsg_pair_score_forward(const float* sub_feats, const float* obj_feats, const float* W1T,
                            const float* b1, const float* w2, const float* b2, float* work_uv,
                            float* tokens_out, float* pair_matrix, int N, int T, int C, int Hd,
                            void* stream);

/* a12: pick_top_pairs_eval (models/relation_head/test_utils.py:4-22): the P = min(N^2 - N, num_total_pairs) best
 * off-diagonal entries of pair_matrix (N, N), best first, as pairs (P, 2) int64 [subject, object]; ties to the lower flat
 * index.  One launch instead of clone + fill_diagonal_ + topk + div + remainder + stack.  P <= 1024, N <= 128
 * (the keys of a video live in the registers of one workgroup; PVSG_ERR_UNSUPPORTED beyond). */
int pvsg_top_pairs(const float* pair_matrix, long long* pairs, int N, int P, void* stream);

/* ---- a7 + a8: fused x4 up-sampling + panoptic fusion ---------------
 * Replaces F.interpolate(mask_pred, batch_input_shape) models/mask2former/mask2former_head.py:675-679
 * (video_head.py:660-667), the crop at mask2former_fusion_head.py:372-374 and
 * panoptic_postprocess_with_query mask2former_fusion_head.py:96-171, for T frames that share one
 * kept-query set (clip mode) or T = 1.
 *   mask_logits (T, Q, h, w)   last layer's stride-4 logits
 *   kept_idx / kept_score / kept_class (K)   queries with label != background and score > thr, in
 *                                            query order; score = softmax max (fusion_head.py:117-120)
 *   panoptic (T, ih, iw) int32; seg_id (T, K) int32 (-1 = dropped, else class + 1000*instance)
 *   panoptic (T, oh, ow) int32; owner_ws (T*oh*ow) bytes, counter_ws (T*3*128) int32: scratch
 *   (H, W) = batch_input_shape, (ih, iw) = img_shape crop, (oh, ow) = ori_shape: the second resize under
 *   rescale=True (mask2former_fusion_head.py:376-383) is composed with the first one per output pixel;
 *   pass (oh, ow) = (ih, iw) when rescale is off */
int pvsg_panoptic_fuse(const float* mask_logits, const int* kept_idx, const float* kept_score,
                       const int* kept_class, int* panoptic, int* seg_id, unsigned char* owner_ws,
                       int* counter_ws, int T, int Q, int K, int h, int w, int H, int W, int ih, int iw,
                       int oh, int ow, int num_things, int num_classes, double iou_thr,
                       int filter_low_score, void* stream);

/* ---- a8 + f2 on the device: class decision, fusion and tube bookkeeping without host round trips ----------
 * pvsg_panoptic_select: the keep decision of mask2former_fusion_head.py:117-124 (`labels.ne(num_classes) & (scores >
 * object_mask_thr)`) and the compaction `scores[keep] / labels[keep] / mask_pred[keep]` need, written as a device record
 *   sel[0] = kept count clamped to 127, sel[1] = kept count, sel[2..3] = 0,
 *   sel[4 + k] = query index, sel[4 + 128 + k] = class, sel[4 + 256 + k] = score bits   of kept query k (query order)
 * scores (Q) f32 / labels (Q) int64 = F.softmax(mask_cls, -1).max(-1)  (the caller's softmax: bit-identical scores).
 * pvsg_panoptic_fuse_sel = pvsg_panoptic_fuse with K and the kept tables read from `sel`; seg_id (T, 128) int32, unused = -1.
 * If sel[1] > 127 the fused kernels cannot hold the kept set: the caller (who reads sel[1] with the tube record) must take
 * the un-fused path for that input. */
#define PVSG_SEL_MAXK 128
#define PVSG_SEL_WORDS (4 + 3 * PVSG_SEL_MAXK)
int pvsg_panoptic_select(const float* scores, const long long* labels, int Q, int num_classes, float score_thr, int* sel,
                         void* stream);
int pvsg_panoptic_fuse_sel(const float* mask_logits, const int* sel, int* panoptic, int* seg_id, unsigned char* owner_ws,
                           int* counter_ws, int T, int Q, int h, int w, int H, int W, int ih, int iw, int oh, int ow,
                           int num_things, int num_classes, double iou_thr, int filter_low_score, void* stream);

/* Tube bookkeeping of models/mask2former_vps/utils.py:20-89 (`concat_seq`: tubes keyed by segment id in order of first
 * appearance; per frame the FIRST query carrying an id supplies the feature; absent frames are zeros,
 * utils/relation_matching.py:431-444) on the per-frame segment ids pvsg_panoptic_fuse_sel wrote.
 *   seg_id   frames of 128 ids; frame t lives at row (t / frames_per_block) * rows_per_block + t % frames_per_block
 *            (a local clip: both = T; an all-gathered frame shard with one trailing row per rank whose first word is that
 *            rank's f16x2 overflow count: T_local and T_local + 1)
 *   overflow the local f16x2 overflow counter (or NULL); table_ws: pvsg_tube_index_table_words() ints of scratch
 *   rec      8 ints: [N tubes, K, K before clamping, overflow count, 0...]  -- the ONE record the host reads per clip
 *   tube_ids (T * 128) int64, first N valid; rowmap (T, 128) int32: tube row of a frame-first query, else -1
 * pvsg_tube_scatter: feats (N, T, C) from the query rows (row stride in floats) of the frame-first queries; zeros elsewhere.
 * Segment ids must be < 128 000 (class < 1000 = [3P] INSTANCE_OFFSET, at most 127 instances). */
/* Run boundaries of (n, H, W) 0 / 1 byte masks in COCO's column-major scan order ([3P] pycocotools mask.encode behind mmdet
 * `encode_mask_results`, applied to every `ins_results` mask by tools/test.py's single_gpu_test; models/unitrack/utils/io.py:14-36):
 * position p = x * H + y; boundary value = the index of the LAST element of a run (numpy: flat[1:] != flat[:-1]).
 *   pvsg_rle_count      counts (n, W, pvsg_rle_segments(H)) int32: boundaries per (mask, column, row segment), in scan order
 *   pvsg_rle_positions  positions: written at `offsets` = the EXCLUSIVE prefix sum of `counts` (caller: any scan), i.e. sorted by
 *                       (mask, position); sum(counts) ints.  W % 4 == 0, H * W < 2^31. */
int pvsg_rle_segments(int H);
int pvsg_rle_count(const unsigned char* masks, int n, int H, int W, int* counts, void* stream);
int pvsg_rle_positions(const unsigned char* masks, int n, int H, int W, const int* offsets, int* positions, void* stream);
long long pvsg_tube_index_table_words(void);
int pvsg_tube_index(const int* seg_id, const int* sel, int T, int frames_per_block, int rows_per_block,
                    const uint32_t* overflow, int* table_ws, int* rec, long long* tube_ids, int* rowmap, void* stream);
int pvsg_tube_scatter(const float* query, long long query_row_stride, const int* sel, const int* rowmap, float* feats,
                      int N, int T, int C, void* stream);

/* ---- a8 (instance branch): instance_postprocess without the (Q,H,W) float tensor ----------------
 * Replaces the per-query mask work of MaskFormerFusionHeadCustom.instance_postprocess,
 * models/mask2former/mask2former_fusion_head.py:192-242 (`mask_pred[query_indices]`, `> 0`, sigmoid-weighted
 * mask score, mask2bbox) on the same composed up-sample / crop / resize as pvsg_panoptic_fuse.
 *   mask_logits (T, Q, h, w); sel_idx (n) query index of each selected (query, class) entry, shared by the
 *   frames, or (T, n) with sel_per_frame = 1 (per-frame top-k re-selection, mask2former_vps/mask2former.py:196-200)
 *   masks (T, n, oh, ow) bytes 0/1 = resized logit > 0, or NULL (statistics only)
 *   stat_sum (T, n) float64 = sum over the mask of sigmoid(logit)
 *   stat_box (T, n, 5) int32 = {pixel count, min x, min y, max x, max y} (count 0: empty mask) */
int pvsg_instance_masks(const float* mask_logits, const int* sel_idx, unsigned char* masks, double* stat_sum,
                        int* stat_box, int T, int Q, int n, int sel_per_frame, int h, int w, int H, int W,
                        int ih, int iw, int oh, int ow, void* stream);

/* ---- a1/a2 fused forms used inside the pixel decoder ------------------------------------------
 * pvsg_msda_fused_forward: [3P] MultiScaleDeformableAttention.forward steps 4-6 (softmax over the
 * L*P logits, loc = ref + off/(W_l,H_l), sampling) from the raw projection outputs:
 *   value (B, S, 256) with row stride value_row_stride floats; oa (B, Lq, 288) raw [offsets(192) |
 *   logits(96)] with row stride oa_row_stride; pos_oa (Lq, 288) additive term or NULL;
 *   ref_points (Lq, 2) normalised (x, y), shared by the levels (valid_ratios = 1);  M=8 D=32 L=3 P=4.
 * pvsg_add_layernorm: out = LayerNorm_256(a + b + bias) (b, bias optional): the residual add +
 *   norm pair of BaseTransformerLayer in one pass. */
int pvsg_msda_fused_forward(const float* value, long long value_row_stride, const float* oa,
                            long long oa_row_stride, const float* pos_oa, const float* ref_points,
                            const int64_t* spatial_shapes, const int64_t* level_start_index, float* out,
                            int B, int S, int M, int D, int Lq, int L, int P, void* stream);
/* pvsg_msda_proj_ln_forward: the same sampling + [3P] MultiScaleDeformableAttention.forward steps 7-8 (output_proj,
 * + identity) + the BaseTransformerLayer norm that follows, in one launch:
 *   out = LayerNorm_256(identity + MSDA(...) Wo^T + bo);  wo_packed = pvsg_pack_rows_weight(output_proj.weight (256,256)),
 *   identity / out (B*Lq, 256).  The 256x256 projection runs on the matrix cores under the texture-bound gather. */
int pvsg_msda_proj_ln_forward(const float* value, long long value_row_stride, const float* oa,
                              long long oa_row_stride, const float* pos_oa, const float* ref_points,
                              const int64_t* spatial_shapes, const int64_t* level_start_index,
                              const float* wo_packed, const float* wo_bias, const float* identity,
                              const float* gamma, const float* beta, float* out, int B, int S, int M, int D, int Lq,
                              int L, int P, float eps, void* stream);
int pvsg_add_layernorm(const float* a, const float* b, const float* bias, const float* gamma,
                       const float* beta, float* out, long long rows, int C, float eps, void* stream);

/* ---- backbone glue: frozen BatchNorm (+ residual) (+ ReLU) in one in-place pass ------------------
 * [3P] mmdet ResNet (norm_eval=True): y = relu(x * scale[c] + shift[c] (+ residual)) over (planes = N*C, HW). */
int pvsg_affine_act_nchw(float* x, const float* scale, const float* shift, const float* residual, float* out,
                         long long planes, int C, long long HW, int relu, void* stream);   /* out NULL = in place */

/* ---- a9: MinVIS frame-to-frame query matching, whole video in one launch ------
 * Replaces match_from_embds (models/mask2former_vps/mask2former_min_vis.py:244-258: cosine cost, C.cpu(),
 * scipy linear_sum_assignment per frame) + the chaining loop models/mask2former_vps/mask2former.py:146-158.
 *   embds (V, T, Q, C) per-frame query embeddings of V videos;  perm (V, T, Q) int32:
 *   perm[v,0] = identity, perm[v,t][j] = query of frame t placed on slot j (chained through t-1).
 *   workspace: pvsg_minvis_chain_workspace_bytes(V, T, Q) bytes of device scratch (the T-1 cosine tables of every video,
 *   computed by the whole GPU ahead of the sequential assignment chain). */
long long pvsg_minvis_chain_workspace_bytes(int V, int T, int Q);
int pvsg_minvis_chain(const float* embds, int* perm, float* workspace, int V, int T, int Q, int C, void* stream);

/* ---- 8f row 4: IPS tube association (UniTrack flavour) -- per-object appearance embeddings ----------
 * Replaces MaskAssociationTracker.extract_emb, models/unitrack/mask.py:21-47 (mask * feature map,
 * F.interpolate(bilinear, scale_factor = sqrt(max_mask_area/area)) of the whole product, nearest resize of
 * the mask, boolean gather): only the kept cells are evaluated.
 *   feat_hwd (h,w,d) NHWC appearance features;  pan_low (h,w) int32 object id per feature cell;
 *   entries (k,3) int32 = (object slot, out row, out col), row-major per object;  obj_id (n) int32;
 *   obj_inv_scale (n) = float(1/scale_factor) (1 = not rescaled);
 *   out (k,d) raw embedding rows;  out_normalised (k,d) or NULL = rows / max(||row||, 1e-12). */
int pvsg_mask_embed_forward(const float* feat_hwd, const int* pan_low, const int* entries, const int* obj_id,
                            const float* obj_inv_scale, float* out, float* out_normalised, int h, int w, int d,
                            int k, int n_obj, void* stream);

/* ---- 8f row 4: IPS tube association -- reconstruction distance of the first association ----------
 * Replaces the tail of reconsdot_distance, models/unitrack/core/association/matching.py:194-225 (two soft-maxes over
 * the cell affinities, the einsum reconstructions, their normalisation and the two dot products), without forming
 * the soft-max matrices or the reconstructions (csrc/reconsdot.hip).  Ptp / Pdp = Pt / Pd rounded up to 32.
 *   A  (Nt*Ptp, Nd*Pdp) = F_trk F_det^T, the L2-normalised cell features of the tracks (rows, object-major) and of the
 *      observations (columns), each object zero-padded to Pt (Pd) cells as get_track_feat :174-191 pads them, then to Ptp
 *      (Pdp); cells p >= Pt / s >= Pd are ignored, the zero cells below that take part in the soft-max as in the reference;
 *   Gt (Nt, Ptp, Ptp) = F_trk[t] F_trk[t]^T,  Gd (Nd, Pdp, Pdp) likewise (zero in the padding);  tmp = 100 there;
 *   needed (Nt, Nd) bytes or NULL: pairs with 0 are not evaluated and get cost +inf -- the caller's class gate
 *      (class_aware_distance, models/unitrack/core/multitracker.py:27-34, sets exactly those to inf afterwards); the soft-max
 *      statistics still run over every cell, as in the reference;
 *   workspace: pvsg_reconsdot_workspace_bytes(Nt, Pt, Nd, Pd) bytes;  cost (Nt, Nd). */
long long pvsg_reconsdot_workspace_bytes(int Nt, int Pt, int Nd, int Pd);
int pvsg_reconsdot_cost(const float* A, const float* Gt, const float* Gd, int Nt, int Pt, int Nd, int Pd, float tmp,
                        const unsigned char* needed, float* workspace, float* cost, void* stream);

/* ---- pixel-decoder / backbone glue around the library convolutions (HBM streaming) -----------------
 * [3P] mmdet MSDeformAttnPixelDecoder.forward FPN step:
 *   out = lateral * scale[plane] + shift[plane] + bilinear_x2(top)      (GroupNorm as per-(image,channel) affine,
 *   F.interpolate(size = 2x, mode='bilinear', align_corners=False));  lateral/out (planes, 2h, 2w), top (planes, h, w),
 *   scale/shift (planes) or both NULL. */
int pvsg_fpn_merge_up2x(const float* lateral, const float* scale, const float* shift, const float* top, float* out,
                        long long planes, int h, int w, void* stream);
/* [3P] mmdet ResNet stem: MaxPool2d(3, stride 2, pad 1)(ReLU(BatchNorm_eval(x))), BN as scale[c]/shift[c];
 *   x (planes = N*C, H, W) -> out (planes, (H-1)/2+1, (W-1)/2+1). */
int pvsg_stem_bn_relu_pool(const float* x, const float* scale, const float* shift, float* out, long long planes,
                           int C, int H, int W, void* stream);
/* pixel-decoder hand-off `feat.flatten(2).transpose(1, 2)` written into a slice of the (B, S, C) token tensor,
 * with the preceding GroupNorm applied as per-(image,channel) affine (or NULL/NULL):
 *   dst[b*dst_batch_stride + p*C + c] = src[b, c, p] * scale[b*C+c] + shift[b*C+c]. */
int pvsg_nchw_to_tokens(const float* src, const float* scale, const float* shift, float* dst, int B, int C, int HW,
                        long long dst_batch_stride, void* stream);
/* the inverse (`memory[:, start:start+hw].transpose(1, 2).reshape(B, C, h, w)` made contiguous for the FPN branch):
 *   dst[b, c, p] = src[b*src_batch_stride + p*C + c]. */
int pvsg_tokens_to_nchw(const float* src, float* dst, int B, int C, int HW, long long src_batch_stride, void* stream);

/* decoder key / value inputs of one level (mask2former_head.py:421-436, mask2former_video_head.py:392-410): from the
 * encoder's token tensor (frames, S, 256) whose rows start .. start+hw of every frame are level l,
 *   v_out[f*hw + p] = tokens[f, start + p] + level_embed;   k_out = v_out + pos_enc[(f*hw + p) % pe_rows]
 * `tokens` points at row `start` of frame 0; frame_stride = S*256 floats; pos_enc (pe_rows, 256) with pe_rows = frames*hw
 * (3-D encoding of a clip) or hw (2-D encoding shared by the frames of a batch). */
int pvsg_decoder_kv_inputs(const float* tokens, const float* level_embed, const float* pos_enc, float* v_out, float* k_out,
                           long long frames, int hw, int C, long long frame_stride, long long pe_rows, void* stream);

/* [3P] mmdet ResNet bottleneck tail (norm_eval): out = relu(BN(conv1x1(x)) + identity) in one pass, for the layers
 * whose 1x1 GEMM is HBM-bound (Cin <= 256):  out[b] = act((W (Cout x Cin) @ x[b] (Cin x HW)) * scale[c] + shift[c]
 * (+ residual[b])).  Requires Cout % 32 == 0, Cin % 16 == 0, Cin <= 256, HW % 4 == 0; residual NULL = none. */
int pvsg_conv1x1_affine(const float* weight, const float* x, const float* scale, const float* shift,
                        const float* residual, float* out, int B, int Cout, int Cin, long long HW, int relu,
                        void* stream);

/* [3P] 3x3 / stride 1 / pad 1 convolution (NCHW f32) as Winograd F(2x2,3x3) on the f32 matrix cores, replacing the
 * library call behind mmdet ResNet Bottleneck.conv2 -> bn2 -> relu and the pixel decoder's FPN output convolutions
 * (mmdet MSDeformAttnPixelDecoder.output_convs[i].conv):
 *   y[n, co] = act( conv3x3(x[n], w[co]) * scale[co] + shift[co] )      scale == shift == NULL: no affine
 * `u_packed` = 16 * Cin * Cout floats written by pvsg_conv3x3_winograd_pack from the (Cout, Cin, 3, 3) weight (once per
 * weight; U = G g G^T in f64, stored in the kernel's operand order).  Requires Cin % 8 == 0, Cout % 64 == 0, even W,
 * Cin*H*W < 2^29, Cin*Cout < 2^25; other shapes return PVSG_ERR_UNSUPPORTED (the caller keeps its library convolution for those). */
int pvsg_conv3x3_winograd_pack(const float* weight, float* u_packed, int Cin, int Cout, void* stream);
int pvsg_conv3x3_winograd(const float* x, const float* u_packed, const float* scale, const float* shift, float* y,
                          int N, int Cin, int Cout, int H, int W, int relu, void* stream);

/* [3P] 3x3 / stride 2 / pad 1 convolution (NCHW f32) as a direct convolution on the f32 matrix cores with the frozen-BN
 * affine (+ ReLU) in the epilogue, replacing the library call behind the stride-2 Bottleneck.conv2 -> bn2 -> relu of mmdet
 * ResNet layers 2-4 (style='pytorch'):  y[n, co] = act( conv3x3_s2(x[n], w[co]) * scale[co] + shift[co] ),
 * y (N, Cout, (H-1)/2+1, (W-1)/2+1).  `w_packed` = 24 * Cin * Cout floats written by pvsg_conv3x3s2_pack from the
 * (Cout, Cin, 3, 3) weight (once per weight).  Requires Cin % 8 == 0, Cout % 128 == 0, Cin*H*W < 2^29, Cin*Cout < 2^25;
 * other shapes return PVSG_ERR_UNSUPPORTED (the caller keeps its library convolution for those). */
int pvsg_conv3x3s2_pack(const float* weight, float* w_packed, int Cin, int Cout, void* stream);
int pvsg_conv3x3s2_affine(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                          int N, int Cin, int Cout, int H, int W, int relu, void* stream);

/* Token-major linear layer out[M, N] = act(a[M, K] . w[N, K]^T + bias) (f32 in / out) on the bf16 matrix cores from an
 * exact three-limb bf16 split of both operands (six limb products per multiply, f32 accumulation: f32-class accuracy,
 * see csrc/gemm_bf16x3.hip), replacing the library f32 GEMM behind [3P] mmcv FFN / MultiScaleDeformableAttention /
 * MultiheadAttention projections (torch.nn.functional.linear).  `w_packed`: pvsg_gemm_bf16x3_packed_elems(N, K) bf16
 * elements written by pvsg_gemm_bf16x3_pack from the (N, K) f32 weight (once per weight).  bias NULL = none.
 * Requires K % 16 == 0; other shapes return PVSG_ERR_UNSUPPORTED (the caller keeps its library GEMM). */
long long pvsg_gemm_bf16x3_packed_elems(int N, int K);
int pvsg_gemm_bf16x3_pack(const float* weight, void* w_packed, int N, int K, void* stream);
int pvsg_gemm_bf16x3(const float* a, const void* w_packed, const float* bias, float* out, long long M, int N, int K,
                     int relu, void* stream);

/* 1x1 convolution in NCHW (stride 1 or 2) on the same exact three-limb bf16 arithmetic, with the frozen-BN affine, the
 * bottleneck identity and ReLU in the epilogue:
 *   y[b, co, p] = act( conv1x1(x[b], w)[co, p] * scale[co] + shift[co] (+ residual[b, co, p]) )
 * `w_packed` = pvsg_gemm_bf16x3_pack of the (Cout, Cin) weight; scale / shift / residual may be NULL (1 / 0 / none).
 * in_scale / in_shift (B, Cin) or NULL: the input is first normalised per (image, channel) and rectified,
 * x' = relu(x * in_scale + in_shift) -- a GroupNorm + ReLU with known statistics folded into the operand staging
 * (mmdet ConvModule(norm GN, act ReLU) in front of the pixel decoder's mask_feature convolution); only with relu = 0 and
 * residual = NULL.
 * Replaces the library GEMM / MIOpen call + separate BN pass behind [3P] mmdet ResNet Bottleneck.conv1 / conv3 /
 * downsample and the pixel decoder's 1x1 convolutions.  Requires Cin % 16 == 0, Cin*H*W < 2^29. */
int pvsg_conv1x1_bf16x3(const float* x, const void* w_packed, const float* scale, const float* shift,
                        const float* residual, const float* in_scale, const float* in_shift, float* y, int B, int Cin,
                        int Cout, int H, int W, int stride, int relu, void* stream);
/* The weight of a 3x3 convolution as the (Cout, 9*Cin) matrix pvsg_conv3x3_bf16x3 / pvsg_conv3x3_f16x2 multiply by: K order
 * [block of 32 input channels][tap ky,kx][32 channels] -- column ((ci/32)*9 + ky*3 + kx)*32 + ci%32.  Pack THIS matrix with
 * pvsg_gemm_bf16x3_pack / pvsg_gemm_f16x2_pack (N = Cout, K = 9*Cin).  (Rounds 1-3 used tap-major order; ABI 7 callers must
 * not hand-roll the order.)  weight (Cout, Cin, 3, 3) f32, matrix (Cout, 9*Cin) f32, Cin % 32 == 0. */
int pvsg_conv3x3_weight_matrix(const float* weight, float* matrix, int Cout, int Cin, void* stream);

/* [3P] mmdet ResNet Bottleneck.conv2 (3x3, pad 1, stride 1 or 2) -> frozen BN -> ReLU on the split-bf16 kernel: implicit GEMM
 * over the nine taps.  w_packed = pvsg_gemm_bf16x3_pack (pvsg_gemm_f16x2_pack for the _f16x2 entry) of the (Cout, 9*Cin) matrix whose
 * K index runs [block of 32 input channels][tap ky*3+kx][32 channels]: w.reshape(Cout, Cin/32, 32, 3, 3).permute(0, 1, 3, 4, 2).
 * Cin % 32 == 0.  f32-MFMA forms: pvsg_conv3x3_winograd (stride 1), pvsg_conv3x3s2_affine (stride 2). */
int pvsg_conv3x3_bf16x3(const float* x, const void* w_packed, const float* scale, const float* shift, float* y, int B, int Cin,
                        int Cout, int H, int W, int stride, int relu, void* stream);

/* ---- the same five entry points on a TWO-limb f16 split (csrc/gemm_bf16x3.hip, F16 kernels) ------------------------------
 * An f16 carries 11 significant bits: a = a_h + a_l with a_h = f16(a), a_l = f16(a - a_h) is good to 2^-24 |a|, and a product
 * needs three limb products (hh, hl, lh) instead of six -- half the matrix work for the same f32-class dot product
 * (tests/test_gemm_f16x2.py measures both forms and the library's f32 GEMM against float64).  The missing exponent range of f16
 * is handled by exact power-of-two factors: the packed weight is scaled so that max|w| 2^e lies in [2^13, 2^14) (2^-e is
 * applied to the accumulator), the activation's low limb is staged as 2^11 (a - a_h) and multiplied by 2^-11 w_h, which the
 * kernels derive from the w_h fragment in registers.
 * Full accuracy for activations 2^-13 <= |a| <= 65504 (absolute error <= 2^-36 below), weights down to 2^-16 max|w|.
 * |a| > 65504 is NOT representable: the results of such a call are invalid and every kernel adds the number of staging
 * threads that met such an operand to *overflow (a device uint32 owned by the caller; may be NULL = not reported).  The host
 * mirror checks the counter at its next synchronisation point (openpvsg_amd/ops.py: split_overflow_check) and refuses to
 * hand out results; PVSG_SPLIT=bf16x3 selects the three-limb form, which has the whole f32 range.
 *   w_packed   pvsg_gemm_f16x2_packed_elems(N, K) = 2 * pad128(N) * K + 8 16-bit elements written by pvsg_gemm_f16x2_pack
 *              (arrays w_h, w_l in the staging order of the bf16 form, then max|w| and 2^-e as floats); K % 32 == 0
 * Arguments, layouts and the interfaces replaced are those of the _bf16x3 entry points above. */
long long pvsg_gemm_f16x2_packed_elems(int N, int K);
int pvsg_gemm_f16x2_pack(const float* weight, void* w_packed, int N, int K, void* stream);
int pvsg_gemm_f16x2(const float* a, const void* w_packed, const float* bias, float* out, long long M, int N, int K,
                    int relu, uint32_t* overflow, void* stream);
/* out = LayerNorm(residual + a w^T + bias) * gamma + beta over rows of N == 256: the [3P] mmcv encoder layer's projection (MSDA
 * output_proj / second FFN layer) + identity + LayerNorm in ONE launch on the 256 x 256-tile f16x2 kernel (a workgroup owns whole
 * rows; two-pass row statistics like F.layer_norm).  Replaces pvsg_gemm_* followed by pvsg_add_layernorm for those layers: the
 * projection's output never reaches HBM un-normalised.  K % 32 == 0; every pointer 16-byte aligned. */
int pvsg_gemm_f16x2_add_layernorm(const float* a, const void* w_packed, const float* bias, const float* residual,
                                  const float* gamma, const float* beta, float eps, float* out, long long M, int N, int K,
                                  uint32_t* overflow, void* stream);

/* Key AND value projections of one decoder level in one launch, straight from the encoder's token tensor: per decoder layer
 * `k = (memory + level_embed + pos) Wk^T + bk`, `v = (memory + level_embed) Wv^T + bv` ([3P] nn.MultiheadAttention in_proj on the
 * key / value inputs that models/mask2former/mask2former_head.py:421-436 builds and :457-468 passes).  level_embed and the
 * positional encoding are linear terms of the input, so they enter through the epilogue as two small tables; the (K, 256) key /
 * value INPUT tensors are never written.
 *   tokens (frames, S, 256) encoder memory; the level's tokens are rows start .. start + hw of every frame
 *   w_packed = pvsg_gemm_f16x2_pack of [Wk ; Wv] (512, 256)
 *   tab_cell (hw, 256) = (pe_yx + level_embed) Wk^T + bk;  tab_frame (zrows, 256) = pe_z Wk^T, frame f uses row f % zrows
 *   (image head: zrows = 1, zeros);  bias_v (256) = level_embed Wv^T + bv;  k_out, v_out (frames * hw, 256)
 * The token tensor must stay below 4 GB (else PVSG_ERR_UNSUPPORTED: use pvsg_decoder_kv_inputs + two pvsg_gemm_f16x2). */
int pvsg_decoder_kv_project_f16x2(const float* tokens, int frames, int S, int start, int hw, const void* w_packed,
                                  const float* tab_cell, const float* tab_frame, int zrows, const float* bias_v, float* k_out,
                                  float* v_out, uint32_t* overflow, void* stream);
int pvsg_conv1x1_f16x2(const float* x, const void* w_packed, const float* scale, const float* shift,
                       const float* residual, const float* in_scale, const float* in_shift, float* y, int B, int Cin,
                       int Cout, int H, int W, int stride, int relu, uint32_t* overflow, void* stream);
int pvsg_conv3x3_f16x2(const float* x, const void* w_packed, const float* scale, const float* shift, float* y, int B, int Cin,
                       int Cout, int H, int W, int stride, int relu, uint32_t* overflow, void* stream);
/* K-sliced forms for SMALL maps ([3P] mmdet ResNet Bottleneck.conv1 / conv2 / conv3 / downsample of layer2-4 when tools/test.py
 * feeds one 720p image per call: 46 x 80 and 23 x 40 pixels): a handful of 128 x 128 tiles with a K loop of up to 144 steps leave
 * most of the 256 CUs idle and expose a memory round trip per step.  `slices` workgroups per tile each multiply a range of the
 * K-steps (1x1, 3x3 stride 2) or of the 32-channel blocks (3x3 stride 1) and leave raw sums in `workspace` (slices * B * Cout * Ho *
 * Wo floats); one pass folds the slices in index order and applies scale / shift / identity / ReLU.  Needs Ho * Wo % 4 == 0.
 * pvsg_conv_slices(taps = 1 | 9, ...) returns the slice count worth using for a shape (1 = use the plain entry). */
int pvsg_conv_slices(int taps, int B, int Cin, int Cout, int H, int W, int stride);
int pvsg_conv1x1_f16x2_sliced(const float* x, const void* w_packed, const float* scale, const float* shift,
                              const float* residual, float* y, float* workspace, int slices, int B, int Cin, int Cout, int H,
                              int W, int stride, int relu, uint32_t* overflow, void* stream);
int pvsg_conv3x3_f16x2_sliced(const float* x, const void* w_packed, const float* scale, const float* shift, float* y,
                              float* workspace, int slices, int B, int Cin, int Cout, int H, int W, int stride, int relu,
                              uint32_t* overflow, void* stream);
int pvsg_mask_logits_f16x2(const float* mask_embed, const float* mask_feature, void* w_scratch, float* out, int B, int T,
                           int Q, int C, long long N, uint32_t* overflow, void* stream);
int pvsg_attn_mask_bits_f16x2(const float* mask_embed, const float* feature_lowres, void* w_scratch, uint32_t* bits,
                              uint32_t* flags, int B, int T, int Q, int C, long long N, uint32_t* overflow, void* stream);
/* The same bits from embeddings ALREADY packed by pvsg_decoder_rows_post (emb_pack_f16x2) into flag words ALREADY zero
 * (flags_zero): one launch for the whole batch.  mask2former_head.py:383-393, :453-454. */
int pvsg_attn_mask_bits_packed_f16x2(const void* emb_packed, const float* feature_lowres, uint32_t* bits, uint32_t* flags, int B,
                                     int T, int Q, int C, long long N, uint32_t* overflow, void* stream);

/* [3P] mmdet ResNet stem in one launch: conv1 (7x7 / 2, pad 3, 3 -> 64, no bias) -> frozen BN (scale, shift) -> ReLU ->
 * MaxPool2d(3, 2, 1):  x (N, 3, H, W) -> out (N, 64, Hp, Wp), Hc = (H-1)/2+1, Hp = (Hc-1)/2+1 (same for W).
 * `w_packed` = 21*64*8 floats written by pvsg_stem7x7_pack from the (64, 3, 7, 7) weight (once per weight). */
int pvsg_stem7x7_pack(const float* weight, float* w_packed, void* stream);
int pvsg_stem7x7_bn_relu_pool(const float* x, const float* w_packed, const float* scale, const float* shift, float* out,
                              int N, int H, int W, void* stream);
/* The same on the f16 matrix pipe (two-limb split; default for PVSG_SPLIT=f16x2): w_packed = pvsg_gemm_f16x2_pack(N = 64, K = 192) of
 * the (64, 192) matrix pvsg_stem7x7_f16x2_matrix writes (column 8 g + e = weight[ch][g / 7][g % 7][e]; the 8th tap of a row and
 * groups 21..23 are zero).  Inputs beyond +-65504 are counted into `overflow` (then use the f32 entry above). */
int pvsg_stem7x7_f16x2_matrix(const float* weight, float* matrix, void* stream);
int pvsg_stem7x7_f16x2_bn_relu_pool(const float* x, const void* w_packed, const float* scale, const float* shift, float* out, int N,
                                    int H, int W, uint32_t* overflow, void* stream);

/* [3P] torch.nn.GroupNorm (mmdet ConvModule norm_cfg=GN) as per-(image, channel) scale / shift:
 * GroupNorm(x)[b, c] == x[b, c] * scale[b*C + c] + shift[b*C + c]  (biased variance over the group's channels x pixels),
 * for consumers that apply it on the fly.  x (B, C, HW) f32; weight / bias (C) or NULL; workspace: B*G*128 doubles;
 * two launches with a fixed reduction order.  Requires (C/G)*HW % 4 == 0. */
int pvsg_group_norm_affine(const float* x, const float* weight, const float* bias, double* workspace, float* scale,
                           float* shift, int B, int C, int G, long long HW, float eps, void* stream);

/* The same scale / shift from partial sums produced by a convolution's epilogue (no pass over its output):
 * pvsg_conv1x1_f16x2_stats = pvsg_conv1x1_f16x2 (no input normalisation) that also writes, per (image, group of 8 output channels,
 * chunk), the sum and sum of squares of what it stores into gn_partials (B * Cout/8 * pvsg_conv1x1_stats_chunks(H, W, stride)
 * pairs of doubles); pvsg_group_norm_finish turns them into scale / shift as pvsg_group_norm_affine would.
 * [3P] mmcv ConvModule(norm_cfg=GN) of the pixel decoder's input / lateral convolutions. */
int pvsg_conv1x1_stats_chunks(int H, int W, int stride);
int pvsg_conv1x1_f16x2_stats(const float* x, const void* w_packed, const float* scale, const float* shift, const float* residual,
                             float* y, double* gn_partials, int B, int Cin, int Cout, int H, int W, int stride, int relu,
                             uint32_t* overflow, void* stream);
/* The 3x3 form (stride 1, pad 1, Cout > 64 in groups of 8): the FPN output convolution -> GN -> ReLU of
 * [3P] mmdet MSDeformAttnPixelDecoder.output_convs; chunks = pvsg_conv3x3_stats_chunks(H, W). */
int pvsg_conv3x3_stats_chunks(int H, int W);
int pvsg_conv3x3_f16x2_stats(const float* x, const void* w_packed, const float* scale, const float* shift, float* y,
                             double* gn_partials, int B, int Cin, int Cout, int H, int W, int relu, uint32_t* overflow, void* stream);
int pvsg_group_norm_finish(const double* partials, int nchunks, const float* weight, const float* bias, float* scale, float* shift,
                           int B, int C, int G, long long HW, float eps, void* stream);

/* [3P] mmdet ResNet Bottleneck.forward (64 planes, stride 1 = ResNet-50 layer1), the tail of one block and the head of the next in
 * one pass over the pixels:  y = relu(conv3(mid) * scale3 + shift3 + identity)  (64 -> 256 channels) and, from y while it is in
 * registers,  mid_next = relu(conv1_next(y) * scale1n + shift1n)  (256 -> Cnext) -- the next block's conv1 never re-reads y.
 * mid (B,64,H,W) = the block's conv2 -> bn2 -> relu output; identity, y (B,256,H,W); mid_next (B,Cnext,H,W), Cnext = 64 (next block
 * of the stage) or 128 (first block of the NEXT stage, whose conv1 runs at this resolution: style='pytorch').
 * y_stride2 (B,256,ceil(H/2),W/2) or NULL: y's even rows / columns once more, compact -- what the next stage's stride-2 downsample
 * convolution reads (pvsg_conv1x1_f16x2 with stride 1 on it); needs even W.
 * w3_packed = pvsg_gemm_f16x2_pack of conv3's (256, 64) matrix; w1n_packed = pvsg_gemm_f16x2_pack of
 * pvsg_bottleneck_next_weight_matrix(next conv1's (Cnext, 256) matrix) (the K order the kernel multiplies in).  w1n_packed, scale1n,
 * shift1n, mid_next all NULL: the first line only.
 * identity == NULL: the head of the stage's FIRST block from one read of its input x (given as `mid`, 64 channels):
 * y = downsample_conv(x) * scale3 + shift3 (w3_packed = the downsample's (256, 64) pack; no ReLU) and mid_next = relu(conv1(x) *
 * scale1n + shift1n) with w1n_packed = the plain pvsg_gemm_f16x2_pack of conv1's (64, 64) matrix (Cnext = 64).
 * Exactly these channel counts, even H*W. */
int pvsg_bottleneck_next_weight_matrix(const float* weight, float* matrix, int Cn, int K, void* stream);
int pvsg_bottleneck_tail_f16x2(const float* mid, const void* w3_packed, const float* scale3, const float* shift3, const float* identity,
                               float* y, float* y_stride2, const void* w1n_packed, const float* scale1n, const float* shift1n,
                               float* mid_next, int B, int Cmid, int Cout, int Cnext, int H, int W, uint32_t* overflow, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OPENPVSG_HIP_H_ */
