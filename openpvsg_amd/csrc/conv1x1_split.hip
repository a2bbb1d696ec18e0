// C-ABI entries of the split-arithmetic 1x1 convolution (pvsg_conv1x1_{f16x2,bf16x3}[_stats]) and of the two mask GEMMs that run
// on the same kernels (pvsg_mask_logits_*, pvsg_attn_mask_bits_*): kernels in split_conv1x1.h, arithmetic in split_common.h.
#include "split_conv1x1.h"

// weight packs (token_gemm.hip): the mask GEMMs pack their query rows on the fly
extern "C" long long pvsg_gemm_f16x2_packed_elems(int N, int K);
extern "C" long long pvsg_gemm_bf16x3_packed_elems(int N, int K);
extern "C" int pvsg_gemm_f16x2_pack(const float* weight, void* w_packed, int N, int K, void* stream);
extern "C" int pvsg_gemm_bf16x3_pack(const float* weight, void* w_packed, int N, int K, void* stream);

static int conv1x1_split_run(const float* x, const void* w_packed, const float* scale, const float* shift,
                             const float* residual, const float* in_scale, const float* in_shift, float* y, int B, int Cin,
                             int Cout, int H, int W, int stride, int relu, bool f16, uint32_t* overflow, void* stream,
                             double* gn_part = nullptr, float* ws = nullptr, int slices = 1) {
  using namespace pvsg;
  const char* nm = f16 ? "conv1x1_f16x2" : "conv1x1_bf16x3";
  PVSG_REQUIRE(x && w_packed && y, "%s: null pointer argument", nm);
  PVSG_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "%s: bad shape", nm);
  PVSG_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "%s: in_scale and in_shift go together", nm);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int Cpad = (Cout + 127) / 128 * 128;
  // the epilogue's 32-bit store offsets: pixel offset + padded channel row * plane must stay below 2^31 (out-of-range pixels
  // carry 0x80000000 and rely on the bounds check)
  if (Cin % (f16 ? 32 : GB_K) || Cout % 4 || (long long)Cin * H * W >= (1LL << 29) || (long long)Cpad * Ho * Wo >= (1LL << 29))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for Cin %% %d == 0, Cout %% 4 == 0, C*H*W < 2^29 (got Cin=%d Cout=%d H=%d W=%d)",
                   nm, f16 ? 32 : GB_K, Cin, Cout, H, W);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15u),
               "%s: w_packed, scale and shift must be 16-byte aligned", nm);
  const int tiles_c = Cpad / GB_M, tiles_p = (Ho * Wo + GB_N - 1) / GB_N;
  const long long blocks = (long long)B * tiles_c * tiles_p;
  PVSG_REQUIRE(blocks < (1LL << 31), "%s: too many blocks", nm);
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const __bf16* wp = static_cast<const __bf16*>(w_packed);
  unsigned* const noflags = nullptr;
  if (slices > 1) {
    // K slices: `slices` workgroups per tile leave raw sums in `ws` (slices x B x Cout x Ho x Wo), one pass folds them and applies
    // the affine / identity / ReLU
    if (!f16 || !ws || in_scale || gn_part || (Ho * Wo) % 4 || slices > Cin / 32 || blocks * slices >= (1LL << 31))
      return set_err(PVSG_ERR_UNSUPPORTED, "%s: K slices need the f16x2 form, a workspace, Ho*Wo %% 4 == 0, slices <= Cin / 32", nm);
    PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15u),
                 "%s: workspace, y and residual must be 16-byte aligned", nm);
    const long long slice_elems = (long long)B * Cout * Ho * Wo;
    const dim3 gs((unsigned)(blocks * slices));
    const float* nulf = nullptr;
    if (Cout <= 64)
      hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<false, false, false, false, 64, 1, true>), gs, block, 0, st, x, wp, nulf, nulf, nulf,
                         nulf, nulf, ws, Cin, Cout, Cpad, H * W, W, Ho * Wo, Wo, stride, tiles_c, tiles_p, noflags, overflow, 0, 0LL,
                         (double*)nullptr, slices, slice_elems);
    else
      hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<false, false, false, false, 128, 1, true>), gs, block, 0, st, x, wp, nulf, nulf, nulf,
                         nulf, nulf, ws, Cin, Cout, Cpad, H * W, W, Ho * Wo, Wo, stride, tiles_c, tiles_p, noflags, overflow, 0, 0LL,
                         (double*)nullptr, slices, slice_elems);
    PVSG_LAUNCH_CHECK(nm);
    launch_conv_slices_finish(ws, slices, slice_elems, scale, shift, residual, y, Cout, Ho * Wo, relu, st);
    PVSG_LAUNCH_CHECK(nm);
    return PVSG_OK;
  }
  // PVSG_GEMM_K32=0: the 32x32x16 / K = 16 kernel for every shape, =1: the K = 32 kernel wherever Cin allows (A/B tests);
  // default: K = 32 except on the small stride-1 maps (23 x 40 at 720p: layer4 and its input convolution measured 3-5 %
  // slower there, profiles/r03_conv1x1_bf16x3_bench.jsonl).  The f16 form exists on the K = 32 kernel only.
  const char* sel = getenv("PVSG_GEMM_K32");
  const bool k32 = f16 || (Cin % 32 == 0 && !(sel && sel[0] == '0') && ((sel && sel[0] == '1') || stride == 2 || Ho * Wo >= 2048));
  const bool tm64 = k32 && Cout <= 64;
  if (gn_part && (!f16 || tm64 || Cout % 8))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: GroupNorm statistics come with the f16x2 form, Cout > 64, groups of 8 channels", nm);
#define PVSG_C1_K32(R, S, NORM, TMV, F)                                                                                  \
  hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<R, S, NORM, false, TMV, 1, F>), grid, block, 0, st, x, wp, scale, shift,  \
                     residual, in_scale, in_shift, y, Cin, Cout, Cpad, H * W, W, Ho * Wo, Wo, stride, tiles_c, tiles_p, \
                     noflags, overflow, 0, 0LL, gn_part)
#define PVSG_C1_LAUNCH(R, S)                                                                                            \
  do {                                                                                                                  \
    if (f16) { if (tm64) PVSG_C1_K32(R, S, false, 64, true); else PVSG_C1_K32(R, S, false, 128, true); }                 \
    else if (tm64) PVSG_C1_K32(R, S, false, 64, false);                                                                 \
    else if (k32) PVSG_C1_K32(R, S, false, 128, false);                                                                 \
    else                                                                                                                \
      hipLaunchKernelGGL((conv1x1_bf16x3_kernel<R, S, false>), grid, block, 0, st, x, wp, scale, shift, residual,      \
                         in_scale, in_shift, y, Cin, Cout, Cpad, H * W, W, Ho * Wo, Wo, stride, tiles_c, tiles_p);       \
  } while (0)
  if (in_scale) {          // normalised input: the pixel decoder's mask-feature convolution (no ReLU / identity behind it)
    if (relu || residual)
      return set_err(PVSG_ERR_UNSUPPORTED, "%s: in_scale / in_shift come without relu / residual", nm);
    if (f16) PVSG_C1_K32(false, false, true, 128, true);
    else if (k32) PVSG_C1_K32(false, false, true, 128, false);
    else
      hipLaunchKernelGGL((conv1x1_bf16x3_kernel<false, false, true>), grid, block, 0, st, x, wp, scale, shift, residual, in_scale,
                         in_shift, y, Cin, Cout, Cpad, H * W, W, Ho * Wo, Wo, stride, tiles_c, tiles_p);
  } else if (relu) {
    if (residual) PVSG_C1_LAUNCH(true, true); else PVSG_C1_LAUNCH(true, false);
  } else {
    if (residual) PVSG_C1_LAUNCH(false, true); else PVSG_C1_LAUNCH(false, false);
  }
#undef PVSG_C1_LAUNCH
#undef PVSG_C1_K32
  PVSG_LAUNCH_CHECK(nm);
  return PVSG_OK;
}

extern "C" int pvsg_conv1x1_bf16x3(const float* x, const void* w_packed, const float* scale, const float* shift,
                                   const float* residual, const float* in_scale, const float* in_shift, float* y, int B,
                                   int Cin, int Cout, int H, int W, int stride, int relu, void* stream) {
  return conv1x1_split_run(x, w_packed, scale, shift, residual, in_scale, in_shift, y, B, Cin, Cout, H, W, stride, relu, false,
                           nullptr, stream);
}

extern "C" int pvsg_conv1x1_f16x2(const float* x, const void* w_packed, const float* scale, const float* shift,
                                  const float* residual, const float* in_scale, const float* in_shift, float* y, int B,
                                  int Cin, int Cout, int H, int W, int stride, int relu, uint32_t* overflow, void* stream) {
  return conv1x1_split_run(x, w_packed, scale, shift, residual, in_scale, in_shift, y, B, Cin, Cout, H, W, stride, relu, true,
                           overflow, stream);
}

// Small maps (one 720p image: 46 x 80 and 23 x 40 in layer3 / layer4): a handful of 128 x 128 tiles each walking a K loop of up to 64
// steps -- 30 .. 60 workgroups on 256 CUs, every step an exposed memory round trip (92 us for 2048 -> 512 channels at 920 pixels,
// 3.8 GF).  pvsg_conv_slices says into how many K slices a convolution of this shape should be cut (1 = not worth it); the
// `_sliced` entries run `slices` workgroups per tile into a workspace of slices * B * Cout * Ho * Wo floats and fold them.
//   taps = 1: [3P] Bottleneck.conv1 / conv3 / downsample;  taps = 9: Bottleneck.conv2 (stride 1: halo kernel, slices over the
//   32-channel blocks; stride 2: the tap-by-tap form)
extern "C" int pvsg_conv_slices(int taps, int B, int Cin, int Cout, int H, int W, int stride) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || (taps != 1 && taps != 9) || (stride != 1 && stride != 2)) return 1;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long long HWo = (long long)Ho * Wo;
  if (Cin % 32 || Cout % 4 || HWo % 4) return 1;
  const int Cpad = (Cout + 127) / 128 * 128;
  long long blocks;
  int units, min_units;                                           // K-steps (halo form: 32-channel blocks of nine steps)
  if (taps == 9 && stride == 1) {
    const bool wide = Cout > 64;
    blocks = (long long)B * (wide ? Cpad / 128 : (Cout + 63) / 64) * ((W + 15) / 16) * (wide ? (H + 7) / 8 : (H + 15) / 16);
    units = Cin / 32;
    min_units = 2;
  } else {
    blocks = (long long)B * (Cpad / 128) * ((HWo + 127) / 128);
    units = taps * Cin / 32;
    min_units = 8;
  }
  // lab knobs (scripts/lab/slices_ab.sh): workgroups aimed at, least K-steps per slice of the tap-by-tap / 1x1 form
  static const int target = [] { const char* e = getenv("PVSG_SLICE_TARGET"); return e ? atoi(e) : 512; }();
  static const int min_steps = [] { const char* e = getenv("PVSG_SLICE_MIN_STEPS"); return e ? atoi(e) : 8; }();
  static const int max_blocks = [] { const char* e = getenv("PVSG_SLICE_MAX_BLOCKS"); return e ? atoi(e) : 160; }();
  if (!(taps == 9 && stride == 1)) min_units = min_steps > 0 ? min_steps : 8;
  if (blocks > max_blocks) return 1;
  long long s = (target + blocks - 1) / blocks;
  if (s > units / min_units) s = units / min_units;
  if (s > 16) s = 16;
  while (s > 1 && s * B * Cout * HWo * 4 > (64LL << 20)) --s;
  return s < 2 ? 1 : (int)s;
}

extern "C" int pvsg_conv1x1_f16x2_sliced(const float* x, const void* w_packed, const float* scale, const float* shift,
                                         const float* residual, float* y, float* workspace, int slices, int B, int Cin, int Cout,
                                         int H, int W, int stride, int relu, uint32_t* overflow, void* stream) {
  PVSG_REQUIRE(slices >= 1, "conv1x1_f16x2_sliced: slices must be >= 1");
  return conv1x1_split_run(x, w_packed, scale, shift, residual, nullptr, nullptr, y, B, Cin, Cout, H, W, stride, relu, true, overflow,
                           stream, nullptr, workspace, slices);
}

// pvsg_conv1x1_f16x2 that also leaves the GroupNorm statistics of its OUTPUT behind ([3P] mmcv ConvModule(norm_cfg=GN): conv -> GN;
// groups of 8 channels): gn_partials receives B * (Cout / 8) * pvsg_conv1x1_stats_chunks(H, W, stride) pairs of doubles (sum, sum of
// squares), to be turned into per-(image, channel) scale / shift by pvsg_group_norm_finish.
extern "C" int pvsg_conv1x1_stats_chunks(int H, int W, int stride) {
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  return 2 * ((Ho * Wo + pvsg::GB_N - 1) / pvsg::GB_N);
}
extern "C" int pvsg_conv1x1_f16x2_stats(const float* x, const void* w_packed, const float* scale, const float* shift,
                                        const float* residual, float* y, double* gn_partials, int B, int Cin, int Cout, int H, int W,
                                        int stride, int relu, uint32_t* overflow, void* stream) {
  PVSG_REQUIRE(gn_partials, "conv1x1_f16x2_stats: null pointer argument");
  return conv1x1_split_run(x, w_packed, scale, shift, residual, nullptr, nullptr, y, B, Cin, Cout, H, W, stride, relu, true, overflow,
                           stream, gn_partials);
}

// einsum('bqc,b[t]chw->b[t]qhw') (mask2former_head.py:382, mask2former_video_head.py:344) on the split kernels: per batch
// element a 1x1 "convolution" of the (T, C, N) mask features with the Q mask embeddings as the weight (packed on the fly:
// Q x C is 100 x 256), output (T, Q, N).  Same f32-class arithmetic as above; the f32-MFMA form stays as
// pvsg_mask_logits_forward (csrc/mask_gemm.hip).  w_scratch: B * pvsg_gemm_{bf16x3,f16x2}_packed_elems(Q, C) 16-bit elements.
static int mask_logits_split_run(const float* mask_embed, const float* mask_feature, void* w_scratch, float* out, int B, int T,
                                 int Q, int C, long long N, bool f16, uint32_t* overflow, void* stream) {
  using namespace pvsg;
  const char* nm = f16 ? "mask_logits_f16x2" : "mask_logits_bf16x3";
  PVSG_REQUIRE(mask_embed && mask_feature && w_scratch && out, "%s: null pointer argument", nm);
  PVSG_REQUIRE(B > 0 && T > 0 && Q > 0 && C > 0 && N > 0, "%s: bad shape", nm);
  const long long Qpad = (Q + 127) / 128 * 128;
  if (C % (f16 ? 32 : GB_K) || Q % 4 || N >= (1LL << 31) || (long long)C * N >= (1LL << 29) || Qpad * N >= (1LL << 29))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for C %% %d == 0, Q %% 4 == 0, C*N and pad128(Q)*N < 2^29 (got Q=%d C=%d N=%lld)",
                   nm, f16 ? 32 : GB_K, Q, C, N);
  const long long welems = f16 ? pvsg_gemm_f16x2_packed_elems(Q, C) : pvsg_gemm_bf16x3_packed_elems(Q, C);
  for (int b = 0; b < B; ++b) {
    __bf16* wp = static_cast<__bf16*>(w_scratch) + (size_t)b * welems;
    int rc = f16 ? pvsg_gemm_f16x2_pack(mask_embed + (size_t)b * Q * C, wp, Q, C, stream)
                 : pvsg_gemm_bf16x3_pack(mask_embed + (size_t)b * Q * C, wp, Q, C, stream);
    if (rc != PVSG_OK) return rc;
    rc = conv1x1_split_run(mask_feature + (size_t)b * T * C * N, wp, nullptr, nullptr, nullptr, nullptr, nullptr,
                           out + (size_t)b * T * Q * N, T, C, Q, 1, (int)N, 1, 0, f16, overflow, stream);
    if (rc != PVSG_OK) return rc;
  }
  return PVSG_OK;
}

extern "C" int pvsg_mask_logits_bf16x3(const float* mask_embed, const float* mask_feature, void* w_scratch, float* out, int B,
                                       int T, int Q, int C, long long N, void* stream) {
  return mask_logits_split_run(mask_embed, mask_feature, w_scratch, out, B, T, Q, C, N, false, nullptr, stream);
}

extern "C" int pvsg_mask_logits_f16x2(const float* mask_embed, const float* mask_feature, void* w_scratch, float* out, int B,
                                      int T, int Q, int C, long long N, uint32_t* overflow, void* stream) {
  return mask_logits_split_run(mask_embed, mask_feature, w_scratch, out, B, T, Q, C, N, true, overflow, stream);
}

// Attention-mask bits of a decoder level straight from the down-sampled mask features (mask2former_head.py:383-393,
// video_head.py:346-357; the all-masked-row test of mask2former_head.py:453-454 becomes the flag words) on the split
// kernels: same record format as pvsg_attn_mask_bits_forward (csrc/mask_gemm.hip), which stays as the f32-MFMA form.
static int attn_mask_bits_split_run(const float* mask_embed, const float* feature_lowres, void* w_scratch, uint32_t* bits,
                                    uint32_t* flags, int B, int T, int Q, int C, long long N, bool f16, uint32_t* overflow,
                                    void* stream) {
  using namespace pvsg;
  const char* nm = f16 ? "attn_mask_bits_f16x2" : "attn_mask_bits_bf16x3";
  PVSG_REQUIRE(mask_embed && feature_lowres && w_scratch && bits && flags, "%s: null pointer argument", nm);
  PVSG_REQUIRE(B > 0 && T > 0 && Q > 0 && C > 0 && N > 0, "%s: bad shape", nm);
  if (C % (f16 ? 32 : GB_K) || Q > GB_M || (long long)C * N >= (1LL << 29) || (reinterpret_cast<uintptr_t>(bits) & 15u))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for C %% %d == 0, Q <= 128, C*N < 2^29, 16B-aligned bits (got Q=%d C=%d N=%lld)",
                   nm, f16 ? 32 : GB_K, Q, C, N);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = zero_words_async(flags, (size_t)B * 4 * sizeof(uint32_t), st);
  if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "%s: memset: %s", nm, hipGetErrorString(e));
  const long long welems = f16 ? pvsg_gemm_f16x2_packed_elems(Q, C) : pvsg_gemm_bf16x3_packed_elems(Q, C);
  const int tiles_p = (int)((N + GB_N - 1) / GB_N);
  const float* nul = nullptr;
  for (int b = 0; b < B; ++b) {
    __bf16* wp = static_cast<__bf16*>(w_scratch) + (size_t)b * welems;
    const int rc = f16 ? pvsg_gemm_f16x2_pack(mask_embed + (size_t)b * Q * C, wp, Q, C, stream)
                       : pvsg_gemm_bf16x3_pack(mask_embed + (size_t)b * Q * C, wp, Q, C, stream);
    if (rc != PVSG_OK) return rc;
    const char* sel = getenv("PVSG_GEMM_K32");
    const float* fl = feature_lowres + (size_t)b * T * C * N;
    float* rec = reinterpret_cast<float*>(bits + (size_t)b * T * N * 4);
    const dim3 grid((unsigned)(T * tiles_p)), block(256);
    if (f16)
      hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<false, false, false, true, 128, 1, true>), grid, block, 0, st, fl, wp, nul, nul,
                         nul, nul, nul, rec, C, Q, GB_M, (int)N, (int)N, (int)N, (int)N, 1, 1, tiles_p, flags + (size_t)b * 4, overflow);
    else if (C % 32 == 0 && !(sel && sel[0] == '0'))
      hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<false, false, false, true>), grid, block, 0, st, fl, wp, nul, nul, nul, nul, nul,
                         rec, C, Q, GB_M, (int)N, (int)N, (int)N, (int)N, 1, 1, tiles_p, flags + (size_t)b * 4, (unsigned*)nullptr);
    else
      hipLaunchKernelGGL((conv1x1_bf16x3_kernel<false, false, false, true>), grid, block, 0, st, fl, wp, nul, nul, nul, nul, nul,
                         rec, C, Q, GB_M, (int)N, (int)N, (int)N, (int)N, 1, 1, tiles_p, flags + (size_t)b * 4);
    PVSG_LAUNCH_CHECK(nm);
  }
  return PVSG_OK;
}

// The bits from embeddings that are ALREADY packed (pvsg_decoder_rows_post writes them in its epilogue, one exact power-of-two
// scale per query row -- the bits are signs) into flag words that are already zero: ONE launch for the whole batch, where the
// entry above issues zero + amax + pack + GEMM per batch element.
extern "C" int pvsg_attn_mask_bits_packed_f16x2(const void* emb_packed, const float* feature_lowres, uint32_t* bits,
                                                uint32_t* flags, int B, int T, int Q, int C, long long N, uint32_t* overflow,
                                                void* stream) {
  using namespace pvsg;
  const char* nm = "attn_mask_bits_packed_f16x2";
  PVSG_REQUIRE(emb_packed && feature_lowres && bits && flags, "%s: null pointer argument", nm);
  PVSG_REQUIRE(B > 0 && T > 0 && Q > 0 && C > 0 && N > 0, "%s: bad shape", nm);
  if (C % 32 || Q > GB_M || (long long)C * N >= (1LL << 29) || (long long)B * T * N >= (1LL << 27) ||
      ((reinterpret_cast<uintptr_t>(bits) | reinterpret_cast<uintptr_t>(emb_packed)) & 15u))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for C %% 32 == 0, Q <= 128, C*N < 2^29, B*T*N < 2^27, 16B-aligned buffers "
                   "(got B=%d T=%d Q=%d C=%d N=%lld)", nm, B, T, Q, C, N);
  const int tiles_p = (int)((N + GB_N - 1) / GB_N);
  const float* nul = nullptr;
  hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<false, false, false, true, 128, 1, true>), dim3((unsigned)(B * T * tiles_p)), dim3(256),
                     0, static_cast<hipStream_t>(stream), feature_lowres, static_cast<const __bf16*>(emb_packed), nul, nul, nul, nul,
                     nul, reinterpret_cast<float*>(bits), C, Q, GB_M, (int)N, (int)N, (int)N, (int)N, 1, 1, tiles_p, flags, overflow,
                     T, pvsg_gemm_f16x2_packed_elems(Q, C));
  PVSG_LAUNCH_CHECK(nm);
  return PVSG_OK;
}

extern "C" int pvsg_attn_mask_bits_bf16x3(const float* mask_embed, const float* feature_lowres, void* w_scratch, uint32_t* bits,
                                          uint32_t* flags, int B, int T, int Q, int C, long long N, void* stream) {
  return attn_mask_bits_split_run(mask_embed, feature_lowres, w_scratch, bits, flags, B, T, Q, C, N, false, nullptr, stream);
}

extern "C" int pvsg_attn_mask_bits_f16x2(const float* mask_embed, const float* feature_lowres, void* w_scratch, uint32_t* bits,
                                         uint32_t* flags, int B, int T, int Q, int C, long long N, uint32_t* overflow, void* stream) {
  return attn_mask_bits_split_run(mask_embed, feature_lowres, w_scratch, bits, flags, B, T, Q, C, N, true, overflow, stream);
}

