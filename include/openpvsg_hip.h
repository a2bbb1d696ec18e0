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

#ifdef __cplusplus
}
#endif
#endif /* OPENPVSG_HIP_H_ */
