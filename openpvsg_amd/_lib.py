"""ctypes binding of the C ABI in include/openpvsg_hip.h.

torch is imported first so that the library's NEEDED `libamdhip64.so.7` resolves to the HIP
runtime torch already loaded (one runtime per process: device pointers and streams are shared).
There is no fallback: if the shared library is missing or a symbol is absent this raises, and
every op in `ops.py` goes through here.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
# PVSG_LIB_PATH: a lab build of the same library (scripts/lab/abl_split.sh); the product always loads the in-tree one
LIB_PATH = os.environ.get('PVSG_LIB_PATH') or os.path.join(_HERE, 'lib', 'libopenpvsg_hip.so')

_c_f = ctypes.c_void_p  # device pointers travel as raw addresses
_i = ctypes.c_int
_f = ctypes.c_float
_ll = ctypes.c_longlong



class DecoderLayer(ctypes.Structure):
    """pvsg_decoder_layer (include/openpvsg_hip.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        'xo_w', 'xo_b', 'n0_g', 'n0_b', 'sa_in_w', 'sa_in_b', 'sa_out_w', 'sa_out_b', 'n1_g', 'n1_b',
        'f1_w', 'f1_b', 'f2_w', 'f2_b', 'n2_g', 'n2_b')] + [('embed_dims', _i), ('num_heads', _i), ('ffn_dim', _i)]


class DecoderHead(ctypes.Structure):
    """pvsg_decoder_head (include/openpvsg_hip.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        'pn_g', 'pn_b', 'cls_w', 'cls_b', 'm0_w', 'm0_b', 'm1_w', 'm1_b', 'm2_w', 'm2_b')] + [('num_cls_out', _i)]


class EncoderLayer(ctypes.Structure):
    """pvsg_encoder_layer (include/openpvsg_hip.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        'in_w', 'in_b', 'out_w', 'out_b', 'n1_g', 'n1_b', 'f1_w', 'f1_b', 'f2_w', 'f2_b', 'n2_g', 'n2_b')] + [
        ('d_model', _i), ('num_heads', _i), ('ffn_dim', _i), ('eps1', _f), ('eps2', _f)]


class RelationTail(ctypes.Structure):
    """pvsg_relation_tail (include/openpvsg_hip.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        'ln_g', 'ln_b', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'head_w', 'head_b', 'filter')] + [
        ('dim', _i), ('num_relations', _i), ('eps', _f)]


# name -> argtypes; must list every function include/openpvsg_hip.h declares
# (tests/test_capi.py cross-checks this table against the header).
SIGNATURES = {
    'pvsg_ms_deform_attn_forward': [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _i, _i, _i, _i, _c_f],
    'pvsg_mask_logits_forward': [_c_f, _c_f, _c_f, _i, _i, _i, _i, _i, _c_f],
    'pvsg_mask_logits_bf16x3': [_c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _ll, _c_f],
    'pvsg_attn_mask_bits_bf16x3': [_c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _ll, _c_f],
    'pvsg_attn_mask_bits_forward': [_c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _i, _c_f],
    'pvsg_attn_mask_pack': [_c_f, _c_f, _c_f, _i, _i, _i, _i, _c_f],
    'pvsg_center_downsample': [_c_f, _c_f, _c_f, _c_f, _ll, _i, _i, _c_f],
    'pvsg_xattn_num_splits': [_i, _ll],
    'pvsg_masked_xattn_partial': [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _ll, _i, _i, _i, _c_f],
    'pvsg_masked_xattn_partial_strided': [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _ll, _i, _i, _i, _ll, _c_f],
    'pvsg_xattn_combine': [_c_f, _c_f, _c_f, _i, _i, _i, _i, _i, _c_f],
    'pvsg_xattn_merge_local': [_c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _i, _c_f],
    'pvsg_xattn_combine_packed': [_c_f, _c_f, _i, _i, _i, _i, _i, _c_f],
    'pvsg_pack_rows_weight': [_c_f, _c_f, _i, _i, _c_f],
    'pvsg_decoder_rows_pre': [ctypes.POINTER(DecoderLayer), _c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _c_f],
    'pvsg_decoder_rows_post_workspace_bytes': [_i, _i],
    'pvsg_decoder_rows_post': [ctypes.POINTER(DecoderLayer), ctypes.POINTER(DecoderHead)] + [_c_f] * 12 + [_i, _i, _c_f],
    'pvsg_rows_f16x2_packed_floats': [_i, _i],
    'pvsg_pack_rows_weight_f16x2': [_c_f, _c_f, _i, _i, _c_f],
    'pvsg_decoder_rows_pre_f16x2': [ctypes.POINTER(DecoderLayer), _c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _c_f, _c_f],
    'pvsg_decoder_rows_post_f16x2': [ctypes.POINTER(DecoderLayer), ctypes.POINTER(DecoderHead)] + [_c_f] * 12 + [_i, _i, _c_f, _c_f],
    'pvsg_rel_qkv': [ctypes.POINTER(EncoderLayer), _i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _ll, _i, _c_f],
    'pvsg_rel_encoder_layer': [ctypes.POINTER(EncoderLayer), ctypes.POINTER(EncoderLayer), _i, _c_f, _ll, _c_f, _c_f, _c_f, _i, _i,
                               _ll, _ll, _c_f],
    'pvsg_rel_attention': [_c_f, _c_f, _i, _i, _ll, _ll, _i, _i, _c_f],
    'pvsg_rel_conv5': [_c_f, _c_f, _c_f, _c_f, _i, _i, _i, _c_f],
    'pvsg_rel_tail_workspace_bytes': [_i, _i],
    'pvsg_rel_tail': [ctypes.POINTER(RelationTail), _c_f, _c_f, _c_f, _c_f, _i, _i, _c_f],
    'pvsg_top_pairs': [_c_f, _c_f, _i, _i, _c_f],
    'pvsg_pair_prepare_weights': [_c_f, _c_f, _i, _i, _c_f],
    'pvsg_pair_score_forward': [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _c_f],
    'pvsg_panoptic_fuse': [_c_f] * 8 + [_i] * 13 + [ctypes.c_double, _i, _c_f],
    'pvsg_instance_masks': [_c_f] * 5 + [_i] * 12 + [_c_f],
    'pvsg_panoptic_select': [_c_f, _c_f, _i, _i, _f, _c_f, _c_f],
    'pvsg_panoptic_fuse_sel': [_c_f] * 6 + [_i] * 12 + [ctypes.c_double, _i, _c_f],
    'pvsg_tube_index_table_words': [],
    'pvsg_tube_index': [_c_f, _c_f, _i, _i, _i, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f],
    'pvsg_tube_scatter': [_c_f, _ll, _c_f, _c_f, _c_f, _i, _i, _i, _c_f],
    'pvsg_msda_fused_forward': [_c_f, _ll, _c_f, _ll, _c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _i, _i, _i, _c_f],
    'pvsg_msda_proj_ln_forward': [_c_f, _ll, _c_f, _ll] + [_c_f] * 10 + [_i] * 7 + [_f, _c_f],
    'pvsg_add_layernorm': [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _ll, _i, _f, _c_f],
    'pvsg_affine_act_nchw': [_c_f, _c_f, _c_f, _c_f, _c_f, _ll, _i, _ll, _i, _c_f],
    'pvsg_minvis_chain_workspace_bytes': [_i, _i, _i],
    'pvsg_minvis_chain': [_c_f, _c_f, _c_f, _i, _i, _i, _i, _c_f],
    'pvsg_mask_embed_forward': [_c_f] * 7 + [_i] * 5 + [_c_f],
    'pvsg_reconsdot_workspace_bytes': [_i, _i, _i, _i],
    'pvsg_rle_counts_to_chars': [_c_f, _c_f, _i, _c_f, _c_f],
    'pvsg_reconsdot_cost': [_c_f] * 3 + [_i] * 4 + [_f, _c_f, _c_f, _c_f, _c_f],
    'pvsg_fpn_merge_up2x': [_c_f] * 5 + [_ll, _i, _i, _c_f],
    'pvsg_stem_bn_relu_pool': [_c_f] * 4 + [_ll, _i, _i, _i, _c_f],
    'pvsg_nchw_to_tokens': [_c_f] * 4 + [_i, _i, _i, _ll, _c_f],
    'pvsg_tokens_to_nchw': [_c_f, _c_f, _i, _i, _i, _ll, _c_f],
    'pvsg_decoder_kv_inputs': [_c_f] * 5 + [_ll, _i, _i, _ll, _ll, _c_f],
    'pvsg_conv1x1_affine': [_c_f] * 6 + [_i, _i, _i, _ll, _i, _c_f],
    'pvsg_conv3x3_winograd_pack': [_c_f, _c_f, _i, _i, _c_f],
    'pvsg_conv3x3_winograd': [_c_f] * 5 + [_i] * 6 + [_c_f],
    'pvsg_conv3x3s2_pack': [_c_f, _c_f, _i, _i, _c_f],
    'pvsg_gemm_bf16x3_packed_elems': [_i, _i],
    'pvsg_gemm_bf16x3_pack': [_c_f, _c_f, _i, _i, _c_f],
    'pvsg_gemm_bf16x3': [_c_f, _c_f, _c_f, _c_f, _ll, _i, _i, _i, _c_f],
    'pvsg_conv1x1_bf16x3': [_c_f] * 8 + [_i] * 7 + [_c_f],
    'pvsg_conv3x3_bf16x3': [_c_f] * 5 + [_i] * 7 + [_c_f],
    'pvsg_conv3x3_weight_matrix': [_c_f, _c_f, _i, _i, _c_f],
    'pvsg_gemm_f16x2_packed_elems': [_i, _i],
    'pvsg_gemm_f16x2_pack': [_c_f, _c_f, _i, _i, _c_f],
    'pvsg_gemm_f16x2': [_c_f, _c_f, _c_f, _c_f, _ll, _i, _i, _i, _c_f, _c_f],
    'pvsg_gemm_f16x2_add_layernorm': [_c_f] * 6 + [_f, _c_f, _ll, _i, _i, _c_f, _c_f],
    'pvsg_decoder_kv_project_f16x2': [_c_f, _i, _i, _i, _i, _c_f, _c_f, _c_f, _i, _c_f, _c_f, _c_f, _c_f, _c_f],
    'pvsg_conv1x1_f16x2': [_c_f] * 8 + [_i] * 7 + [_c_f, _c_f],
    'pvsg_conv3x3_f16x2': [_c_f] * 5 + [_i] * 7 + [_c_f, _c_f],
    'pvsg_conv1x1_f16x2_sliced': [_c_f] * 7 + [_i] * 8 + [_c_f, _c_f],
    'pvsg_conv3x3_f16x2_sliced': [_c_f] * 6 + [_i] * 8 + [_c_f, _c_f],
    'pvsg_mask_logits_f16x2': [_c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _ll, _c_f, _c_f],
    'pvsg_attn_mask_bits_f16x2': [_c_f, _c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _ll, _c_f, _c_f],
    'pvsg_attn_mask_bits_packed_f16x2': [_c_f, _c_f, _c_f, _c_f, _i, _i, _i, _i, _ll, _c_f, _c_f],
    'pvsg_stem7x7_pack': [_c_f, _c_f, _c_f],
    'pvsg_group_norm_affine': [_c_f] * 6 + [_i, _i, _i, _ll, _f, _c_f],
    'pvsg_conv1x1_stats_chunks': [_i, _i, _i],
    'pvsg_conv_slices': [_i] * 7,
    'pvsg_rle_segments': [_i],
    'pvsg_rle_count': [_c_f, _i, _i, _i, _c_f, _c_f],
    'pvsg_rle_positions': [_c_f, _i, _i, _i, _c_f, _c_f, _c_f],
    'pvsg_conv1x1_f16x2_stats': [_c_f] * 7 + [_i] * 7 + [_c_f, _c_f],
    'pvsg_conv3x3_stats_chunks': [_i, _i],
    'pvsg_bottleneck_next_weight_matrix': [_c_f, _c_f, _i, _i, _c_f],
    'pvsg_bottleneck_tail_f16x2': [_c_f] * 11 + [_i] * 6 + [_c_f, _c_f],
    'pvsg_conv3x3_f16x2_stats': [_c_f] * 6 + [_i] * 6 + [_c_f, _c_f],
    'pvsg_group_norm_finish': [_c_f, _i, _c_f, _c_f, _c_f, _c_f, _i, _i, _i, _ll, _f, _c_f],
    'pvsg_stem7x7_bn_relu_pool': [_c_f] * 5 + [_i, _i, _i, _c_f],
    'pvsg_stem7x7_f16x2_matrix': [_c_f, _c_f, _c_f],
    'pvsg_stem7x7_f16x2_bn_relu_pool': [_c_f] * 5 + [_i, _i, _i, _c_f, _c_f],
    'pvsg_conv3x3s2_affine': [_c_f] * 5 + [_i] * 6 + [_c_f],
}
# entry points that return a value instead of a status code
VALUE_RETURNING = ('pvsg_xattn_num_splits', 'pvsg_gemm_bf16x3_packed_elems', 'pvsg_gemm_f16x2_packed_elems',
                   'pvsg_minvis_chain_workspace_bytes', 'pvsg_reconsdot_workspace_bytes', 'pvsg_rle_counts_to_chars',
                   'pvsg_decoder_rows_post_workspace_bytes', 'pvsg_tube_index_table_words', 'pvsg_conv1x1_stats_chunks',
                   'pvsg_conv3x3_stats_chunks', 'pvsg_rel_tail_workspace_bytes', 'pvsg_conv_slices', 'pvsg_rle_segments',
                   'pvsg_rows_f16x2_packed_floats')

_lib = None


class BackendMissingError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendMissingError(
            'HIP backend library not built: %s (run `python -c "import __graft_entry__ as g; '
            'g.build()"` or `python -m openpvsg_amd.build`). There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for fn in ('pvsg_last_error', 'pvsg_version'):
        getattr(lib, fn).restype = ctypes.c_char_p
        getattr(lib, fn).argtypes = []
    lib.pvsg_abi_version.restype = _i
    lib.pvsg_abi_version.argtypes = []
    for name, argtypes in SIGNATURES.items():
        try:
            f = getattr(lib, name)
        except AttributeError as e:
            raise BackendMissingError('symbol %s missing from %s' % (name, LIB_PATH)) from e
        f.restype = _ll if name in ('pvsg_gemm_bf16x3_packed_elems', 'pvsg_gemm_f16x2_packed_elems', 'pvsg_minvis_chain_workspace_bytes',
                                   'pvsg_decoder_rows_post_workspace_bytes', 'pvsg_reconsdot_workspace_bytes',
                                   'pvsg_rle_counts_to_chars', 'pvsg_tube_index_table_words', 'pvsg_rel_tail_workspace_bytes',
                                   'pvsg_rows_f16x2_packed_floats') else _i
        f.argtypes = argtypes
    _lib = lib
    try:                                    # loud, once: a second tenant on the GPU without a CU partition (parallel.py)
        from . import parallel
        parallel.warn_if_gpu_shared()
    except ImportError:
        pass
    return lib


# ---- one stream at a time ------------------------------------------------------------------------------------------------
# Waves of the 16-bit-MFMA kernels (split GEMM / convolution / attention) that share a CU with waves of ANOTHER kernel were
# observed to corrupt that kernel's registers (scripts/coresidency_repro.hip) -- from another process
# (parallel.warn_if_gpu_shared / isolate_shared_gpu) and just as well from a second stream of this process.  Inside one stream
# kernels run back to back and nothing co-resides.  So launches of this backend may move from one stream to another (graph
# capture warm-ups do), but never run on two at once: a launch on stream S while the stream that carried the previous launches
# still has work pending is an error when either side is a 16-bit-MFMA kernel.  PVSG_MULTI_STREAM=allow switches the check off
# (measurements of the effect itself).
_MFMA16 = ('bf16x3', 'f16x2', 'masked_xattn')
_MFMA16_NAMES = frozenset(n for n in SIGNATURES if any(k in n for k in _MFMA16))
# Per DEVICE: [stream handle of the previous launch there, has that stream run a 16-bit-MFMA kernel since it took over].
# Kernels on different devices cannot co-reside, so a process that drives two GPUs (or two threads, one GPU each) alternates
# freely; two threads on ONE device with different streams is exactly the overlap the rule forbids, so the key is not the thread.
_cur = {}
_hip = None
_HIP_ERROR_NOT_READY = 600

try:
    _cur_dev = torch._C._cuda_getDevice
except AttributeError:                    # pragma: no cover
    _cur_dev = torch.cuda.current_device


def _stream_busy(handle):
    """work still pending on `handle` (a stream of the CURRENT device: ops.py enters the tensor's device before it calls).
    Only hipErrorNotReady means busy: an invalid / destroyed handle (an ExternalStream that is gone) has nothing in flight."""
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL('libamdhip64.so')          # already mapped by torch: same runtime instance
        _hip.hipStreamQuery.argtypes = [ctypes.c_void_p]
        _hip.hipStreamQuery.restype = ctypes.c_int
        _hip.hipGetLastError.argtypes = []
        _hip.hipGetLastError.restype = ctypes.c_int
    rc = _hip.hipStreamQuery(ctypes.c_void_p(handle))
    if rc not in (0, _HIP_ERROR_NOT_READY):
        _hip.hipGetLastError()                        # clear the sticky error of a dead handle: it is not this launch's
    return rc == _HIP_ERROR_NOT_READY


class ConcurrentStreamError(RuntimeError):
    pass


def _stream_changed(name, stream, ent):
    """slow path of `call`: the launch goes to another stream of its device than the previous one"""
    if torch.cuda.is_current_stream_capturing():       # captured launches do not execute; the replay runs on ONE stream
        return
    mfma16 = name in _MFMA16_NAMES
    prev = ent[0]
    if prev is not None and (mfma16 or ent[1]) and _stream_busy(prev):
        raise ConcurrentStreamError(
            '%s launched on HIP stream %#x while stream %#x of the same device still runs kernels of this backend: '
            '16-bit-MFMA kernels must not share the GPU with other kernels (co-residency corruption, DESIGN.md section 3).  '
            'Keep the backend on one stream per device, or order the streams (wait_stream + synchronize) before switching; '
            'PVSG_MULTI_STREAM=allow disables this check.' % (name, stream, prev))
    ent[0], ent[1] = stream, mfma16


def note_replay(stream=None):
    """A hipGraph of this backend's launches was replayed on `stream` (default: torch's current one): the replayed kernels
    are 16-bit-MFMA work on that stream as far as the one-stream rule is concerned (detectors._graphed,
    pipeline._graphed_forward call this after graph.replay())."""
    if not _CHECK_STREAMS:
        return
    dev = _cur_dev()
    if stream is None:
        stream = torch._C._cuda_getCurrentRawStream(dev) if hasattr(torch._C, '_cuda_getCurrentRawStream') \
            else torch.cuda.current_stream().cuda_stream
    ent = _cur.get(dev)
    if ent is None:
        ent = _cur[dev] = [None, False]
    if ent[0] is not None and ent[0] != stream and _stream_busy(ent[0]):
        raise ConcurrentStreamError('hipGraph replayed on HIP stream %#x while stream %#x of the same device still runs '
                                    'kernels of this backend (one stream per device)' % (stream, ent[0]))
    ent[0], ent[1] = stream, True


_CHECK_STREAMS = os.environ.get('PVSG_MULTI_STREAM', 'deny') != 'allow'


def call(name, *args):
    """Invoke a C-ABI entry point; non-zero status -> RuntimeError with the library's message."""
    lib = load()
    if _CHECK_STREAMS:
        stream = args[-1] or 0
        ent = _cur.get(_cur_dev())
        if ent is None:
            ent = _cur[_cur_dev()] = [None, False]
        if stream != ent[0]:
            _stream_changed(name, stream, ent)
        elif not ent[1] and name in _MFMA16_NAMES:
            ent[1] = True
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.pvsg_last_error()
        raise RuntimeError('%s failed (code %d): %s' % (name, rc, msg.decode() if msg else '?'))
