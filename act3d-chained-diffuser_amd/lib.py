"""ctypes binding of libact3d_hip.so (declared in include/act3d_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# development aid: A3D_LIB names another build of the same library inside the package directory (A/B builds made with
# A3D_HIPCC_FLAGS and A3D_LIB_OUT, e.g. libact3d_hip_occ3.so); the default library is the one build() makes
LIB_PATH = os.path.join(_PKG_DIR, os.path.basename(os.environ.get("A3D_LIB", "libact3d_hip.so")))

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_z = C.c_size_t
_u64 = C.c_ulonglong



def _struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": [(f, t) for f, t in fields]})


# parameter blocks of the fused denoise kernels (include/act3d_hip.h, same field order)
DnHeadParams = _struct("DnHeadParams", [(n, _p) for n in ("enc_w0", "enc_b0", "enc_w1", "enc_b1", "sem", "lang_kv")] +
                       [("S_lang", _i)] + [(n, _p) for n in ("q_w", "q_b", "out_w", "out_b", "ln_g", "ln_b")])
DnCrossParams = _struct("DnCrossParams", [(n, _p) for n in ("sem", "mod", "q_w", "q_b", "freq", "Kf", "Vt")])
DnRestParams = _struct("DnRestParams", [(n, _p) for n in (
    "c_out_w", "c_out_b", "c_ln_g", "c_ln_b", "sem", "s_mod", "s_in_w", "s_in_b", "s_out_w", "s_out_b", "s_ln_g", "s_ln_b", "freq",
    "kmask", "f_mod", "f_w1", "f_b1", "f_w2", "f_b2", "f_ln_g", "f_ln_b")] + [("F", _i)])
DnLayerParams = _struct("DnLayerParams", [("cross", DnCrossParams), ("rest", DnRestParams)])
DnTailParams = _struct("DnTailParams", [(n, _p) for n in (
    "pos_w0", "pos_b0", "pos_w1", "pos_b1", "rot_w0", "rot_b0", "rot_w1", "rot_b1", "noise", "cond_data", "cond_mask", "coef_pos",
    "coef_rot")])

# parameter / gradient blocks of the fused query-stream layer (a3d_qs_params / a3d_qs_grads)
QsParams = _struct("QsParams", [(n, _p) for n in ("wv", "bv", "wo", "bo", "g1", "b1", "w1", "c1", "w2", "c2", "g2", "b2")])
QsGrads = _struct("QsGrads", [(n, _p) for n in ("dwv", "dbv", "dwo", "dbo", "dg1", "db1", "dw1", "dc1", "dw2", "dc2", "dg2", "db2")])

# name -> (restype, argtypes); mirrors include/act3d_hip.h one to one
SIGNATURES = {
    "a3d_version": (_i, []),
    "a3d_last_error_string": (C.c_char_p, []),
    "a3d_linear_fwd": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "a3d_linear_wgrad": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _p]),
    "a3d_linear_wgrad_ws": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _p, _z, _p]),
    "a3d_linear_wgrad_ws_bytes": (_z, [_i, _i, _i, _i]),
    "a3d_add_layernorm_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "a3d_add_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "a3d_rope_split_qk": (_i, [_p, _i, _p, _p, _f, _p, _i, _i, _i, _i, _i, _p]),
    "a3d_split_vt": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "a3d_proj_rope_split": (_i, [_p, _i, _p, _i, _p, _i, _p, _f, _p, _i, _p, _p, _f, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "a3d_rope_split": (_i, [_p, _i, _p, _p, _f, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "a3d_rope_merge_bwd": (_i, [_p, _i, _p, _p, _f, _p, _i, _i, _i, _i, _i, _i, _p]),
    "a3d_attn_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "a3d_attn_fwd_ws_floats": (_z, [_i, _i, _i, _i]),
    "a3d_attn_bwd_bf16": (_i, [_p] * 15 + [_i] * 7 + [_p]),
    "a3d_rope_split16": (_i, [_p, _i, _p, _p, _f, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "a3d_proj_rope_split16": (_i, [_p, _i, _p, _i, _p, _i, _p, _f, _p, _p, _i, _p, _f, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "a3d_attn16_fwd": (_i, [_p] * 7 + [_i] * 7 + [_p, C.c_uint, _f, _p]),
    "a3d_attn16_fwd_rows": (_i, [_p] * 7 + [_i] * 7 + [_p, C.c_uint, _f, _i, _p]),
    "a3d_conv3x3_tile_count": (_z, [_z, _i, _i]),
    "a3d_conv3x3_dgrad_tiles_ws_ints": (_z, [_z, _i, _i]),
    "a3d_conv3x3_mark_tiles": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "a3d_conv3x3_dgrad_tiles": (_i, [_p, _p, _p, _p, _p, _z, _i, _i, _p]),
    "a3d_conv3x3_wgrad_tokens_ws_floats": (_z, []),
    "a3d_conv3x3_wgrad_tokens": (_i, [_p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "a3d_attn16_bwd_pack_bytes": (_z, [_i, _i, _i]),
    "a3d_dbg_dn_prof": (_i, [_p]),
    "a3d_attn8_operand_bytes": (_z, [_i, _i, _i]),
    "a3d_attn8_fwd": (_i, [_p] * 8 + [_i] * 7 + [_p]),
    "a3d_attn16_bwd": (_i, [_p] * 16 + [_i] * 7 + [_p, C.c_uint, _f, _p]),
    "a3d_pose_to_signal": (_i, [_p, _p, _p, _i, _i, _p]),
    "a3d_signal_to_pose": (_i, [_p, _p, _p, _i, _i, _p]),
    "a3d_traj_errors": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "a3d_keypose_errors": (_i, [_p, _p, _p, _p, _i, _p, _i, _i, _i, _p]),
    "a3d_sym_quat_loss": (_i, [_p, _p, _i, _f, _p, _p, _i, _p]),
    "a3d_sq_fwd_ws_floats": (_z, [_i, _i, _i, _i]),
    "a3d_sq_attn_fwd": (_i, [_p, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "a3d_sq_bwd_ws_floats": (_z, [_i, _i, _i, _i]),
    "a3d_sq_attn_bwd": (_i, [_p, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i,
                             _i, _p]),
    "a3d_sq_attn_bwd_acc": (_i, [_p, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i,
                                 _i, _i, _p]),
    "a3d_qs_pre_fwd": (_i, [_p, _p, _p, _p, _p, _f, _p, _i, _i, _i, _p]),
    "a3d_qs_pre_bwd": (_i, [_p, _i, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "a3d_qs_save_floats": (_z, [_i, _i]),
    "a3d_qs_post_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "a3d_qs_post_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "a3d_sq_wgrad_reduce": (_i, [_p, _i, _p, _i, _p, _i, _p]),
    "a3d_dn_head": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _p]),
    "a3d_dn_cross_ws_floats": (_z, [_i, _i, _i]),
    "a3d_dn_cross": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "a3d_dn_rest": (_i, [_p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "a3d_dn_tail": (_i, [_p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _p]),
    "a3d_dn_persist_splits": (_i, [_i, _i]),
    "a3d_dn_persist_sync_ints": (_z, [_i, _i, _i, _i]),
    "a3d_dn_persist_kvx_floats": (_z, [_i, _i, _i]),
    "a3d_dn_persist_prof": (_i, [_p, _i, _i, _i, _i, _p]),
    "a3d_dn_persist_xbuf_floats": (_z, [_i, _i]),
    "a3d_dn_persist": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p] + [_i] * 10 + [_p]),
    "a3d_rope_rows_f32": (_i, [_p, _i, _p, _p, _f, _p, _i, _i, _i, _i, _i, _p]),
    "a3d_dropout": (_i, [_p, _p, _z, _p, C.c_uint, _f, _p]),
    "a3d_dropout_mask": (_i, [_p, _z, _p, C.c_uint, C.c_uint, C.c_uint, _f, _p]),
    "a3d_linear_fwd_drop": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, C.c_uint, _f, _p]),
    "a3d_add_layernorm_bwd_drop": (_i, [_p] * 10 + [_i, _i, _p, C.c_uint, _f, _p]),
    "a3d_attn_fwd_dropout": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, C.c_uint, _f, _p]),
    "a3d_attn_bwd_bf16_dropout": (_i, [_p] * 15 + [_i] * 7 + [_p, C.c_uint, _f, _p]),
    "a3d_pcd_downsample": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "a3d_knn_topk_ws_bytes": (_z, [_i, _i]),
    "a3d_knn_topk": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "a3d_traj_nn_topk": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "a3d_build_context": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "a3d_build_context_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "a3d_build_context_bf16": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "a3d_colsum_rows_ws_floats": (_z, [_i, _i, _i]),
    "a3d_colsum_rows": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _p, _p]),
    "a3d_build_context_bwd_bf16": (_i, [_p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "a3d_mask_logits_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "a3d_mask_logits_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "a3d_argmax_gather": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "a3d_soft_ce_loss": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _f, _f, _f, _p]),
    "a3d_elem_loss": (_i, [_p, _p, _i, _i, _f, _p, _p, _p]),
    "a3d_scale_by_scalar": (_i, [_p, _p, _p, _z, _p]),
    "a3d_quat_sigmoid_fwd": (_i, [_p, _p, _p, _i, _p]),
    "a3d_quat_sigmoid_bwd": (_i, [_p, _p, _p, _p, _i, _p]),
    "a3d_ortho6d_sigmoid_fwd": (_i, [_p, _p, _p, _i, _p]),
    "a3d_ortho6d_sigmoid_bwd": (_i, [_p, _p, _p, _p, _i, _p]),
    "a3d_select_row_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "a3d_select_row_bwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "a3d_sample_ghost_points": (_i, [_p, _p, _p, _f, _p, _i, _i, _i, _i, _p]),
    "a3d_rng_advance": (_i, [_p, _u64, _p]),
    "a3d_philox4x32_10_host": (None, [_p, _p, _p]),
    "a3d_sincos_host": (None, [_p, _p, _p, _z]),
    "a3d_adamw_step": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _z, _z, _f, _f, _f, _f, _f, _f, _f, _p]),
    "a3d_dbg_mfma_bf16": (_i, [_p, _p, _p, _p]),
    "a3d_dbg_mfma_f32": (_i, [_p, _p, _p, _p]),
    "a3d_dbg_cvt_pk_bf16": (_i, [_p, _p, _i, _p]),
    "a3d_bn_nslab": (_i, [_z, _i]),
    "a3d_bn_grid_cap": (_i, [_i]),
    "a3d_conv1x1_streams": (_i, [_i, _i]),
    "a3d_conv1x1_nslab": (_i, [_z, _i, _i]),
    "a3d_conv1x1_bn_fwd": (_i, [_p, _p, _p, _p, _i, _p, _p, _z, _i, _i, _p]),
    "a3d_conv1x1_topdown_serves": (_i, [_i, _i]),
    "a3d_conv1x1_topdown_fwd": (_i, [_p, _p, _p, _i, _p, _p, _z, _i, _i, _i, _i, _p]),
    "a3d_conv3x3_serves": (_i, [_i, _i, _i, _i]),
    "a3d_conv1x1_deep_mode": (_i, [_i]),
    "a3d_stem_conv_nslab": (_i, [_z, _i, _i]),
    "a3d_stem_conv_bn_fwd": (_i, [_p, _p, _p, _p, _p, _p, _z, _i, _i, _p]),
    "a3d_conv3x3_nslab": (_i, [_z, _i, _i, _i, _i]),
    "a3d_conv3x3_bn_fwd": (_i, [_p, _p, _p, _p, _i, _p, _p, _z, _i, _i, _i, _i, _p]),
    "a3d_bn_stats": (_i, [_p, _p, _z, _i, _i, _p]),
    "a3d_bn_finalize": (_i, [_p, _i, _z, _i, _f, _f, _p, _p, _p, _p, _p, _p, _i, _p]),
    "a3d_bn_apply": (_i, [_p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _p]),
    "a3d_bn_apply_pool2": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "a3d_rgb_normalize_nhwc_bf16": (_i, [_p, _p, _p, _p, _z, _i, _i, _p]),
    "a3d_upsample2_add_fwd": (_i, [_p, _p, _p, _i, _p, _i, _i, _i, _i, _p]),
    "a3d_upsample2_add_bwd_ws_floats": (_z, [_i, _i, _i, _i]),
    "a3d_upsample2_add_bwd": (_i, [_p, _p, _p, _i, _p, _i, _i, _i, _i, _p]),
    # diffusion.hip
    "a3d_ddpm_add_noise": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "a3d_ddpm_step": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "a3d_adaln_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "a3d_adaln_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "a3d_sinusoidal_emb": (_i, [_p, _p, _i, _i, _p]),
    "a3d_silu_fwd": (_i, [_p, _p, _z, _p]),
    "a3d_silu_bwd": (_i, [_p, _p, _p, _z, _p]),
    "a3d_add_rows": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "a3d_add_rows_bwd": (_i, [_p, _p, _i, _i, _i, _p]),
    "a3d_resize_crop": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _f, _p]),
    "a3d_traj_update": (_i, [_p, _p, _p, _i, _i, _i, _p]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libact3d_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(needs hipcc); there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError("libact3d_hip.so does not export %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return list(SIGNATURES.keys())


def last_error():
    s = load().a3d_last_error_string()
    return s.decode() if s else ""


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("libact3d_hip %s failed with code %d: %s" % (what, rc, last_error()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), "libact3d_hip needs contiguous tensors"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("act3d_amd ops run on an MI355X device tensor; got a %s tensor "
                               "(there is no CPU fallback)" % t.device)


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed with code %d: %s" % (name, rc, last_error()))
