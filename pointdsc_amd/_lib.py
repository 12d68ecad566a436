"""ctypes binding of include/pointdsc_hip.h.

There is NO CPU fallback: if libpointdsc_hip.so is missing or does not export every declared symbol the
import of the product path fails loudly (``PointDSCLibraryError``).  Loading the library needs only the HIP
runtime, so it also works (for symbol checks) on a box without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("POINTDSC_HIP_LIB", PKG / "libpointdsc_hip.so"))


class PointDSCLibraryError(RuntimeError):
    pass


class PdscConfig(C.Structure):
    """struct pdsc_config (include/pointdsc_hip.h)."""
    _fields_ = [
        ("in_dim", C.c_int), ("num_layers", C.c_int), ("num_channels", C.c_int),
        ("num_iterations", C.c_int), ("k", C.c_int), ("refine_iters", C.c_int),
        ("inlier_threshold", C.c_float), ("nms_radius", C.c_float), ("refine_threshold", C.c_float),
        ("attention_precision", C.c_int), ("compat_format", C.c_int), ("layer_gemm", C.c_int),
        ("att_leaves", C.c_int),
    ]


# enum pdsc_wsection
W_SECTIONS = [
    "LAYER0_W", "LAYER0_B", "PCN_W", "PCN_B", "QKV_W", "QKV_B", "FC1_W", "FC1_B", "FC2_W", "FC2_B",
    "FC3_W", "FC3_B", "CLS1_W", "CLS1_B", "CLS2_W", "CLS2_B", "CLS3_W", "CLS3_B", "SIGMA", "SIGMA_SPAT",
]
W = {name: i for i, name in enumerate(W_SECTIONS)}

_vp, _i, _f, _ll, _sz = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t
_cfgp = C.POINTER(PdscConfig)

# name -> (restype, argtypes); must list every function include/pointdsc_hip.h declares
SIGNATURES = {
    "pdsc_version": (_i, []),
    "pdsc_set_range_report": (_i, [_vp]),
    "pdsc_last_error": (C.c_char_p, []),
    "pdsc_experiments_enabled": (_i, []),
    "pdsc_wpack_floats": (_ll, [_cfgp]),
    "pdsc_wpack_offset": (_ll, [_cfgp, _i, _i]),
    "pdsc_compat_ld": (_ll, [_i]),
    "pdsc_workspace_bytes": (_sz, [_cfgp, _i, _i, _i]),
    "pdsc_workspace_offset": (_ll, [_cfgp, _i, _i, _i, C.c_char_p]),
    "pdsc_spatial_compat": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    "pdsc_spatial_compat_u16": (_i, [_vp, _vp, _vp, _vp, _ll, _i, _i, _vp]),
    "pdsc_selftest_exact_math": (_i, [_vp, _f, _vp, _vp, _ll, _vp]),
    "pdsc_selftest_mfma_valu_neighbour": (_i, [_vp, _i, _i, _vp]),
    "pdsc_classifier_hidden": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "pdsc_linear": (_i, [_vp, _ll, _vp, _vp, _vp, _ll, _vp, _ll, _i, _i, _i, _i, _vp]),
    "pdsc_layer0": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp]),
    "pdsc_layer_fused": (_i, [_vp] * 16 + [_i, _vp]),
    "pdsc_layer_fused_split": (_i, [_vp, _vp, _vp, _i, _i] + [_vp] * 18 + [_i, _i, _vp]),
    "pdsc_wsplit_bytes": (_sz, [_cfgp]),
    "pdsc_wsplit_offset": (_ll, [_cfgp, _i, _i]),
    "pdsc_wsplit_build": (_i, [_cfgp, _vp, _vp, _vp]),
    "pdsc_layer_fused_x3": (_i, [_vp, _vp, _vp, _i, _i] + [_vp] * 17 + [_i, _i, _vp]),
    "pdsc_layer_prefers_block": (_i, [_i, _i]),
    "pdsc_layer_h3_uses_coop": (_i, [_i, _i]),
    "pdsc_wfrag_tail_bytes": (_sz, []),
    "pdsc_wfrag_head_bytes": (_sz, []),
    "pdsc_wfrag_build_tail": (_i, [_vp] * 8),
    "pdsc_wfrag_build_head": (_i, [_vp] * 6),
    "pdsc_wfrag_build_tail_fmt": (_i, [_vp] * 7 + [_i, _vp]),
    "pdsc_wfrag_build_head_fmt": (_i, [_vp] * 5 + [_i, _vp]),
    "pdsc_layer_fused_frag_fmt": (_i, [_vp, _vp, _vp, _i, _i] + [_vp] * 9 + [_i, _i, _i, _vp]),
    "pdsc_layer_fused_frag_io": (_i, [_vp, _vp, _vp, _i, _i] + [_vp] * 8 + [_i, _i, _i, _i, _vp]),
    "pdsc_sc_attention_split_partials": (_i, [_vp, _vp, _vp, _i, _ll, _vp, _sz, _i, _i, _i, _i, _vp]),
    "pdsc_layer_fused_frag": (_i, [_vp, _vp, _vp, _i, _i] + [_vp] * 9 + [_i, _i, _vp]),
    "pdsc_split_q_bytes": (_sz, [_i, _i]),
    "pdsc_split_kv_bytes": (_sz, [_i, _i]),
    "pdsc_pack_qkv_split": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "pdsc_attention_split_scratch_bytes": (_sz, [_i, _i, _i]),
    "pdsc_attention_split_default_split": (_i, [_i, _i]),
    "pdsc_attention_leaf_count": (_i, [_i]),
    "pdsc_attention_leaf_plan": (_i, [_i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pdsc_attention_leaf_scratch_bytes": (_sz, [_i, _i, _i]),
    "pdsc_attention_trace": (_i, [_vp]),
    "pdsc_layer_trace": (_i, [_vp]),
    "pdsc_sc_attention_split": (_i, [_vp, _vp, _vp, _ll, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "pdsc_sc_attention_split_u16": (_i, [_vp, _vp, _vp, _ll, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "pdsc_attention_scratch_bytes": (_sz, [_i, _i, _i]),
    "pdsc_attention_default_split": (_i, [_i, _i]),
    "pdsc_sc_attention": (_i, [_vp, _vp, _ll, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "pdsc_normalize_confidence": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "pdsc_nms_keys": (_i, [_vp, _vp, _f, _vp, _i, _i, _vp]),
    "pdsc_nms_workspace_bytes": (_sz, [_i, _i]),
    "pdsc_nms_keys_grid": (_i, [_vp, _vp, _f, _vp, _vp, _sz, _i, _i, _vp]),
    "pdsc_rank_select": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "pdsc_knn_seeds": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pdsc_knn_seeds_form": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pdsc_normalize_confidence_pf": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "pdsc_seed_power_iteration": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pdsc_seed_solve": (_i, [_vp] * 11 + [_i, _i, _i, _i, _i, _vp]),
    "pdsc_seed_transforms": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pdsc_rigid_transform_3d": (_i, [_vp, _vp, _vp, _f, _vp, _i, _i, _vp]),
    "pdsc_score_hypotheses": (_i, [_vp, _vp, _vp, _f, _vp, _i, _i, _i, _vp]),
    "pdsc_select_best": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pdsc_post_refinement": (_i, [_vp, _vp, _vp, _f, _i, _vp, _vp, _i, _i, _vp]),
    "pdsc_profile_enable": (_i, [_i]),
    "pdsc_profile_reset": (_i, []),
    "pdsc_profile_set_stride": (_i, [_i, _i]),
    "pdsc_profile_read": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "pdsc_match_scratch_bytes": (_sz, [_i, _i]),
    "pdsc_match_descriptors": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "pdsc_match_descriptors_ip": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "pdsc_select_correspondences": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "pdsc_build_corr_pos": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pdsc_sm_workspace_bytes": (_sz, [_i, _i]),
    "pdsc_sm_baseline": (_i, [_vp, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp]),
    "pdsc_sm_baseline_form": (_i, [_vp, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    "pdsc_cal_confidence_workspace_bytes": (_sz, [_i, _i]),
    "pdsc_cal_confidence": (_i, [_vp, _ll, _vp, _i, _i, _vp, _vp, _sz, _i, _i, _vp]),
    "pdsc_eval_stats": (_i, [_vp, _vp, _vp, _vp, _f, _f, _vp, _i, _i, _vp]),
    "pdsc_encoder_range_probe": (_i, [_cfgp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "pdsc_forward_validation": (_i, [_cfgp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _ll, _vp, _sz, _vp]),
    "pdsc_feature_compat": (_i, [_vp, _vp, _vp, _ll, _i, _i, _vp]),
    "pdsc_conv_mask_all_pairs": (_i, [_vp, _i, _vp]),
    "pdsc_forward_testing": (_i, [_cfgp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "pdsc_forward_testing_ragged": (_i, [_cfgp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "pdsc_forward_testing_streams": (_i, [_cfgp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load (once) and type the shared library; raise PointDSCLibraryError if unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise PointDSCLibraryError(
            f"{LIB_PATH} not found: build it with `python -m pointdsc_amd.build` (needs hipcc, gfx950). "
            "pointdsc_amd has no CPU fallback.")
    try:
        lib = C.CDLL(str(LIB_PATH))
    except OSError as e:  # missing HIP runtime etc.
        raise PointDSCLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PointDSCLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.pdsc_version() != 8:
        raise PointDSCLibraryError(f"unexpected library version {lib.pdsc_version()}")
    _lib = lib
    return lib


def last_error() -> str:
    return load().pdsc_last_error().decode(errors="replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (status {rc}): {last_error()}")
