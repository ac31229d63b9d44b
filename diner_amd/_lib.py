"""ctypes binding of libdiner_hip.so (the C ABI declared in include/diner_hip.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DINER_AMD_LIB") or os.path.join(_HERE, "libdiner_hip.so")   # override: timing experiments


class DinerScene(C.Structure):
    _fields_ = [("latent_cl", C.c_void_p), ("latent_proj", C.c_void_p), ("depth", C.c_void_p), ("depth_std", C.c_void_p),
                ("normals", C.c_void_p), ("poses_host", C.c_void_p), ("focal_host", C.c_void_p), ("c_host", C.c_void_p),
                ("std_pad_scale", C.c_void_p),
                ("img_w", C.c_float), ("img_h", C.c_float), ("feature_padding", C.c_float),
                ("nv", C.c_int32), ("C", C.c_int32), ("Hf", C.c_int32), ("Wf", C.c_int32),
                ("Hs", C.c_int32), ("Ws", C.c_int32), ("proj_stamp", C.c_uint64), ("latent_proj_f16", C.c_void_p),
                ("proj_stamp_f16", C.c_uint64)]


class DinerMlpParams(C.Structure):
    _fields_ = [("d_in", C.c_int32), ("d_latent", C.c_int32), ("d_hidden", C.c_int32), ("d_out", C.c_int32),
                ("n_blocks", C.c_int32), ("combine_layer", C.c_int32),
                ("num_freqs", C.c_int32), ("include_input", C.c_int32), ("freq_factor", C.c_float),
                ("lin_in_w", C.c_void_p), ("lin_in_b", C.c_void_p),
                ("lin_out_w", C.c_void_p), ("lin_out_b", C.c_void_p),
                ("fc0_w", C.POINTER(C.c_void_p)), ("fc0_b", C.POINTER(C.c_void_p)),
                ("fc1_w", C.POINTER(C.c_void_p)), ("fc1_b", C.POINTER(C.c_void_p)),
                ("lin_z_w", C.POINTER(C.c_void_p)), ("lin_z_b", C.POINTER(C.c_void_p))]


# name -> (restype, argtypes); must list every symbol include/diner_hip.h declares
SIGNATURES = {
    "diner_abi_version": (C.c_int, []),
    "diner_last_error": (C.c_char_p, []),
    "diner_mlp_create": (C.c_int, [C.POINTER(DinerMlpParams), C.c_void_p, C.POINTER(C.c_void_p)]),
    "diner_mlp_destroy": (C.c_int, [C.c_void_p]),
    "diner_mlp_update": (C.c_int, [C.c_void_p, C.POINTER(DinerMlpParams), C.c_int, C.c_void_p]),
    "diner_mlp_weights_fit_f16x3": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "diner_mlp_stamp": (C.c_uint64, [C.c_void_p]),
    "diner_mlp_fallback_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.c_int, C.c_void_p]),
    "diner_sample_depthguided_f32": (C.c_int, [C.POINTER(DinerScene), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_uint64, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_fill_uniform_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_longlong,
                                         C.c_void_p, C.c_void_p]),
    "diner_scene_proj_bytes": (C.c_size_t, [C.POINTER(DinerScene)]),
    "diner_scene_prepare_f32": (C.c_int, [C.POINTER(DinerScene), C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_scene_proj_f16_bytes": (C.c_size_t, [C.POINTER(DinerScene)]),
    "diner_scene_prepare_f16": (C.c_int, [C.POINTER(DinerScene), C.c_void_p, C.c_void_p]),
    "diner_field_workspace_bytes": (C.c_size_t, [C.c_longlong]),
    "diner_field_from_rays_f32": (C.c_int, [C.POINTER(DinerScene), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_field_from_points_f32": (C.c_int, [C.POINTER(DinerScene), C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_mlp_forward_workspace_bytes": (C.c_size_t, [C.c_longlong]),
    "diner_mlp_forward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_composite_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_render_f32": (C.c_int, [C.POINTER(DinerScene), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "diner_posenc_f32": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p,
                                   C.c_void_p]),
    "diner_profile_enable": (C.c_int, [C.c_int]),
    "diner_profile_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong),
                                        C.POINTER(C.c_longlong)]),
    "diner_index_f32": (C.c_int, [C.POINTER(DinerScene), C.c_int, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]),
    "diner_gemm_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "diner_linear512_pack_bytes": (C.c_size_t, []),
    "diner_linear512_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_channels_last_to_nchw_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    "diner_wgrad512_scratch_bytes": (C.c_size_t, []),
    "diner_wgrad512_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p]),
    "diner_train_inputs_f32": (C.c_int, [C.POINTER(DinerScene), C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_scatter_latent_grad_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]),
    "diner_view_mean_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_int, C.c_void_p]),
    "diner_colsum_f32": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "diner_field_act_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    "diner_composite_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_field_train_workspace_bytes": (C.c_size_t, [C.c_longlong, C.c_int]),
    "diner_field_train_ws_layout": (C.c_int, [C.c_longlong, C.c_int, C.POINTER(C.c_longlong), C.c_int]),
    "diner_field_train_forward_f32": (C.c_int, [C.POINTER(DinerScene), C.POINTER(DinerMlpParams), C.c_void_p, C.c_void_p,
                                                C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_field_train_backward_f32": (C.c_int, [C.POINTER(DinerScene), C.POINTER(DinerMlpParams),
                                                 C.POINTER(DinerMlpParams), C.c_longlong, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]),
    "diner_mlp_generic_workspace_bytes": (C.c_size_t, [C.POINTER(DinerMlpParams), C.c_int, C.c_longlong]),
    "diner_mlp_generic_forward_f32": (C.c_int, [C.POINTER(DinerMlpParams), C.c_float, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p,
                                                C.c_void_p, C.c_void_p]),
    "diner_mlp_generic_train_workspace_bytes": (C.c_size_t, [C.POINTER(DinerMlpParams), C.c_int, C.c_longlong]),
    "diner_mlp_generic_train_forward_f32": (C.c_int, [C.POINTER(DinerMlpParams), C.c_float, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p,
                                                      C.c_void_p, C.c_void_p]),
    "diner_mlp_generic_backward_f32": (C.c_int, [C.POINTER(DinerMlpParams), C.POINTER(DinerMlpParams), C.c_float, C.c_void_p, C.c_int, C.c_longlong,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_field_inputs_generic_bwd_f32": (C.c_int, [C.POINTER(DinerScene), C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p,
                                                     C.c_void_p]),
    "diner_field_inputs_generic_f32": (C.c_int, [C.POINTER(DinerScene), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.c_longlong, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "diner_field_train_forward_fused_f32": (C.c_int, [C.POINTER(DinerScene), C.c_void_p, C.POINTER(DinerMlpParams), C.c_void_p, C.c_void_p,
                                                      C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_field_train_workspace_split": (C.c_int, [C.c_longlong, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "diner_field_train_forward_s_f32": (C.c_int, [C.POINTER(DinerScene), C.POINTER(DinerMlpParams), C.c_void_p, C.c_void_p,
                                                  C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_field_train_backward_s_f32": (C.c_int, [C.POINTER(DinerScene), C.POINTER(DinerMlpParams), C.POINTER(DinerMlpParams),
                                                   C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_field_train_fused_overflowed": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "diner_field_train_batch_workspace_split": (C.c_int, [C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "diner_field_train_forward_batch_f32": (C.c_int, [C.POINTER(C.POINTER(DinerScene)), C.c_int, C.c_void_p, C.POINTER(DinerMlpParams), C.c_void_p,
                                                      C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "diner_field_train_backward_batch_f32": (C.c_int, [C.POINTER(C.POINTER(DinerScene)), C.c_int, C.POINTER(DinerMlpParams), C.POINTER(DinerMlpParams),
                                                       C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    "diner_quantize_rgb_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "diner_minmax_f32": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]),
    "diner_colormap_u8": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "diner_depth2normal_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "diner_gen_rays_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p]),
}

ABI_VERSION = 6          # DINER_ABI_VERSION of include/diner_hip.h
_lib = None


def load():
    """Load the shared library (once).  Raises ImportError -- never falls back -- when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: the HIP extension is not built. "
                          f"Run `python -m diner_amd.build` (needs hipcc). There is no CPU fallback.")
    # libdiner_hip.so links against libamdhip64.so.7.  PyTorch bundles its own copy of the HIP runtime; if this library were
    # loaded first the dynamic linker would bring in /opt/rocm's copy as a SECOND runtime next to PyTorch's, and the second
    # one finds no device (measured: tools/diag_loadorder.py).  Importing torch first makes both share PyTorch's runtime --
    # the one that owns the memory and streams the C ABI is handed.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.diner_abi_version() != ABI_VERSION:
        raise ImportError(f"libdiner_hip.so ABI version {lib.diner_abi_version()} != {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


MLP_UPDATE_TRAIN_ONLY = 1      # DINER_MLP_UPDATE_TRAIN_ONLY
E_INVALID, E_UNSUPPORTED, E_HIP = -1, -2, -3            # include/diner_hip.h:31-33


def check(rc):
    if rc != 0:
        msg = load().diner_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libdiner_hip: {msg} (code {rc})")
