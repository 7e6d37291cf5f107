"""ctypes binding of libltu_hip.so (declared in include/ltu_hip.h).

The product path has no fallback: if the shared library is missing or a symbol cannot be
resolved, importing the ops raises immediately.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# LTU_LIB: another in-tree build of the same sources (e.g. `make EXPERIMENTS=1` output kept beside the product library for A/B runs)
LIB_PATH = os.path.join(_HERE, os.environ.get('LTU_LIB', 'libltu_hip.so'))

P, I, L, F, U = c_void_p, c_int, c_longlong, c_float, c_uint64


class ReduceJob(ctypes.Structure):
    """struct ltu_reduce_job (include/ltu_hip.h): a pending second-stage reduction"""
    _fields_ = [('part', c_void_p), ('nsplit', c_int), ('n', c_int), ('k', c_int), ('nseg', c_int),
                ('out', c_void_p * 3), ('outb', c_void_p * 3), ('mode', c_int)]


class WgradJob(ctypes.Structure):
    """struct ltu_wgrad_job (include/ltu_hip.h)"""
    _fields_ = [('grad', c_void_p), ('a', c_void_p), ('dw', c_void_p * 3), ('db', c_void_p * 3),
                ('ldg', c_int), ('lda', c_int), ('nw', c_int), ('M', c_int), ('N', c_int), ('K', c_int)]


WGRAD_GROUP_MAX = 32      # LTU_WGRAD_GROUP_MAX of include/ltu_hip.h


# name -> argument types (return type is always int).  Mirrors include/ltu_hip.h one to one.
SIGNATURES = {
    'ltu_version': [],
    'ltu_build_flags': [],
    'ltu_config_set': [ctypes.c_char_p, I, I],
    'ltu_selftest_group_reduce': [P, P, P, I, I, P],
    'ltu_window_embed': [P, P, I, I, I, I, I, P],
    'ltu_pack_conv_weight': [P, P, P, I, I, I, I, I, P],
    'ltu_unpack_conv_wgrad': [P, P, I, I, I, P],
    'ltu_transpose_f32': [P, P, I, I, I, I, I, P],
    'ltu_cast_f32': [P, P, L, I, P],
    'ltu_linear_fwd': [P, I, P, I, P, P, I, I, I, I, I, I, P],
    'ltu_wgrad_ws_floats': [L, I, I],
    'ltu_upconv_wgrad_ws_floats': [L, I, I, I],
    'ltu_linear_wgrad': [P, I, P, I, P, P, I, I, I, I, P, L, P, I, P],
    'ltu_layer_tail_fwd': [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, F, F, U, U, U, P, I, P, P, P, I, P, P, P, P, P, I, P],
    'ltu_layer_tail_blocks': [L],
    'ltu_layer_tail_bwd': [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, L, I, F, U, U, U, P, I, I, P],
    'ltu_reduce_batch': [P, I, P],
    'ltu_linear_wgrad_group_ws_floats': [P, I, I],
    'ltu_linear_wgrad_group': [P, I, I, P, L, I, P],
    'ltu_conv3d_fwd': [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, L, I, P],
    'ltu_conv3d_ws_floats': [I, I, I, I, I, I],
    'ltu_conv3d_pair_fwd': [P, P, P, P, P, I, I, I, I, I, I, I, P, L, I, P],
    'ltu_conv3d_pair_dgrad': [P, P, P, P, I, I, I, I, I, I, I, P, L, I, P],
    'ltu_conv3d_pair_wgrad': [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P, L, I, P],
    'ltu_conv3d_dgrad': [P, P, P, P, I, I, I, I, I, I, I, I, I, I, P, L, I, P],
    'ltu_conv3d_wgrad': [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P, L, I, P],
    'ltu_affine_sample': [P, P, P, I, I, I, I, P],
    'ltu_zoom_sample': [P, P, P, I, I, I, I, P],
    'ltu_adjust_contrast': [P, P, P, P, I, L, P],
    'ltu_linear_gelu_fwd': [P, I, P, P, P, P, I, I, I, F, U, P, I, P],
    'ltu_weight_prep': [P, I, I, P],
    'ltu_weight_prep_chunks': [P, P, I, I, P],
    'ltu_sumpool2': [P, P, I, I, I, I, I, I, P],
    'ltu_upconv_fwd': [P, P, P, P, I, I, I, I, I, I, I, P],
    'ltu_upconv_dgrad': [P, P, P, I, I, I, I, I, I, P, L, I, P],
    'ltu_igemm_ws_floats': [L, I, I],
    'ltu_upconv_wgrad': [P, P, P, P, P, I, I, P, L, I, I, I, I, I, I, I, I, P],
    'ltu_linattn_splits': [I, I],
    'ltu_linattn_ws_floats': [I, I, I],
    'ltu_linattn_fwd': [P, P, P, P, P, P, L, I, I, I, I, P],
    'ltu_linattn_ctx': [P, P, P, P, L, I, I, I, I, P],
    'ltu_linattn_bwd': [P, P, P, P, P, P, P, P, P, L, I, I, I, I, P],
    'ltu_window_gather': [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    'ltu_vote_accumulate': [P, P, P, P, I, I, I, I, I, I, I, I, P],
    'ltu_vote_finalize': [P, P, P, I, I, I, I, I, I, I, I, I, I, I, P],
    'ltu_keep_largest_component': [P, P, P, P, P, I, I, I, I, I, P],
    'ltu_seg_metrics': [P, P, P, P, I, I, I, I, L, F, P],
    'ltu_ct_preprocess': [P, P, P, P, I, I, I, F, F, F, F, P],
    'ltu_crop_flip': [P, P, P, I, I, I, I, I, I, I, I, P],
    'ltu_adamw': [P, P, P, P, L, F, F, F, F, F, L, F, P],
    'ltu_norm_ws_floats': [],
    'ltu_instnorm_stats': [P, P, P, L, I, L, I, I, P],
    'ltu_instnorm_apply': [P, P, P, P, I, L, I, I, F, F, U, P, I, P],
    'ltu_instnorm_fwd': [P, P, P, L, P, P, I, L, I, I, F, F, U, P, I, P],
    'ltu_instnorm_bwd': [P, P, P, P, P, P, P, L, P, I, L, I, I, F, F, U, P, I, P],
    'ltu_layernorm_fwd': [P, P, P, P, P, P, L, I, F, F, U, P, I, P],
    'ltu_layernorm_bwd': [P, P, P, P, P, P, P, P, P, P, L, P, L, I, F, U, P, I, P],
    'ltu_gelu_dropout_fwd': [P, P, L, F, U, P, I, P],
    'ltu_gelu_dropout_bwd': [P, P, P, L, F, U, P, I, P],
    'ltu_head_softmax_fwd': [P, P, L, I, I, I, P],
    'ltu_head_softmax_bwd': [P, P, P, L, I, I, I, P],
    'ltu_final_softmax_fwd': [P, P, I, I, I, I, I, I, I, P],
    'ltu_final_softmax_bwd': [P, P, P, I, I, I, I, I, I, I, P],
    'ltu_onehot_argmax': [P, P, L, I, P],
    'ltu_gate_fwd': [P, P, P, P, P, P, P, P, P, I, L, I, I, P],
    'ltu_gate_bwd': [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, L, P, P, I, L, I, I, P],
    'ltu_dwconv_fwd': [P, P, P, P, I, I, I, I, I, F, U, P, I, P],
    'ltu_dwconv_bwd_ws_floats': [I, I, I, I, I, I],
    'ltu_dwconv_bwd': [P, P, P, P, P, P, P, P, L, I, I, I, I, I, F, U, P, I, P],
    'ltu_roi_plan_size': [I, I, I, I, P, P, P],
    'ltu_roi_plan': [P, I, I, I, I, I, I, F, P, P, P, P, P],
    'ltu_roi_resample': [P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    'ltu_trilinear_up': [P, P, P, I, I, I, I, I, I, I, I, P],
    'ltu_trilinear_adjoint_ws_elems': [I, I, I, I, I, I],
    'ltu_trilinear_adjoint': [P, P, P, P, L, I, I, I, I, I, I, I, P],
    'ltu_loss_ws_floats': [I, L, I],
    'ltu_loss_fwd': [P, P, P, L, P, P, I, L, I, F, F, P, P, P],
    'ltu_loss_bwd': [P, P, P, P, P, I, L, I, P],
    'ltu_label_maxpool': [P, P, I, I, I, I, I, P],
    'ltu_comm_load': [ctypes.c_char_p],
    'ltu_comm_unique_id': [P],
    'ltu_comm_init': [P, P, I, I],
    'ltu_comm_allreduce_avg': [P, P, L, P],
    'ltu_comm_broadcast': [P, P, L, I, P],
    'ltu_comm_destroy': [P],
}

# exported by an experiments build only (make -C lintransunet_amd/csrc EXPERIMENTS=1): bound when present
EXPERIMENT_SIGNATURES = {
    'ltu_selftest_last_arriver': [P, P, P, P, P, I, I, I, I, I, P],
}

_lib = None


class LtuError(RuntimeError):
    pass


def load():
    """Load the library once; raises if it is not built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LtuError(f'{LIB_PATH} is missing: the HIP extension has not been built; there is no fallback path')
    # PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so); libltu_hip.so must bind to THAT copy, because device
    # pointers and streams come from torch.  Importing torch first puts its runtime into the process before ours is resolved
    # (loaded the other way round, the system runtime under /opt/rocm answers our launches with hipErrorNoDevice).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = c_longlong if (name.endswith(('_ws_floats', '_ws_elems')) or name == 'ltu_layer_tail_blocks') else c_int
    for name, args in EXPERIMENT_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = args, c_int
    _lib = lib
    return lib


def experiments():
    """True when the library was built with EXPERIMENTS=1 (rejected kernel variants and their self-tests compiled in)"""
    return bool(load().ltu_build_flags() & 1)


_ERR = {-1: 'LTU_E_DTYPE', -2: 'LTU_E_SHAPE', -3: 'LTU_E_ALIGN', -4: 'LTU_E_ARG'}


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        what = _ERR.get(rc) or ('LTU_E_COMM (RCCL not loaded)' if rc == -100 else f'ncclResult_t {-100 - rc}' if rc < -100
                                else f'hipError_t {rc}')
        raise LtuError(f'{name} failed: {what}')


def config_set(name, value=None):
    """process-wide knob override (value=None removes it); see ltu_config_set in include/ltu_hip.h"""
    call('ltu_config_set', name.encode(), 0 if value is None else int(value), 1 if value is None else 0)
