"""ctypes binding of libfastnerf.so (the C ABI declared in include/fastnerf.h).

The product path has NO CPU fallback: if the library is missing or a call
fails, a RuntimeError is raised."""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FASTNERF_LIB') or os.path.join(HERE, 'libfastnerf.so')   # FASTNERF_LIB: tuning builds

P = C.c_void_p
I = C.c_int
L = C.c_int64
F = C.c_float
D = C.c_double
U64 = C.c_uint64



class StepArgs(C.Structure):
    """fn_step_args of include/fastnerf.h, field for field."""
    _fields_ = (
        [('n', L)] + [(k, P) for k in ('rays_o', 'rays_d', 'target', 't_rand', 'u', 'noise0', 'noise1')]
        + [('seed0', U64), ('seed1', U64), ('leaf_tag', P), ('table', P), ('net_floats', L)]
        + [(k, P) for k in ('params', 'grads', 'adam_m', 'adam_v', 'packed_fwd_c', 'packed_bwd_c', 'packed_fwd_f', 'packed_bwd_f',
                            'rays11', 'z0', 'raw0', 'act0', 'rgb0', 'disp0', 'acc0', 'w0', 'depth0',
                            'z1', 'z_samples', 'z_std', 'raw1', 'act1', 'rgb1', 'disp1', 'acc1', 'w1', 'depth1',
                            'g_rgb', 'g_rgb0', 'loss2', 'draw_ws', 'act_ws', 'dact_ws', 'partial_ws', 'live_ws', 'counts')]
        + [(k, D) for k in ('focal', 'lr', 'beta1', 'beta2', 'eps')]
        + [(k, F) for k in ('near_plane', 'far_plane', 'grad_scale')]
        + [(k, I) for k in ('math_mode', 'N_samples', 'N_importance', 'lindisp', 'perturb', 'white_bkgd', 'ndc', 'H', 'W', 'live',
                            'fwd_flags', 'max_leaves', 'adam_t')])


STEP_FORWARD, STEP_BWD_FINE, STEP_BWD_COARSE, STEP_UPDATE = 1, 2, 4, 8

# name -> (restype, argtypes); mirrors include/fastnerf.h one to one
SIGNATURES = {
    'fastnerf_version': (I, []),
    'fastnerf_last_error': (C.c_char_p, []),
    'fastnerf_device_cus': (I, []),
    'fastnerf_gen_rays': (I, [I, I, F, F, F, F, P, P, P, P]),
    'fastnerf_gen_rays_pixels': (I, [L, P, P, F, F, F, F, P, P, P]),
    'fastnerf_ndc_rays': (I, [L, I, I, D, F, P, P, P, P, P]),
    'fastnerf_pack_rays': (I, [L, P, P, F, F, I, I, I, D, P, P]),
    'fastnerf_sample_coarse': (I, [L, I, P, I, I, P, U64, P, P]),
    'fastnerf_posenc': (I, [L, I, P, P, P]),
    'fastnerf_mlp_pack': (I, [P, P, P, P]),
    'fastnerf_mlp_fwd': (I, [L, I, P, P, P, P, P, P, P]),
    'fastnerf_mlp_bwd_partial_floats': (L, []),
    'fastnerf_mlp_bwd': (I, [L, I, P, P, P, P, P, P, P, P]),
    'fastnerf_raw2outputs_fwd': (I, [L, I, P, P, P, P, I, P, P, P, P, P, P]),
    'fastnerf_raw2outputs_bwd': (I, [L, I, P, P, P, P, I, P, P, P]),
    'fastnerf_sample_pdf_merge': (I, [L, I, I, P, P, I, P, U64, P, P, P, P]),
    'fastnerf_sample_pdf': (I, [L, I, I, P, P, I, P, U64, P, P]),
    'fastnerf_mse_leafmax': (I, [L, P, P, P, F, P, P, P, P, I, P, P]),
    'fastnerf_adam_step': (I, [L, P, P, P, P, D, D, D, D, I, P]),
    'fastnerf_net_floats': (L, [I, I]),
    'fastnerf_mlp_act_floats': (L, [I, L]),
    'fastnerf_mlp_pack_ex': (I, [I, P, P, P, P]),
    'fastnerf_mlp_fwd_ex': (I, [I, L, I, P, P, P, P, P, P, P]),
    'fastnerf_mlp_bwd_ex': (I, [I, L, I, P, P, P, P, P, P, P, P]),
    'fastnerf_mlp_bf16_floats': (L, [I, I, L]),
    'fastnerf_mlp_bf16_partial_floats': (L, []),
    'fastnerf_mlp_bf16_pack': (I, [I, P, P, P, P]),
    'fastnerf_mlp_bf16_fwd': (I, [I, L, I, P, P, P, P, P, P, P]),
    'fastnerf_mlp_bf16_bwd': (I, [I, L, I, P, P, P, P, P, P, P, P]),
    'fastnerf_render_rays_fwd': (I, [I, L, I, I, P, I, I, I, I, P, P, P, P, U64, U64, P, P, P, P] + [P] * 18 + [P]),
    'fastnerf_render_rays_fwd_ex': (I, [I, L, I, I, P, I, I, I, I, P, P, P, P, U64, U64, P, P, P, P] + [P] * 18 + [I, P]),
    'fastnerf_mlp_bf16_fwd_flags': (I, [I, L, I, P, P, P, P, P, I, P]),
    'fastnerf_mlp_fwd_flags_ex': (I, [I, L, I, P, P, P, P, P, I, P]),
    'fastnerf_render_rays_bwd': (I, [I, L, I, I, P, I] + [P] * 20),
    'fastnerf_pp_intersect_sphere': (I, [L, P, P, P, P]),
    'fastnerf_pp_fg_depths': (I, [L, I, F, P, I, P, U64, P, P]),
    'fastnerf_pp_sample_pdf_merge': (I, [L, I, I, P, P, I, P, U64, P, P, P]),
    'fastnerf_pp_perturb_samples': (I, [L, I, P, P, U64, P, P]),
    'fastnerf_pp_sample_pdf': (I, [L, I, I, P, P, I, P, U64, P, P]),
    'fastnerf_pp_depth2pts_outside': (I, [L, I, P, P, P, P, P, P]),
    'fastnerf_pp_composite_fwd': (I, [L, I, I, P, P, P, P, P, P, P, P, P]),
    'fastnerf_pp_composite_bwd': (I, [L, I, I, P, P, P, P, P, P, P, P]),
    'fastnerf_pp_gen_rays': (I, [I, I, P, P, P, P, P]),
    'fastnerf_leaf_sumcount': (I, [L, P, P, P, I, P, P, P]),
    'fastnerf_tree_adjust_mean': (L, [P, P, P, I, D]),
    'fastnerf_tree_create': (P, [I, I, I, I]),
    'fastnerf_tree_destroy': (None, [P]),
    'fastnerf_tree_num_leaves': (I, [P, I]),
    'fastnerf_tree_max_leaves': (I, [P]),
    'fastnerf_tree_min_area': (D, [P, I]),
    'fastnerf_tree_get_leaves': (I, [P, I, P]),
    'fastnerf_tree_set_leaves': (I, [P, I, I, P, D]),
    'fastnerf_tree_leaf_plan': (I, [P, I, D, I, P]),
    'fastnerf_tree_adjust': (L, [P, P, I, D]),
    'fastnerf_tree_epoch_plan': (L, [P, D, I, P, P]),
    'fastnerf_epoch_rays': (I, [L, I, P, P, P, P, I, I, I, F, F, F, F, U64, I] + [P] * 10 + [P]),
    'fastnerf_epoch_rays_shard': (I, [L, I, P, P, P, P, I, I, I, F, F, F, F, U64, I] + [P] * 5 + [L, I, I] + [P] * 5 + [P]),
    'fastnerf_epoch_shard_rows': (L, [L, L, I, I]),
    'fastnerf_gauss_noise': (I, [L, F, U64, P, P]),
    'fastnerf_compact_ws_ints': (L, [L]),
    'fastnerf_compact_live': (I, [L, P, P, P, P, P]),
    'fastnerf_mlp_bf16_fwd_live': (I, [I, L, I, P, P, P, P, P, P, P, P]),
    'fastnerf_mlp_bf16_bwd_live': (I, [I, L, I, P, P, P, P, P, P, P, P, P, P]),
    'fastnerf_mlp_fwd_live_ex': (I, [I, L, I, P, P, P, P, P, P, P, P]),
    'fastnerf_mlp_bwd_live_ex': (I, [I, L, I, P, P, P, P, P, P, P, P, P, P]),
    'fastnerf_render_rays_bwd_live': (I, [I, L, I, I, P, I] + [P] * 22 + [P]),
    'fastnerf_mlp_x6_packed_floats': (L, [I, I]),
    'fastnerf_mlp_x6_pack': (I, [I, P, P, P, P]),
    'fastnerf_mlp_x6_fwd': (I, [I, L, I, P, P, P, P, P, P, I, P]),
    'fastnerf_mlp_x6_bwd': (I, [I, L, I, P, P, P, P, P, P, P, P]),
    'fastnerf_mlp_x6_fwd_live': (I, [I, L, I, P, P, P, P, P, P, P, P]),
    'fastnerf_mlp_x6_bwd_live': (I, [I, L, I, P, P, P, P, P, P, P, P, P, P]),
    'fastnerf_step_args_size': (L, []),
    'fastnerf_train_step': (I, [C.POINTER(StepArgs), I, P]),
    'fastnerf_comm_unique_id': (I, [C.c_char_p]),
    'fastnerf_comm_init': (I, [C.POINTER(P), C.c_char_p, I, I]),
    'fastnerf_comm_destroy': (I, [P]),
    'fastnerf_allreduce_grads': (I, [P, P, L, C.c_float, P]),
    'fastnerf_allreduce_leaf_table': (I, [P, P, L, P]),
    'fastnerf_allreduce_leaf_sumcount': (I, [P, P, P, L, P]),
    'fastnerf_leaf_table_reset': (I, [P, L, P]),
    'fastnerf_leaf_table_read': (I, [P, P, L, P]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes library; raise loudly when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(the HIP path has no CPU fallback)')
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if l.fastnerf_step_args_size() != C.sizeof(StepArgs):
            raise RuntimeError('fn_step_args: the ctypes mirror does not match the library (stale libfastnerf.so?)')
        _lib = l
    return _lib


def check(rc, what):
    if rc is None or rc < 0:
        msg = lib().fastnerf_last_error()
        raise RuntimeError(f'{what} failed ({rc}): {msg.decode() if msg else "?"}')
    return rc


def ptr(t):
    """Device (or host) pointer of a contiguous fp32/int32 tensor, or NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), 'tensor must be contiguous'
    return t.data_ptr()


def stream():
    """The HIP stream torch is currently enqueueing on."""
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('fastnerf ops run on the GPU only (no CPU fallback): got a CPU tensor')
