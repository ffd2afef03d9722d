"""ctypes binding of libsocialways_hip.so (the C ABI of include/socialways_hip.h).

The library is the product's only compute path.  There is no CPU fallback: if the shared object
is missing or a kernel returns an error the call raises, loudly.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SW_LIB_PATH") or os.path.join(_HERE, "libsocialways_hip.so")   # SW_LIB_PATH: tuning builds (tools/build_variant.sh)

GRP_ENC, GRP_EMB, GRP_ATT, GRP_DEC, GRP_DISC = 0, 1, 2, 3, 4
WS_GSAVE, WS_GDELTA, WS_DSAVE, WS_DDELTA, WS_WGRAD, WS_PAIRS = 0, 1, 2, 3, 4, 5
AMAX = 64
RED_BLOCKS = 64

_vp, _i, _f, _ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong

# name -> (restype, argtypes); mirrors include/socialways_hip.h one to one
PROTOTYPES = {
    "sw_version": (_i, []),
    "sw_last_error": (ctypes.c_char_p, []),
    "sw_param_count": (_i, [_i, _i]),
    "sw_param_offset": (_i, [_i, _i, _i]),
    "sw_param_tensors": (_i, [_i]),
    "sw_workspace_floats": (ctypes.c_size_t, [_i, _i, _i, _i, _i, _ll]),
    "sw_traj_4d": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "sw_enc_lstm_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sw_enc_lstm_fwd_aux": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _ll, _vp]),
    "sw_enc_lstm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "sw_social_pool_fwd": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "sw_social_pool_fwd_aux": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _ll, _vp]),
    "sw_social_features": (_i, [_vp, _i, _vp, _vp]),
    "sw_embed_features": (_i, [_vp, _ll, _vp, _vp, _vp]),
    "sw_attention_pool_dense": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "sw_social_pool_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sw_dec_rollout_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp]),
    "sw_dec_rollout_fwd_aux": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "sw_dec_rollout_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "sw_dec_rollout_bwd_aux": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "sw_gen_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "sw_gen_wgrad_adam": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               ctypes.c_longlong, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp]),
    "sw_enc_lstm_bwd_aux": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "sw_dec_rollout_bwd_dfuse": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp,
                                      _vp]),
    "sw_rows_gemm": (_i, [_vp, _i, _vp, _i, _i, _vp, ctypes.c_longlong, _i, _i, _vp, _i, _i, _vp]),
    "sw_linear_wgrad": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "sw_embed_features_bwd": (_i, [_vp, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp]),
    "sw_attention_dense_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "sw_attention_dense_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "sw_enc_lstm_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sw_dec_fc_dz": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "sw_dec_fc_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sw_wgrad_batch_new": (_vp, []),
    "sw_wgrad_batch_free": (None, [_vp]),
    "sw_disc_fwd": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "sw_disc_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "sw_disc_bwd_gan": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _f, _f, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sw_disc_bwd_gan_adam": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _f, _f, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp]),
    "sw_disc_update_supported": (_i, [_vp, _i, _i, _i]),
    "sw_disc_update": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _f, _f, _vp, _vp, _vp, _vp,
                            _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp]),
    "sw_disc_dpred": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "sw_gan_loss": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sw_l2_grad": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "sw_variety_grad": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "sw_traj_dist": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "sw_stage_step": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sw_gen_image_floats": (_i, []),
    "sw_gen_images": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "sw_stage_step_img": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sw_stage_step_zdev": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sw_disc_image_floats": (_i, [_i]),
    "sw_disc_image_table": (_i, [_i, _vp]),
    "sw_disc_images": (_i, [_vp, _vp, _vp, _i, _vp]),
    "sw_ade_fde": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    "sw_lstm_point_fwd": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sw_lstm_point_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "sw_act_fwd": (_i, [_vp, _ll, _i, _vp, _vp]),
    "sw_act_bwd": (_i, [_vp, _vp, _ll, _i, _vp, _vp]),
    "sw_sqdiff": (_i, [_vp, _i, _vp, _i, _vp, _i, _ll, _i, _f, _vp, _vp, _i, _vp]),
    "sw_pair_features": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "sw_attn_pairs_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sw_attn_pairs_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "sw_adam_packed": (_i, [_vp, _vp, _vp, _vp, _ll, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                            ctypes.c_double, _i, _vp]),
    "sw_wide_gemm": (_i, [_vp, _ll, _i, _vp, _ll, _i, _vp, _vp, _i, _vp, _i, _ll, _i, _i, _vp, _i, _i, _vp]),
    "sw_wide_lstm_fwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "sw_wide_lstm_bwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "sw_wide_out_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "sw_wide_out_bwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "sw_wide_sum_steps": (_i, [_vp, _ll, _i, _i, _ll, _i, _vp, _i, _vp]),
    "sw_wide_wgrad": (_i, [_vp, _i, _vp, _vp]),
    "sw_wide_transpose": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "sw_wide_dec_loop_supported": (_i, [_i]),
    "sw_wide_dec_loop_fwd": (_i, [_vp] * 13 + [_i] + [_vp] * 9 + [_i, _i, _i, _i, _vp]),
    "sw_wide_dec_loop_bwd": (_i, [_vp] * 18 + [_i, _i, _i, _i, _vp]),
    "sw_wide_disc_heads_supported": (_i, [_i, _i, _i]),
    "sw_wide_disc_heads_fwd": (_i, [_vp, _vp]),
    "sw_wide_disc_heads_bwd": (_i, [_vp, _vp]),
    "sw_wide_opimage": (_i, [_vp, _vp, _i, _ll, _vp, _vp]),
    "sw_wide_lstm_seq_supported": (_i, [_i]),
    "sw_wide_lstm_seq_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "sw_wide_lstm_seq_bwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "sw_comm_bytes": (_ll, [_i, _ll]),
    "sw_comm_alloc": (_i, [_ll, _vp]),
    "sw_comm_free": (_i, [_vp]),
    "sw_comm_ipc_export": (_i, [_vp, _vp]),
    "sw_comm_ipc_import": (_i, [_vp, _vp]),
    "sw_comm_ipc_close": (_i, [_vp]),
    "sw_comm_status": (_i, [_vp, _vp]),
    "sw_allreduce_direct": (_i, [_vp, _i, _i, _ll, _vp, _ll, _vp]),
    "sw_allreduce_direct_adam": (_i, [_vp, _i, _i, _ll, _vp, _ll, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_double, _i, _vp]),
}

# the header's MEASUREMENT section (include/socialways_hip.h, behind the product surface): bench.py's per-kernel event pass
DEBUG_PROTOTYPES = {
    "sw_kernel_timing": (_i, [_i]),
    "sw_kernel_timing_read": (_i, [ctypes.c_char_p, _i]),
    "sw_debug_spin": (_i, [ctypes.c_double, _vp]),
}
_lib = None


class SocialWaysHipError(RuntimeError):
    pass


def load():
    """Load the HIP library (import torch first so that its libamdhip64 is the one runtime in the
    process).  Raises if the library has not been built: there is no other compute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SocialWaysHipError(
            "%s not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C socialways_amd/csrc`; socialways_amd has no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in list(PROTOTYPES.items()) + list(DEBUG_PROTOTYPES.items()):
        fn = getattr(lib, name)          # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a contiguous fp32/int tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "socialways_amd kernels need contiguous tensors"
    return t.data_ptr()


def ptr_strided(t):
    """Device pointer of a tensor whose strides the callee is told about (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


TIMING = None   # bench.py: {"names": {abi_call,...}, "events": [(name, start, end)]} -> HIP events around calls


def call(name, *args):
    lib = load()
    t = TIMING
    if t is not None and name in t["names"]:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                      # on the current stream = the stream the kernel is launched on
        rc = getattr(lib, name)(*args)
        e1.record()
        t["events"].append((name, e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = {-1: "bad argument", -2: "unsupported shape", -3: "HIP error: " + lib.sw_last_error().decode()}.get(rc, "?")
        raise SocialWaysHipError("%s failed (%d): %s" % (name, rc, msg))


def ptr_array(tensors):
    """const float* const* argument (host array of device pointers)."""
    arr = (ctypes.c_void_p * len(tensors))(*[ptr(t) for t in tensors])
    return ctypes.cast(arr, ctypes.c_void_p), arr


def workspace_floats(ws_id, B, To, Tp, nb=1, P=0):
    return int(load().sw_workspace_floats(ws_id, B, To, Tp, nb, P))


def require_gpu(t):
    if not t.is_cuda:
        raise SocialWaysHipError("socialways_amd runs on MI355X only: got a %s tensor (no CPU fallback)" % t.device)


def indexed_device(device):
    """torch.device with its index resolved ("cuda" -> cuda:<current>): tensors report indexed devices, and a trainer that
    compares `tensor.device == self.device` (is z already in HBM?) must not be told apart from them by a missing index."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None and torch.cuda.is_available():
        d = torch.device("cuda", torch.cuda.current_device())
    return d
