"""ctypes binding of libghm.so (include/ghm.h).  There is no CPU fallback: if the library is
missing, or a call fails, this raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libghm.so")


class GhmError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """ghm_conv_desc"""
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("K", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("x_nstride", C.c_int64), ("y_nstride", C.c_int64)]


_p, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_D = C.POINTER(ConvDesc)

# name -> argtypes (return type is int unless listed in _SPECIAL)
SIGNATURES = {
    "ghm_options_reload": [],
    "ghm_device_count": [C.POINTER(_i32)],
    "ghm_ctx_create": [_i32, C.POINTER(_p)],
    "ghm_ctx_create_prio": [_i32, _i32, C.POINTER(_p)],
    "ghm_ctx_destroy": [_p],
    "ghm_device_info": [_p, C.c_char_p, _i32, C.POINTER(_i32), C.POINTER(_i64)],
    "ghm_alloc": [_p, C.c_size_t, C.POINTER(_p)],
    "ghm_free": [_p, _p],
    "ghm_h2d": [_p, _p, _p, C.c_size_t],
    "ghm_d2h": [_p, _p, _p, C.c_size_t],
    "ghm_event_create": [_p, C.POINTER(_p)],
    "ghm_event_destroy": [_p],
    "ghm_event_record": [_p, _p],
    "ghm_event_wait": [_p, _p],
    "ghm_event_sync": [_p],
    "ghm_queue_interference": [_p, _p, _p, _i32, C.POINTER(_f)],
    "ghm_host_alloc": [C.c_size_t, C.POINTER(_p)],
    "ghm_host_free": [_p],
    "ghm_h2d_async": [_p, _p, _p, C.c_size_t],
    "ghm_d2d": [_p, _p, _p, C.c_size_t],
    "ghm_memset_zero": [_p, _p, C.c_size_t],
    "ghm_sync": [_p],
    "ghm_scratch_info": [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)],
    "ghm_stream_wait": [_p, _p],
    "ghm_capture_begin": [_p],
    "ghm_capture_end": [_p, C.POINTER(_p)],
    "ghm_graph_launch": [_p, _p],
    "ghm_graph_destroy": [_p],
    "ghm_step_build": [_i32, _p, _p, _p],
    "ghm_step_record_begin": [_i32, _p, C.POINTER(_p)],
    "ghm_step_record_end": [_p],
    "ghm_step_timer_stride": [_p, _i32],
    "ghm_step_run": [_p],
    "ghm_step_destroy": [_p],
    "ghm_timer_start": [_p, _i32],
    "ghm_timer_stop": [_p, _i32],
    "ghm_timer_elapsed_ms": [_p, _i32, C.POINTER(_f)],
    "ghm_conv2d_fwd": [_p, _D, _p, _p, _p, _p, _i32, _f, _i32],
    "ghm_conv2d_dgrad": [_p, _D, _p, _p, _p, _p, _i32, _f, _i32],
    "ghm_conv2d_transpose_weights": [_p, _D, _p, _p],
    "ghm_conv2d_dgrad_t": [_p, _D, _p, _p, _p, _p, _i32, _f, _i32],
    "ghm_conv2d_dgrad_dact": [_p, _D, _p, _p, _p, _p, _i64, _i32, _f, _i32],
    "ghm_conv2d_wgrad_workspace": [_D, C.POINTER(C.c_size_t)],
    "ghm_conv2d_wgrad": [_p, _D, _p, _p, _p, _p, _i32],
    "ghm_lp_weight_bytes": [_D, _i32, C.POINTER(C.c_size_t)],
    "ghm_lp_pack_weights": [_p, _D, _p, _p, _i32, _i32],
    "ghm_lp_pack_batched": [_p, _p, _i32, _i32, _i32],
    "ghm_conv2d_fwd_lp": [_p, _D, _p, _p, _p, _p, _i32, _f, _i32, _i32],
    "ghm_conv2d_dgrad_lp": [_p, _D, _p, _p, _p, _p, _i32, _f, _i32, _i32],
    "ghm_conv2d_wgrad_lp_workspace": [_D, C.POINTER(C.c_size_t)],
    "ghm_conv2d_wgrad_lp": [_p, _D, _p, _p, _p, _p, _i32, _i32],
    "ghm_conv2d_fwd_pool": [_p, _D, _p, _p, _p, _p, _p, _i32, _f, _i32],
    "ghm_q_pack": [_p, _p, _i64, _i32, _i32, _i32, _p, _i64, _i32],
    "ghm_q_unpack": [_p, _p, _i64, _i32, _i32, _i32, _p, _i64, _i32],
    "ghm_conv2d_fwd_lp_q": [_p, _D, _p, _i64, _p, _p, _p, _p, _i64, _i32, _f, _i32, _i32],
    "ghm_conv2d_dgrad_lp_q": [_p, _D, _p, _i64, _p, _p, _p, _p, _i64, _i32, _f, _i32, _i32],
    "ghm_conv2d_dgrad_dact_lp_q": [_p, _D, _p, _i64, _p, _p, _p, _i64, _p, _i64, _i32, _f, _i32],
    "ghm_lp_variant": [_D, _i32, _i32, C.c_char_p, _i32],
    "ghm_conv2d_bn_fwd": [_p, _D, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _f, _f, _i32, _f],
    "ghm_conv2d_bn_fwd_lp_q": [_p, _D, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _f, _f, _i32, _f, _i32],
    "ghm_bn_apply_q": [_p, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _f, _p, _i64, _i32],
    "ghm_bn_backward_q": [_p, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _f, _i32,
                          _p, _p, _i64, _i32],
    "ghm_bn_backward_x": [_p, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _f, _i32, _p],
    "ghm_upsample_bilinear2_fwd_q": [_p, _p, _i64, _p, _i32, _i32, _i32, _i32, _p, _i64, _i32],
    "ghm_bn_apply_hi": [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _f, _p, _i64, _i32],
    "ghm_bn_backward_hi": [_p, _p, _i64, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _f, _i32, _p, _p, _i32],
    "ghm_pp_to_hi_q": [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _p, _i64, _i32],
    "ghm_maxpool2_mask_bwd_q": [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _f, _p, _i32, _p, _i64, _i32],
    "ghm_conv2d_wgrad_lp_q": [_p, _D, _p, _i64, _p, _i64, _p, _p, _i32, _i32],
    "ghm_conv2d_fwd_pool_lp_q": [_p, _D, _p, _i64, _p, _p, _p, _p, _i64, _p, _i32, _f, _i32],
    "ghm_maxpool2_mask_bwd": [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _f],
    "ghm_thin_fwd_q_supported": [_D, _i32, _i32, _i32],
    "ghm_conv2d_fwd_thin_q": [_p, _D, _p, _p, _p, _p, _i32, _f, _p, _i64, _i32],
    "ghm_conv2d_fwd_pool_thin_q": [_p, _D, _p, _p, _p, _p, _p, _i32, _f, _p, _i64, _i32],
    "ghm_thin_pool_lp_served": [_D, _i32, _f, _i32],
    "ghm_split_weight_bytes": [_D, _i32, _p, _i32],
    "ghm_split_pack_weights": [_p, _D, _p, _p, _i32, _i32],
    "ghm_split_pack": [_p, _p, _i64, _i32, _i32, _i32, _p, _i64, _i64, _i32],
    "ghm_split_pack_batched": [_p, _p, _i32, _i32, _i32],
    "ghm_conv2d_dgrad_dact_split": [_p, _D, _p, _i64, _i64, _p, _p, _p, _i64, _p, _i64, _i32, _f, _i32],
    "ghm_conv2d_dgrad_dact_split_q": [_p, _D, _p, _i64, _i64, _p, _p, _p, _i64, _p, _i64, _i32, _f, _i32],
    "ghm_conv2d_wgrad_split_workspace": [_D, _p],
    "ghm_conv2d_wgrad_split": [_p, _D, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _i32, _i32],
    "ghm_conv2d_wgrad_pooled_split": [_p, _D, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _p, _p, _i32, _i32],
    "ghm_maxpool2_mask_bwd_compress_q": [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _f, _p, _i64, _p, _p, _i32],
    "ghm_conv2d_fwd_pool_split": [_p, _D, _p, _p, _i64, _i64, _p, _p, _p, _p, _i64, _p, _i32, _f, _i32],
    "ghm_conv2d_fwd_split": [_p, _D, _p, _p, _i64, _i64, _p, _p, _p, _p, _i64, _i32, _f, _i32, _i32],
    "ghm_conv2d_dgrad_split": [_p, _D, _p, _p, _i64, _i64, _p, _p, _p, _p, _i64, _i32, _f, _i32, _i32],
    "ghm_conv2d_pool_bwd_sparse_supported": [_D, _i32],
    "ghm_conv2d_pool_wgrad_sparse_workspace": [_D, C.POINTER(C.c_size_t)],
    "ghm_conv2d_pool_wgrad_sparse": [_p, _D, _p, _p, _p, _p, _p, _p, _i32, _f, _i32, _p],
    "ghm_conv2d_pool_dgrad_sparse": [_p, _D, _p, _p, _p, _p, _p, _i32, _f, _i32],
    "ghm_maxpool2_mask_bwd_bias": [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _f, _p, _i32],
    "ghm_channel_sum": [_p, _p, _i32, _i32, _i32, _i64, _p, _i32],
    "ghm_bn_stats": [_p, _p, _i32, _i32, _i32, _i64, _f, _p, _p, _p, _p, _f, _p],
    "ghm_bn_apply": [_p, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _f],
    "ghm_bn_forward": [_p, _p, _i64, _p, _i64, _i32, _i32, _i32, _f, _p, _p, _p, _p, _f, _p, _p, _i32, _f, _p],
    "ghm_instance_norm_fwd": [_p, _p, _i64, _p, _i64, _i32, _i32, _i32, _f, _p, _p, _p, _p, _i32, _f, _p, _i32],
    "ghm_instance_norm_bwd": [_p, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _f, _i32, _p, _i32],
    "ghm_bn_backward": [_p, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p,
                        _i32, _f, _i32, _p],
    "ghm_act_fwd": [_p, _p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _f],
    "ghm_act_bwd": [_p, _p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _f, _i32],
    "ghm_maxpool2_fwd": [_p, _p, _p, _i32, _i32, _i32, _i32],
    "ghm_maxpool2_bwd": [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _f],
    "ghm_avgpool_fwd": [_p, _p, _p, _i32, _i32, _i32, _i32, _i32],
    "ghm_avgpool_bwd": [_p, _p, _p, _i32, _i32, _i32, _i32, _i32],
    "ghm_dropout": [_p, _p, _i64, _p, _i64, _i32, _i32, _i32, _f, C.c_uint32, _p],
    "ghm_counter_tick": [_p, _p],
    "ghm_transpose_weights_batched": [_p, _p, _i32, _i32],
    "ghm_upconv_collapse_weights": [_p, _p, _p, _p, _p, _i32, _i32],
    "ghm_upconv_collapse_batched": [_p, _p, _i32, _i32],
    "ghm_upconv_expand_batched": [_p, _p, _i32, _i32, _i32],
    "ghm_blconv_frame_sizes": [_i32, _i32, _i32, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64)],
    "ghm_blconv_frame_fwd": [_p, _p, _i64, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _p],
    "ghm_blconv_frame_gather": [_p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _p],
    "ghm_blconv_frame_dgrad": [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32],
    "ghm_blconv_frame_wgrad": [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32],
    "ghm_blconv_fwd_split": [_p, _D, _p, _i64, _i64, _p, _p, _p, _i32],
    "ghm_blconv_dgrad_split": [_p, _D, _p, _i64, _i64, _p, _p, _i32, _i32],
    "ghm_blconv_wgrad_split": [_p, _D, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _i32, _i32],
    "ghm_upconv_expand_wgrad": [_p, _p, _p, _i32, _i32, _i32],
    "ghm_pp_to_hi": [_p, _p, _p, _i64, _i32, _i32, _i32, _i32],
    "ghm_hi_to_pp": [_p, _p, _i64, _p, _i32, _i32, _i32, _i32],
    "ghm_upsample_nearest2_fwd": [_p, _p, _i64, _p, _i32, _i32, _i32, _i32],
    "ghm_upsample_nearest2_bwd": [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32],
    "ghm_upsample_bilinear2_fwd": [_p, _p, _i64, _p, _i32, _i32, _i32, _i32],
    "ghm_upsample_bilinear2_bwd": [_p, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32],
    "ghm_copy_view": [_p, _p, _i64, _p, _i64, _i32, _i32, _i32, _i32],
    "ghm_scale_samples": [_p, _p, _i64, _i32, _i32, _i32, _p, _i64, _p, _i64],
    "ghm_axpby": [_p, _f, _p, _f, _p, _i64],
    "ghm_image_batch": [_p, _p, _i32, _i32, _i32, _i32, _p, _i32, _p, _i64],
    "ghm_lsgan_loss": [_p, _p, _i64, _f, _p, _p, _f, _i32],
    "ghm_bce_loss": [_p, _p, _i64, _f, _p, _p, _f, _i32],
    "ghm_recon_loss": [_p, _p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _i64, _f, _i32],
    "ghm_rmsprop": [_p, _p, _p, _p, _i64, _p, _f, _f, _f],
    "ghm_adam": [_p, _p, _p, _p, _p, _i64, _p, _f, _f, _f, _f],
    "ghm_adam_tick": [_p, _p],
    "ghm_set_loss_scale_state": [_p, _p],
    "ghm_grad_check": [_p, _p, _i64],
    "ghm_loss_scale_update": [_p, _i32, _f, _f],
    "ghm_comm_unique_id": [C.POINTER(C.c_uint8 * 128)],
    "ghm_comm_init": [_p, _i32, _i32, C.POINTER(C.c_uint8 * 128)],
    "ghm_comm_destroy": [_p],
    "ghm_comm_count": [_p, C.POINTER(_i32)],
    "ghm_allreduce_sum": [_p, _p, _i64],
    "ghm_allreduce_sum_bf16": [_p, _p, _i64, _p],
    "ghm_allreduce_max": [_p, _p, _i64],
    "ghm_reduce_scatter_sum": [_p, _p, _i64],
    "ghm_all_gather": [_p, _p, _i64],
    "ghm_conv2d_variant": [_D, _i32, C.c_char_p, _i32],
}
_SPECIAL = {"ghm_last_error": ([], C.c_char_p), "ghm_bn_workspace": ([_i32], C.c_size_t),
            "ghm_dgrad_t_supported": ([_D], C.c_int), "ghm_lp_supported": ([_D, _i32, _i32], C.c_int),
            "ghm_conv2d_pool_supported": ([_D, _i32, _i32], C.c_int),
            "ghm_dgrad_dact_supported": ([_D, _i32], C.c_int),
            "ghm_blconv_supported": ([_i32, _i32, _i32, _i32, _i32], C.c_int),
            "ghm_blconv_split_supported": ([_D, _i32], C.c_int),
            "ghm_lp_q_direct": ([_D, _i32, _i32], C.c_int),
            "ghm_conv_bn_fused_supported": ([_D, _i32], C.c_int),
            "ghm_conv_bn_fused_supported_f32": ([_D], C.c_int),
            "ghm_lp_wgrad_q_supported": ([_D, _i32], C.c_int),
            "ghm_split_supported": ([_D, _i32], C.c_int),
            "ghm_conv2d_wgrad_pooled_split_supported": ([_D], C.c_int),
            "ghm_split_pool_supported": ([_D, _i32], C.c_int),
            "ghm_split_q_direct": ([_D, _i32], C.c_int),
            "ghm_split_dgrad_dact_supported": ([_D], C.c_int)}
# (ghm_conv_bn_fused_supported returns its answer as the int return value: typed with the plain signatures)

_lib = None


def load():
    """dlopen libghm.so and type every export of include/ghm.h.  Raises GhmError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GhmError("%s not found: build it with `python gan_heightmaps_amd/csrc/build.py` "
                       "(there is no CPU fallback)" % LIB_PATH)
    # the host driver only supports dmabuf IPC: RCCL's cross-process buffer registration needs this before the HIP
    # runtime initialises (no effect on a single process)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one hardware queue per HIP stream (stage A, stage B, gradients, communication, the null stream): with ROCm's
    # default of four, which streams end up sharing a queue -- and serialising -- depends on their creation order
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name, (args, res) in _SPECIAL.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ghm_last_error()
        raise GhmError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def call(name, *args):
    check(getattr(load(), name)(*args), name)


class tuning_env:
    """``with tuning_env(GHM_FORCE_TILE="big"): ...`` -- set GHM_* tuning switches for a block inside a running process.
    The library reads each switch once per call site (nothing calls getenv per launch), so both edges of the block
    tell it to re-read (ghm_options_reload)."""

    def __init__(self, **env):
        self.env = {k: str(v) for k, v in env.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)
        call("ghm_options_reload")
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        call("ghm_options_reload")
        return False


def all_export_names():
    return list(SIGNATURES) + list(_SPECIAL)
