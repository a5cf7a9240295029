"""ctypes binding of libtdm_hip.so (include/tdm_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the library is
missing or a call fails, a RuntimeError is raised."""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TDM_HIP_LIB") or os.path.join(_HERE, "csrc", "libtdm_hip.so")   # TDM_HIP_LIB: A/B a second build

_lib = None
_lock = threading.Lock()

c_f = ctypes.c_void_p      # device pointers travel as void*
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_float = ctypes.c_float
c_u64 = ctypes.c_uint64

_SIGS = {
    "tdm_version": ([], c_int),
    "tdm_last_error": ([], ctypes.c_char_p),
    "tdm_q_sample_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_i64, c_f], c_int),
    "tdm_p_sample_update_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_int, c_f, c_i64, c_f], c_int),
    "tdm_p_sample_update_pert_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_int, c_f, c_i64, c_i64, c_f], c_int),
    "tdm_to_unit_u8_f32": ([c_f, c_f, c_f, c_i64, c_f], c_int),
    "tdm_unet_param_offsets": ([ctypes.POINTER(ctypes.c_int32)], c_int),
    "tdm_unet_workspace_floats": ([c_i64, c_int], c_i64),
    "tdm_unet_slab_floats": ([], c_i64),
    "tdm_unet_fwd_f32": ([c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_f], c_int),
    "tdm_unet_bwd_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_f], c_int),
    "tdm_unet_get_activation": ([c_f, c_i64, c_int, c_f, c_f], c_int),
    "tdm_unet_relu_mask_io": ([c_f, c_i64, c_int, c_int, c_f, c_int, c_f], c_int),
    "tdm_unet_launch_count": ([], c_int),
    "tdm_unet_launch_name": ([c_int], ctypes.c_char_p),
    "tdm_unet_replay_launch_f32": ([c_f] * 9 + [c_i64, c_int, c_f], c_int),
    "tdm_unet_mark_launch": ([c_int, c_int], c_int),
    "tdm_unet_mark_collect": ([c_f, c_int], c_int),
    "tdm_mse_fwd_bwd_f32": ([c_f, c_f, c_f, c_f, c_f, c_i64, c_f], c_int),
    "tdm_adamw_flat_f32": ([c_f, c_f, c_f, c_f, c_i64, c_float, c_float, c_float, c_float, c_float, c_i64,
                            c_float, c_f], c_int),
    "tdm_unet_loss_grad_f32": ([c_f] * 13 + [c_i64, c_f], c_int),
    "tdm_unet_p_sample_step_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_int, c_f, c_f, c_f, c_i64, c_f], c_int),
    "tdm_conv_nhwc_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_int, c_int, c_int, c_f], c_int),
    "tdm_conv_wgrad_nhwc_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_int, c_int, c_f], c_int),
    "tdm_set_conv_mode": ([c_int], c_int),
    "tdm_set_bwd_overlap": ([c_int], c_int),
    "tdm_get_bwd_overlap": ([], c_int),
    "tdm_set_early_grads": ([c_int], c_int),
    "tdm_get_early_grads": ([], c_int),
    "tdm_unet_early_grad_offset": ([], c_i64),
    "tdm_unet_wait_early_grads": ([c_f], c_int),
    "tdm_get_conv_mode": ([], c_int),
    "tdm_conv_nhwc_s16_f32": ([c_f] * 10 + [c_i64, c_int, c_int, c_int, c_int, c_int, c_f], c_int),
    "tdm_conv_wgrad_nhwc_s16_f32": ([c_f] * 5 + [c_i64, c_int, c_int, c_int, c_int, c_f], c_int),
    "tdm_resblock_scratch_floats": ([c_i64, c_int, c_int, c_int], c_i64),
    "tdm_resblock_fwd_f32": ([c_f] * 12 + [c_i64, c_int, c_int, c_int, c_f], c_int),
    "tdm_layernorm_residual_fwd_f32": ([c_f] * 8 + [c_i64, c_int, c_f], c_int),
    "tdm_layernorm_scratch_floats": ([c_int], c_i64),
    "tdm_layernorm_residual_bwd_f32": ([c_f] * 8 + [c_i64, c_int, c_f], c_int),
    "tdm_set_gemm_mode": ([c_int], c_int),
    "tdm_get_gemm_mode": ([], c_int),
    "tdm_tt_param_count": ([c_int, c_int, c_int], c_i64),
    "tdm_tt_param_offsets": ([c_int, c_int, c_int, ctypes.POINTER(ctypes.c_int64)], c_int),
    "tdm_tt_workspace_floats": ([c_i64, c_int, c_int, c_int, c_int, c_int, c_int], c_i64),
    "tdm_tt_slab_floats": ([c_int, c_int, c_int], c_i64),
    "tdm_tt_fwd_f32": ([c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_u64, c_f], c_int),
    "tdm_tt_bwd_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_int, c_int, c_int, c_float, c_u64, c_f], c_int),
    "tdm_tt_loss_grad_f32": ([c_f] * 13 + [c_i64, c_int, c_int, c_int, c_int, c_int, c_float, c_u64, c_f], c_int),
    "tdm_set_attn_mode": ([c_int], c_int),
    "tdm_get_attn_mode": ([], c_int),
    "tdm_attention_fwd_f32": ([c_f, c_f, c_f, c_i64, c_int, c_int, c_int, c_float, c_u64, c_int, c_f], c_int),
    "tdm_attention_bwd_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_int, c_float, c_u64, c_int, c_f], c_int),
    "tdm_attention_step_form_f32": ([c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_int, c_float, c_u64, c_int, c_f], c_int),
    "tdm_dropout_keep_u8": ([c_float, c_u64, c_int, c_i64, c_i64, c_f], c_int),
    "tdm_dropout_keep_salted_u8": ([c_float, c_u64, ctypes.c_uint32, c_int, c_i64, c_i64, c_f], c_int),
    "tdm_tt_loss_grad_philox_f32": ([c_f, c_f, c_f, c_f, c_u64] + [c_f] * 10 + [c_i64, c_int, c_int, c_int, c_int, c_int, c_float, c_u64, c_f], c_int),
    "tdm_tt_p_sample_step_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_int, c_f, c_f, c_f, c_i64, c_int, c_int, c_int,
                                  c_int, c_int, c_f], c_int),
    "tdm_tt_p_sample_step_philox_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_u64, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_int,
                                         c_int, c_int, c_f], c_int),
    "tdm_scale_by_table_f32": ([c_f, c_f, c_f, c_f, c_i64, c_i64, c_f], c_int),
    "tdm_embed_gather_f32": ([c_f, c_f, c_f, c_i64, c_int, c_int, c_f], c_int),
    "tdm_embed_scatter_add_f32": ([c_f, c_f, c_f, c_i64, c_int, c_int, c_float, c_f], c_int),
    "tdm_round_workspace_floats": ([c_i64, c_int, c_int], c_i64),
    "tdm_round_ce_loss_grad_f32": ([c_f, c_f, c_f, c_f, c_float, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_f], c_int),
    "tdm_round_workspace_chunked_floats": ([c_i64, c_int, c_int, c_int], c_i64),
    "tdm_round_fused_ok": ([c_i64, c_int, c_int], c_int),
    "tdm_round_workspace_fused_floats": ([c_i64, c_int, c_int, c_int], c_i64),
    "tdm_round_ce_loss_grad_fused_f32": ([c_f, c_f, c_f, c_f, c_float, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_int, c_f], c_int),
    "tdm_round_ce_loss_grad_chunked_f32": ([c_f, c_f, c_f, c_f, c_float, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_int, c_f], c_int),
    "tdm_round_logits_f32": ([c_f, c_f, c_f, c_f, c_i64, c_i64, c_int, c_int, c_f], c_int),
    "tdm_round_argmax_f32": ([c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_f], c_int),
    "tdm_philox_normal_f32": ([c_u64, c_u64, c_f, c_i64, c_f], c_int),
    "tdm_philox_u32": ([c_u64, c_u64, c_int, c_f, c_i64, c_f], c_int),
    "tdm_philox_u32_host": ([c_u64, c_u64, c_int, c_u64, ctypes.POINTER(ctypes.c_uint32)], c_int),
    "tdm_ddpm_draw_q_sample_f32": ([c_f, c_f, c_f, c_u64, c_f, c_f, c_f, c_f, c_i64, c_i64, c_f], c_int),
    "tdm_p_sample_update_philox_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_u64, c_f, c_f, c_i64, c_i64, c_f], c_int),
    "tdm_adamw_flat_devstep_f32": ([c_f, c_f, c_f, c_f, c_i64, c_float, c_float, c_float, c_float, c_float, c_f,
                                    c_float, c_f], c_int),
    "tdm_unet_loss_grad_philox_f32": ([c_f, c_f, c_f, c_f, c_u64] + [c_f] * 10 + [c_i64, c_f], c_int),
    "tdm_unet_loss_grad_philox_epoch_f32": ([c_f] * 5 + [c_i64] * 3 + [c_f, c_f, c_u64] + [c_f] * 10 + [c_i64, c_f], c_int),
    "tdm_unet_p_sample_step_philox_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_u64, c_f, c_f, c_f, c_f, c_i64, c_f], c_int),
    "tdm_ctx_create": ([c_int, ctypes.POINTER(ctypes.c_void_p)], c_int),
    "tdm_ctx_destroy": ([c_f], c_int),
    "tdm_tt_wait_layer_grads": ([c_f, c_int], c_int),
    "tdm_tt_layer_grad_range": ([c_int, c_int, c_int, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)], c_int),
    "tdm_ctx_make_current": ([c_f], c_int),
    "tdm_ctx_current": ([], c_f),
    "tdm_ctx_set_arithmetic": ([c_f, c_int, c_int, c_int], c_int),
    "tdm_ctx_get_arithmetic": ([c_f, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)], c_int),
    "tdm_ctx_set_overlap": ([c_f, c_int, c_int], c_int),
    "tdm_comm_unique_id_bytes": ([], c_int),
    "tdm_comm_unique_id": ([ctypes.c_char_p], c_int),
    "tdm_comm_init": ([c_f, ctypes.c_char_p, c_int, c_int], c_int),
    "tdm_comm_rank": ([c_f], c_int),
    "tdm_comm_world": ([c_f], c_int),
    "tdm_comm_rccl_version": ([], c_int),
    "tdm_allreduce_sum_f32": ([c_f, c_f, c_i64, c_f], c_int),
    "tdm_broadcast_f32": ([c_f, c_f, c_i64, c_int, c_f], c_int),
    "tdm_cosine_argmax_f32": ([c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_f], c_int),
    "tdm_split_s16_f32": ([c_f, c_f, c_i64, c_f], c_int),
    "tdm_tt_loss_grad_philox_dx_f32": ([c_f, c_f, c_f, c_f, c_u64] + [c_f] * 11 + [c_i64, c_int, c_int, c_int, c_int, c_int, c_float,
                                       c_u64, c_f], c_int),
    "tdm_adamw_flat_devsched_f32": ([c_f, c_f, c_f, c_f, c_i64, c_f, c_int, c_float, c_float, c_float, c_float, c_f, c_float, c_f,
                                    c_int, c_f], c_int),
    "tdm_text_combine_dx0_f32": ([c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_i64, c_f], c_int),
    "tdm_text_loss_f32": ([c_f, c_f, c_f, c_f, c_f, c_f], c_int),
    "tdm_ffn_chain_f32": ([c_int, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_float, c_float, c_u64, c_int, c_int,
                           c_i64, c_int, c_int, c_f], c_int),
    "tdm_ffn_chain_mask_count": ([c_i64, c_int], c_i64),
    "tdm_ffn_chain_set_ablate": ([c_int], c_int),
    "tdm_attn_set_ablate": ([c_int], c_int),
    "tdm_gemm_f32": ([c_f, c_i64, c_i64, c_f, c_i64, c_i64, c_f, c_i64, c_f, c_f, c_int, c_int, c_int, c_int, c_int,
                      c_i64, c_f], c_int),
}


def exported_symbols():
    """Names every build of the library must export (mirrors include/tdm_hip.h)."""
    return sorted(_SIGS)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} not found: build it with `python -m tinydiffusionmodels_amd.build` "
                        "(there is no CPU / eager fallback for the HIP path)")
                L = ctypes.CDLL(LIB_PATH)
                ab_mode = "TDM_HIP_LIB" in os.environ     # A/B timing against an older build: tolerate symbols it lacks
                for name, (argtypes, restype) in _SIGS.items():
                    try:
                        fn = getattr(L, name)      # AttributeError if a symbol is missing
                    except AttributeError:
                        if ab_mode:
                            continue
                        raise
                    fn.argtypes = argtypes
                    fn.restype = restype
                _lib = L
    return _lib


class Context:
    """An explicit library context (`tdm_ctx`, include/tdm_hip.h): the arithmetic of the conv / GEMM / attention kernels, the
    launch-overlap switches and the side queue of the two-queue backward.  `with ctx:` binds it to the calling thread for the block
    (the library calls made there use it) and restores the previous binding; a context serves one thread at a time.  Threads that
    bind nothing work on a default context of their own."""

    def __init__(self, device: int = 0, arithmetic=None, bwd_overlap: int = 1, early_grads: int = 0):
        L = lib()
        self._h = ctypes.c_void_p()
        check(L.tdm_ctx_create(device, ctypes.byref(self._h)), "ctx_create")
        if arithmetic is not None:
            check(L.tdm_ctx_set_arithmetic(self._h, *arithmetic), "ctx_set_arithmetic")
        check(L.tdm_ctx_set_overlap(self._h, bwd_overlap, early_grads), "ctx_set_overlap")
        self._prev = []

    @property
    def handle(self):
        return self._h

    def arithmetic(self):
        c, g, a = c_int(), c_int(), c_int()
        check(lib().tdm_ctx_get_arithmetic(self._h, ctypes.byref(c), ctypes.byref(g), ctypes.byref(a)), "ctx_get_arithmetic")
        return c.value, g.value, a.value

    def __enter__(self):
        L = lib()
        self._prev.append(L.tdm_ctx_current())
        check(L.tdm_ctx_make_current(self._h), "ctx_make_current")
        return self

    def __exit__(self, *exc):
        check(lib().tdm_ctx_make_current(self._prev.pop()), "ctx_make_current")
        return False

    def close(self):
        if self._h:
            check(lib().tdm_ctx_destroy(self._h), "ctx_destroy")
            self._h = ctypes.c_void_p()


def arithmetic():
    """(conv, gemm, attention) arithmetic selectors of the calling thread's CURRENT context (`Context`; a thread that has bound
    none has a default context of its own)."""
    L = lib()
    return L.tdm_get_conv_mode(), L.tdm_get_gemm_mode(), L.tdm_get_attn_mode()


class use_arithmetic:
    """Run a block under the selectors another thread recorded: torch runs autograd backward functions on its own engine
    thread, whose current context is that thread's default (a context cannot be current on two threads) — a backward pass must
    run in the arithmetic of ITS forward (the saved workspace holds that arithmetic's tensors), so the bridges record
    `arithmetic()` in forward and copy it into the engine thread's context here."""

    def __init__(self, modes):
        self.modes = modes

    def __enter__(self):
        L = lib()
        self.saved = arithmetic()
        check(L.tdm_set_conv_mode(self.modes[0])); check(L.tdm_set_gemm_mode(self.modes[1])); check(L.tdm_set_attn_mode(self.modes[2]))

    def __exit__(self, *exc):
        L = lib()
        check(L.tdm_set_conv_mode(self.saved[0])); check(L.tdm_set_gemm_mode(self.saved[1])); check(L.tdm_set_attn_mode(self.saved[2]))
        return False


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().tdm_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libtdm_hip {what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libtdm_hip needs tensors on a HIP device (no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError("libtdm_hip needs contiguous tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
