"""ctypes binding of liblafs_hip.so (C ABI in include/lafs_hip.h).

There is NO fallback: if the shared library is missing or a call fails, a LafsHipError is raised.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C lafs_cvpr2024_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblafs_hip.so")


class LafsHipError(RuntimeError):
    pass


class GemmNTArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("epilogue", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int), ("C2", C.c_void_p), ("ldc2", C.c_int),
                ("bias", C.c_void_p), ("resid", C.c_void_p), ("ldr", C.c_int),
                ("seq_scale", C.c_void_p), ("row2seq", C.c_void_p),
                ("aux", C.c_void_p), ("ldaux", C.c_int), ("pos", C.c_void_p), ("npatch", C.c_int),
                ("splits", C.c_int), ("drop_p", C.c_float), ("drop_seed", C.c_uint32),
                ("drop_step", C.c_void_p), ("drop_row0", C.c_int), ("act", C.c_int), ("operand_f16", C.c_int),
                ("ctx", C.c_void_p)]


class MlpArgs(C.Structure):
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int), ("Wa", C.c_void_p), ("ldwa", C.c_int), ("Wb", C.c_void_p), ("ldwb", C.c_int),
                ("M", C.c_int), ("H", C.c_int), ("mode", C.c_int), ("bias_a", C.c_void_p), ("bias_b", C.c_void_p),
                ("resid", C.c_void_p), ("ldr", C.c_int), ("seq_scale", C.c_void_p), ("row2seq", C.c_void_p),
                ("out", C.c_void_p), ("ldo", C.c_int), ("save_grad", C.c_void_p), ("ldsg", C.c_int),
                ("save_act", C.c_void_p), ("ldsa", C.c_int), ("ctx", C.c_void_p),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float),
                ("ln_stats", C.c_void_p), ("ln_out", C.c_void_p), ("ldln", C.c_int),
                ("ln_g_io", C.c_void_p), ("ldgio", C.c_int), ("ln_gb_out", C.c_void_p), ("ldgb", C.c_int), ("ln_part_out", C.c_void_p),
                ("next_ln_gamma", C.c_void_p), ("next_ln_beta", C.c_void_p), ("next_ln_eps", C.c_float),
                ("next_ln_stats", C.c_void_p), ("next_ln_out", C.c_void_p), ("ldnln_next", C.c_int),
                ("proj_x", C.c_void_p), ("ldpx", C.c_int), ("proj_w", C.c_void_p), ("ldpw", C.c_int), ("proj_bias", C.c_void_p),
                ("proj_resid", C.c_void_p), ("ldpr", C.c_int), ("proj_scale", C.c_void_p)]


class WgradItem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int), ("C", C.c_void_p), ("ldc", C.c_int),
                ("N1", C.c_int), ("N2", C.c_int), ("accumulate", C.c_int), ("colsum_a", C.c_void_p)]


class LnFoldItem(C.Structure):
    _fields_ = [("part", C.c_void_p * 4), ("n_parts", C.c_int * 4), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p)]


class RandAugOp(C.Structure):
    _fields_ = [("op", C.c_int32), ("resample", C.c_int32), ("iarg", C.c_int32), ("farg", C.c_float), ("m", C.c_double * 6)]


class BlockOffsets(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("ln1_g", "ln1_b", "w_qkv", "b_qkv", "w_proj", "b_proj", "ln2_g", "ln2_b", "w_fc1", "b_fc1", "w_fc2",
                 "b_fc2", "w_qkv_t", "w_proj_t", "w_fc1_t", "w_fc2_t")]


class TrunkDesc(C.Structure):
    _fields_ = [("dim", C.c_int), ("inner", C.c_int), ("heads", C.c_int), ("mlp", C.c_int), ("depth", C.c_int),
                ("ln_eps", C.c_float), ("attn_scale", C.c_float),
                ("n_tok", C.c_int), ("n_seq", C.c_int), ("max_len", C.c_int),
                ("cu_seqlens", C.c_void_p), ("row2seq", C.c_void_p), ("drop_scales", C.c_void_p),
                ("dropout_p", C.c_float), ("dropout_seed", C.c_uint32),
                ("master", C.c_void_p), ("shadow", C.c_void_p), ("shadow_t", C.c_void_p), ("grad", C.c_void_p),
                ("blocks", C.POINTER(BlockOffsets)),
                ("n_groups", C.c_int), ("group_n_seq", C.c_int * 4), ("group_max_len", C.c_int * 4), ("wgrad_workgroups", C.c_int),
                ("dropout_step", C.c_void_p), ("wgrad_overwrite", C.c_int), ("wgrad_defer", C.c_int), ("ctx", C.c_void_p)]


EPI_BF16, EPI_BF16_GELU, EPI_RESID_F32, EPI_F32, EPI_DGELU_BF16, EPI_ATOMIC_F32, EPI_EMBED_F32, EPI_BF16_ACT = range(8)
ACT_NONE, ACT_RELU, ACT_HSWISH, ACT_HSIGMOID = range(4)
MLP_FWD, MLP_FWD_SAVE, MLP_BWD = range(3)
PATCH_ORDER_CHW, PATCH_ORDER_HWC = 0, 1
CHUNK = 1024
SEG_DECAY, SEG_LAST_LAYER, SEG_TRAINABLE, SEG_LOW_DECAY, SEG_OVERWRITTEN = 1, 2, 4, 8, 16
HP_LR, HP_WD, HP_BETA1, HP_BETA2, HP_EPS, HP_CLIP, HP_EMA_M, HP_FREEZE_LAST, HP_GRAD_SCALE, HP_WD_LOW, HP_STEP, HP_MIX_LAM = range(12)
HP_COUNT = 16
OPT_SIDE_STREAMS, OPT_ROW_CHAINS, OPT_KRES_MASK, OPT_KRES_MIN_ITEMS, OPT_NT_WIDE, OPT_NT_TALL, OPT_COMM_CUS, OPT_NT_BIG, OPT_MLP_FUSED = range(9)

vp, i32, i64, f32, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32

# name -> argument types (the trailing hipStream_t is appended automatically); all return int unless listed
_PROTOS = {
    "lafs_debug_tr16": [vp, vp],
    "lafs_gemm_nt": [C.POINTER(GemmNTArgs)],
    "lafs_mlp_fused": [C.POINTER(MlpArgs)],
    "lafs_gemm_tn_acc": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "lafs_gemm_tn_part": [vp, i32, vp, i32, vp, i32, i64, i32, i32, i32, i32, vp],
    "lafs_reduce_partials": [vp, i64, i32, i64, vp],
    "lafs_sum_slices": [vp, i64, i32, i64, vp],
    "lafs_wgrad": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, i64],
    "lafs_wgrad_f16": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, i64],
    "lafs_wgrad_group": [C.POINTER(WgradItem), i32, i32, i32, vp, i64],
    "lafs_colsum_bf16_acc": [vp, i32, i32, i32, vp],
    "lafs_layernorm_fwd": [vp, i32, vp, vp, f32, vp, i32, vp, i32, vp, i32, i32],
    "lafs_layernorm_bwd": [vp, i32, vp, i32, vp, i32, vp, vp, vp, i32, i32, vp, i32, vp, vp, vp, vp, i32, i32, f32, u32, vp, i32, vp],
    "lafs_layernorm_bwd_fold": [C.POINTER(LnFoldItem), i32, i32],
    "lafs_scale_cast_bf16": [vp, i32, vp, i32, vp, vp, i32, i32, f32, u32, vp, i32],
    "lafs_dropout_f32": [vp, i32, i32, i32, f32, u32, vp],
    "lafs_debug_dropout_mask": [i32, i32, f32, u32, vp],
    "lafs_attention_fwd": [vp, i32, vp, i32, i32, i32, f32, vp, i32, vp],
    "lafs_attention_bwd": [vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, f32, vp, i32],
    "lafs_patchify": [vp, i32, i32, i32, vp],
    "lafs_embed_cls": [vp, vp, vp, i32, i32, i32, i32],
    "lafs_embed_bwd": [vp, i32, i32, i32, i32, vp, vp, vp, vp],
    "lafs_gather_cls": [vp, i32, vp, i32, i32, vp],
    "lafs_scatter_cls": [vp, vp, i32, i32, vp, i32],
    "lafs_l2norm_fwd": [vp, i32, vp, i32, vp, i32, i32],
    "lafs_l2norm_bwd": [vp, i32, vp, i32, vp, vp, i32, i32, i32],
    "lafs_weightnorm_fwd": [vp, vp, i32, i32, i32, vp, vp, i32, vp],
    "lafs_weightnorm_bwd": [vp, vp, vp, vp, i32, i32, vp, vp, i32],
    "lafs_dino_loss_fwd_bwd": [vp, vp, i32, vp, i32, i32, i32, f32, f32, vp, vp, i32, i32, f32, vp, vp],
    "lafs_colsum_f32": [vp, i32, i32, i32, vp],
    "lafs_center_ema": [vp, vp, i32, f32, f32],
    "lafs_grad_sumsq": [vp, vp, i64, i32, vp, vp, vp],
    "lafs_clip_adamw_ema": [vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, i32, vp, vp],
    "lafs_grad_sumsq_range": [vp, vp, i64, i32, i64, i64, i32, i32, vp, vp, vp],
    "lafs_clip_adamw_ema_range": [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, vp, vp, i32, i32, i32, vp, vp],
    "lafs_cast_bf16": [vp, vp, i64],
    "lafs_cast_f32": [vp, vp, i64],
    "lafs_droppath_scales": [vp, i32, i32, u32, vp, vp],
    "lafs_pos_interp_fwd": [vp, vp, vp, i32, i32, i32],
    "lafs_pos_interp_bwd": [vp, vp, vp, i32, i32, i32],
    "lafs_zero_chunks": [vp, vp, vp, i64, i32],
    "lafs_fill_zero": [vp, i64],
    "lafs_transpose_cast_bf16": [vp, i32, i32, vp, i32],
    "lafs_transpose_cast_table": [vp, vp, vp, vp, i32, i32],
    "lafs_trunk_forward": [C.POINTER(TrunkDesc), vp, vp, vp, i32],
    "lafs_trunk_backward": [C.POINTER(TrunkDesc), vp, vp, vp, i32, i32, vp],
    "lafs_trunk_wgrad": [C.POINTER(TrunkDesc), vp, i32, i32],
    "lafs_margin_softmax_ce": [vp, i32, i32, i32, vp, vp, f32, f32, f32, i32, f32, vp, vp],
    "lafs_shard_margin_rowmax": [vp, i32, i32, i32, vp, vp, vp, f32, f32, i32, vp],
    "lafs_shard_margin_rowsum": [vp, i32, i32, i32, vp, vp, vp, f32, f32, i32, vp, vp, vp],
    "lafs_shard_margin_grad": [vp, i32, i32, i32, vp, vp, vp, f32, f32, i32, vp, vp, f32],
    "lafs_cnn_stem": [vp, vp, vp, i32, i32, i32, vp, i32],
    "lafs_cnn_dwconv": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "lafs_cnn_pool": [vp, i32, i32, i32, vp, i32],
    "lafs_cnn_scale_act": [vp, vp, i32, i32, i32, i32, i32],
    "lafs_dwconv_nchw_fwd": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "lafs_dwconv_nchw_bwd_data": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "lafs_dwconv_nchw_bwd_weight": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "lafs_bn_act_fwd_nchw": [vp, vp, vp, vp, vp, f32, f32, i32, i32, i32, i32, i32, vp, vp, vp],
    "lafs_bn_act_bwd_nchw": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp],
    "lafs_cnn_im2col_stem": [vp, i32, i32, vp],
    "lafs_cnn_bn_stats": [vp, i32, i64, i32, vp],
    "lafs_cnn_bn_apply": [vp, i32, i64, i32, vp, vp, vp, f32, f32, vp, vp, i32, vp, i32, vp, i32, vp],
    "lafs_cnn_bn_bwd": [vp, i32, vp, i32, i64, i32, vp, vp, vp, i32, vp, i32, i32, vp, vp, i32, vp, vp, vp],
    "lafs_cnn_bn_bwd_eval": [vp, i32, vp, i32, i64, i32, vp, vp, vp, i32, vp, i32, i32, vp, vp, i32, vp, vp, vp],
    "lafs_cnn_bn_eval_sums": [vp, vp, i64, i32, vp],
    "lafs_cnn_dwconv_train_fwd": [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "lafs_cnn_dwconv_train_bwd": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "lafs_cnn_scale_act_out": [vp, vp, i32, i32, i32, i32, i32, vp],
    "lafs_cnn_se_bwd": [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32],
    "lafs_cnn_act_bwd_post": [vp, vp, vp, i64, i32, vp],
    "lafs_cnn_pool_bwd": [vp, i32, i32, i32, i32, vp],
    "lafs_cnn_pad_cast_table": [vp, vp, vp, vp, i32, i32],
    "lafs_cnn_dw_layout_table": [vp, vp, vp, vp, i32, i32],
    "lafs_cnn_pool_train": [vp, i32, i32, i32, vp, i32],
    "lafs_cnn_unpad_add_table": [vp, vp, vp, vp, i32, i32, vp],
    "lafs_cnn_grad_scale": [vp, i64, f32, vp, i32],
    "lafs_cnn_grad_guard": [vp, i64, vp, f32],
    "lafs_cnn_cast_pad_f16": [vp, i32, i32, vp, i32, vp],
    "lafs_cnn_cast_f16_f32": [vp, vp, i64],
    "lafs_landmark_theta_bwd": [vp, vp, i32, i32, vp],
    "lafs_augment_views": [vp, vp, vp, i32, i32, vp],
    "lafs_randaug_apply": [vp, vp, vp, i32, i32, i32, i32, i32],
    "lafs_landmark_theta": [vp, i32, i32, vp, f32, vp, i32, vp],
    "lafs_mixup_normalize": [vp, vp, i32, i32, f32, vp],
    "lafs_margin_softmax_ce_bf16": [vp, i32, i32, i32, vp, vp, f32, vp, f32, f32, i32, f32, vp, i32, vp, vp, vp],
    "lafs_cast_i64_i32": [vp, vp, i32],
    "lafs_dino_head_loss": [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, f32, f32, vp, vp, vp, i32, f32, vp, vp],
    "lafs_transpose_bf16": [vp, i32, i32, i32, vp, i32],
    "lafs_unpatchify_f32": [vp, i32, i32, i32, vp],
    "lafs_patch_gather_fwd": [vp, vp, i32, i32, i32, vp],
    "lafs_patch_gather_bwd": [vp, vp, vp, i32, i32, i32, vp, vp],
}
_NO_STREAM = {
    "lafs_version": ([], i32),
    "lafs_gemm_nt_slices": ([i32, i32], i32),
    "lafs_gemm_nt_route": ([C.POINTER(GemmNTArgs)], i32),
    "lafs_mlp_fused_supported": ([i32, i32, i32], i32),
    "lafs_mlp_fused_ln_parts": ([i32], i32),
    "lafs_ctx_create": ([i32], vp),
    "lafs_ctx_destroy": ([vp], None),
    "lafs_ctx_set": ([vp, i32, i32], i32),
    "lafs_ctx_get": ([vp, i32], i32),
    "lafs_debug_set": ([i32], i32),
    "lafs_debug_get": ([], i32),
    "lafs_ablation_build": ([], i32),
    "lafs_last_error": ([], C.c_char_p),
    "lafs_embed_bwd_workspace_bytes": ([i32, i32, i32], i64),
    "lafs_dino_loss_workspace": ([i32, i32, i32], i64),
    "lafs_dino_head_loss_workspace": ([i32, i32, i32], i64),
    "lafs_trunk_workspace_bytes": ([C.POINTER(TrunkDesc), i32], i64),
    "lafs_trunk_row_ranges": ([C.POINTER(TrunkDesc)], i32),
    "lafs_layernorm_bwd_parts": ([i32, i32], i32),
    "lafs_wgrad_workspace_bytes": ([i32, i32, i32], i64),
    "lafs_wgrad_group_workspace_bytes": ([C.POINTER(WgradItem), i32, i32, i32], i64),
}
EXPORTED = sorted(list(_PROTOS) + list(_NO_STREAM))

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises LafsHipError when the library is absent."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64.so (same SONAME as /opt/rocm's): it must be mapped FIRST so that this
        # library binds to the very same HIP runtime instance (streams and device pointers are shared with torch).
        import torch  # noqa: F401
        path = LIB_PATH
        if os.environ.get("LAFS_USE_ABLATE_LIB") == "1":          # tools/bench_kernels.py ablate: result-changing timing flags
            path = os.path.join(_HERE, "liblafs_hip_ablate.so")
        elif os.environ.get("LAFS_LIB_VARIANT"):                  # lab: an A/B build of the same sources (csrc/Makefile `variant`)
            path = os.path.join(_HERE, "liblafs_hip_%s.so" % os.environ["LAFS_LIB_VARIANT"])
        if not os.path.isfile(path):
            raise LafsHipError(f"{path} not found: the HIP extension is not built (run __graft_entry__.build()); "
                               "there is no CPU fallback")
        h = C.CDLL(path)
        for name, args in _PROTOS.items():
            fn = getattr(h, name)
            fn.argtypes = list(args) + [vp]
            fn.restype = i32
        for name, (args, res) in _NO_STREAM.items():
            fn = getattr(h, name)
            fn.argtypes = list(args)
            fn.restype = res
        _lib = h
        flags = os.environ.get("LAFS_DEBUG_FLAGS")               # A/B experiments on whole steps (see lafs_debug_set in lafs_hip.h)
        if flags:
            h.lafs_debug_set(int(flags, 0))
    return _lib


class Ctx:
    """Owner of one lafs_ctx (include/lafs_hip.h): the side streams / events of the trunk passes and the kernel-selection options.
    One per engine; the environment is read HERE, never inside the library: LAFS_SINGLE_STREAM=1 / LAFS_ATTN_STREAM=0 (no side
    streams), LAFS_ROW_CHAINS, LAFS_KRES, LAFS_KRES_MIN_ITEMS, LAFS_NT_WIDE, LAFS_NT_TALL, LAFS_NT_BIG, LAFS_MLP_FUSED, LAFS_COMM_CUS."""

    def __init__(self, device=None, options=None, from_env=True):
        import torch
        idx = torch.cuda.current_device() if device is None else torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device()
        self.handle = lib().lafs_ctx_create(int(idx))
        if not self.handle:
            raise LafsHipError("lafs_ctx_create failed: " + lib().lafs_last_error().decode("utf-8", "replace"))
        opts = dict(self.env_options() if from_env else {})
        opts.update(options or {})
        for k, v in opts.items():
            self.set(k, v)

    @staticmethod
    def env_options():
        e, o = os.environ, {}
        if e.get("LAFS_SINGLE_STREAM") == "1" or e.get("LAFS_ATTN_STREAM") == "0":
            o[OPT_SIDE_STREAMS] = 0
        for name, key in (("LAFS_ROW_CHAINS", OPT_ROW_CHAINS), ("LAFS_KRES", OPT_KRES_MASK), ("LAFS_KRES_MIN_ITEMS", OPT_KRES_MIN_ITEMS),
                          ("LAFS_NT_WIDE", OPT_NT_WIDE), ("LAFS_NT_TALL", OPT_NT_TALL), ("LAFS_NT_BIG", OPT_NT_BIG), ("LAFS_MLP_FUSED", OPT_MLP_FUSED),
                          ("LAFS_COMM_CUS", OPT_COMM_CUS)):
            if e.get(name) not in (None, ""):
                o[key] = int(e[name])
        return o

    def set(self, opt, value):
        check(lib().lafs_ctx_set(self.handle, int(opt), int(value)), "lafs_ctx_set")

    def get(self, opt):
        return int(lib().lafs_ctx_get(self.handle, int(opt)))

    def close(self):
        h, self.handle = getattr(self, "handle", None), None
        if h and _lib is not None:
            _lib.lafs_ctx_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_ctx(device=None):
    """The process's default context of `device` (module-level ops and autograd wrappers; engines own theirs)."""
    import torch
    idx = torch.cuda.current_device() if device is None else (torch.device(device).index or 0)
    c = _default_ctx.get(idx)
    if c is None:
        c = _default_ctx[idx] = Ctx(torch.device("cuda", idx))
    return c


def check(rc, what):
    if rc != 0:
        msg = lib().lafs_last_error().decode("utf-8", "replace")
        raise LafsHipError(f"{what} failed with status {rc}: {msg}")


def call(name, *args, stream=None):
    """Invoke a stream-taking entry point on `stream` (default: torch's current HIP stream)."""
    if stream is None:
        import torch
        stream = torch.cuda.current_stream().cuda_stream
    check(getattr(lib(), name)(*args, C.c_void_p(stream)), name)
