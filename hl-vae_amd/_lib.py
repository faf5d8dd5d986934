"""ctypes binding of libhlvae_hip.so (C ABI declared in include/hlvae_hip.h).

There is no CPU fallback: if the shared library is missing the first use raises, and every
non-zero return code of the library becomes a Python exception (the reference's errors are
Python exceptions too)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HLVAE_LIB_PATH", os.path.join(_HERE, "libhlvae_hip.so"))      # (override: diagnostic builds)
ABI_VERSION = 35
STAT_CHUNKS = 16
HEAD_ACC = 95

_vp = C.c_void_p


class HlvaeVar(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "ncls", "xoff", "sidx", "w_off", "b_off", "e_off", "r_off", "rb_off", "poff", "poff2", "w2_off",
                                            "b2_off", "pad")]


MAX_EXTRA = 3


class HlvaeLayer(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_in", "n_out", "n_in_p", "n_out_p")] + [("o_w", C.c_int64), ("o_b", C.c_int64)]


class HlvaeLayerWs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w", "wT", "a", "aT", "d", "dT")]


class HlvaeDims(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("D", "X", "y_dim", "h_e", "h_d", "L", "n_real", "n_pos", "conv", "Theta",
                                            "Xp", "hep", "hdp", "Lp", "NY", "NYp", "n_stat", "Xe", "Xep", "NYl", "NYlp")]
                + [(n, C.c_int64) for n in ("o_w1", "o_b1", "o_wmu", "o_bmu", "o_wlv", "o_blv", "o_wd", "o_bd",
                                            "o_wy", "o_by", "o_c1w", "o_c1b", "o_c2w", "o_c2b", "o_t1w", "o_t1b", "o_t2w", "o_t2b",
                                            "o_cv_lo", "cv_n", "arena_size", "atomic_region", "frozen_lo", "frozen_hi")]
                + [(n, C.c_int32) for n in ("n_xe", "n_xd", "h_d0", "K1", "K1p", "hd0p")]
                + [("xe", HlvaeLayer * MAX_EXTRA), ("xd", HlvaeLayer * MAX_EXTRA), ("o_xw", C.c_int64),
                   ("lin_e", C.c_int32), ("lin_d", C.c_int32)])


WS_POINTERS = ("P", "G", "w1s", "wmls", "wmlTs", "wds", "wdTs", "wys", "wyTs", "sums", "norm", "xn", "xnT", "xt", "m8",
               "slab", "t", "tT", "mu", "lv", "z", "zb", "zbT", "u", "uT", "dy", "dyT", "log_p_x", "log_p_x_missing",
               "rowpart", "hgpart", "nll", "scal", "klpart", "eps", "rng", "pfull", "xhat", "metpart", "du", "duT", "dz", "dml", "dmlT", "dt", "dtT",
               "w1Ts", "cpack", "img", "yc", "a2", "yv", "da2", "dyc", "dycT", "dfeat", "dimg", "cvpart")
CONV_PART_ROWS = 512
CONV_PACK_ELEMS = 32 * 160 + 16 * 288 + 4 * 16 * 128 + 32 * 256 + 4 * 16 * 64 + 16 * 128      # csrc/conv.hip CP_TOTAL
CONV_FEATURES = 32 * 9 * 9


class HlvaeWs(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("Bp_max", "splitk_enc", "splitk_dec")] + [(n, _vp) for n in WS_POINTERS]
                + [("u0", _vp), ("u0T", _vp), ("xe", HlvaeLayerWs * MAX_EXTRA), ("xd", HlvaeLayerWs * MAX_EXTRA),
                   ("wys_next", _vp), ("wyTs_next", _vp)])


GP_MAX_TERMS, GP_MAX_FACTORS = 8, 4
GP_CAT, GP_BIN, GP_RBF = 0, 1, 2


class HlvaeGpKernel(C.Structure):
    _fields_ = [("n_terms", C.c_int32), ("scale_slot", C.c_int32 * GP_MAX_TERMS), ("n_factors", C.c_int32 * GP_MAX_TERMS),
                ("kind", (C.c_int32 * GP_MAX_FACTORS) * GP_MAX_TERMS), ("dim", (C.c_int32 * GP_MAX_FACTORS) * GP_MAX_TERMS),
                ("ls_slot", (C.c_int32 * GP_MAX_FACTORS) * GP_MAX_TERMS)]


class HlvaeError(RuntimeError):
    pass


_lib = None

_SIGS = {
    "hlvae_abi_version": (C.c_int, []),
    "hlvae_last_error": (C.c_char_p, []),
    "hlvae_struct_sizes": (None, [C.POINTER(C.c_int32)] * 3),
    "hlvae_dims_fill": (None, [C.POINTER(HlvaeDims)]),
    "hlvae_plan_create": (C.c_int, [C.POINTER(_vp), C.POINTER(HlvaeDims), C.POINTER(HlvaeVar), _vp]),
    "hlvae_plan_destroy": (None, [_vp]),
    "hlvae_refresh_shadows": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp]),
    "hlvae_normalize_stats": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, _vp, C.c_int, _vp]),
    "hlvae_normalize_pack": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, _vp, C.c_int, _vp]),
    "hlvae_normalize_fused": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, _vp, C.c_int, _vp]),
    "hlvae_feed_fused": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, _vp, _vp, C.c_int, _vp]),
    "hlvae_feed_prefetch": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, _vp, _vp, C.c_int, _vp]),
    "hlvae_feed_stats": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, _vp, _vp, C.c_int, _vp]),
    "hlvae_feed_pack": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, _vp, _vp, C.c_int, _vp]),
    "hlvae_encoder_fwd": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, C.c_int, C.c_uint64, C.c_int, _vp]),
    "hlvae_decoder_fwd": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "hlvae_step_metrics": (C.c_int, [_vp, C.POINTER(HlvaeWs), C.c_int, _vp, _vp]),
    "hlvae_join": (C.c_int, [_vp, _vp]),
    "hlvae_set_defer_join": (C.c_int, [_vp, C.c_int]),
    "hlvae_scale_dy": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, C.c_int, _vp]),
    "hlvae_backward": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, _vp, C.c_float, C.c_int, C.c_int, _vp]),
    "hlvae_backward_wy": (C.c_int, [_vp, C.POINTER(HlvaeWs), C.c_int, _vp]),
    "hlvae_zero_grad": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp]),
    "hlvae_adam_step": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_float, _vp]),
    "hlvae_backward_adam": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, _vp, C.c_float, C.c_int, _vp, _vp, _vp, C.c_float, C.c_float,
                                      C.c_float, C.c_float, C.c_float, _vp]),
    "hlvae_backward_adam_fused": (C.c_int, [_vp, C.c_int]),
    "hlvae_adam_shard": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_float, C.c_float,
                                   C.c_float, C.c_float, C.c_float, _vp]),
    "hlvae_adam_small": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                   _vp]),
    "hlvae_shadows_from_bf16": (C.c_int, [_vp, C.POINTER(HlvaeWs), _vp, C.c_int64, C.c_uint, _vp]),
    "hlvae_gp_kernel_matrix": (C.c_int, [C.POINTER(HlvaeGpKernel), _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp,
                                         C.c_int, C.c_int, C.c_double, _vp, _vp]),
    "hlvae_gp_chol_inv": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "hlvae_gp_subject_fwd": (C.c_int, [C.POINTER(HlvaeGpKernel), C.POINTER(HlvaeGpKernel), _vp, C.c_int, C.c_int, C.c_int, _vp,
                                       _vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, C.c_double, _vp, _vp, _vp,
                                       _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hlvae_gp_subject_bwd": (C.c_int, [C.POINTER(HlvaeGpKernel), C.POINTER(HlvaeGpKernel), _vp, C.c_int, C.c_int, C.c_int, _vp,
                                       _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _vp,
                                       _vp]),
    "hlvae_gp_param_grad": (C.c_int, [C.POINTER(HlvaeGpKernel), _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp,
                                      C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_double, _vp]),
    "hlvae_gp_transform": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "hlvae_gp_spd_inv2": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "hlvae_gp_gemm": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int, _vp, C.c_int, C.c_int64, _vp, C.c_int, C.c_int64, _vp, C.c_int,
                                C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _vp]),
    "hlvae_gp_gemm_acc": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int, _vp, C.c_int, C.c_int64, _vp, C.c_int, C.c_int64, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_double, _vp]),
    "hlvae_gp_bmv": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_double, C.c_double, _vp]),
    "hlvae_gp_resid": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "hlvae_gp_gemv_t_f32": (C.c_int, [_vp, _vp, C.c_long, C.c_long, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "hlvae_gp_natgrad": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_double, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "hlvae_gp_natgrad_apply": (C.c_int, [_vp, _vp, C.c_double, C.c_int, C.c_int, _vp]),
    "hlvae_gp_bmm": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_double, C.c_double, _vp]),
    "hlvae_gp_gemv_t": (C.c_int, [_vp, _vp, C.c_long, C.c_long, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "hlvae_gp_gkxz": (C.c_int, [_vp, _vp, _vp, C.c_double, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "hlvae_gp_rsym": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_double, C.c_int, C.c_int, _vp, _vp]),
    "hlvae_gp_chain": (C.c_int, [_vp] * 8 + [C.c_double] * 4 + [C.c_int, C.c_int] + [_vp] * 9 + [_vp]),
    "hlvae_gp_chain_rb": (C.c_int, [_vp] * 8 + [C.c_double] * 4 + [C.c_int, C.c_int] + [_vp] * 5 + [_vp]),
    "hlvae_gp_bound": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                 C.c_double, C.c_double, C.c_double, _vp, _vp]),
    "hlvae_gp_adam": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, C.c_double, C.c_double, C.c_double, C.c_double, _vp]),
    "hlvae_gp_state_head": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, _vp,
                                      _vp, _vp, C.c_double, C.c_int, C.c_int, _vp]),
    "hlvae_reset_pending": (C.c_int, [_vp]),
    "hlvae_stamp_slots": (C.c_int, []),
    "hlvae_stamp_words": (C.c_int, []),
    "hlvae_stamp_buffer": (None, [_vp]),
    "hlvae_prof_enable": (None, [C.c_int]),
    "hlvae_prof_report": (C.c_int, [C.c_char_p, C.c_int]),
    "hlvae_gemm_nt_f32": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)


def load():
    """dlopen the in-tree library once; verify ABI version and struct layout."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HlvaeError(f"{LIB_PATH} not found: build it with `make -C hl-vae_amd/csrc` "
                         "(or __graft_entry__.build()); there is no CPU fallback for the HIP path")
    import torch  # noqa: F401  -- FIRST: the library must bind to the HIP runtime torch ships and initialises (a second
    #                              copy of libamdhip64 loaded ahead of torch's sees no device)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.hlvae_abi_version() != ABI_VERSION:
        raise HlvaeError(f"ABI version mismatch: library {lib.hlvae_abi_version()} vs binding {ABI_VERSION}")
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    lib.hlvae_struct_sizes(C.byref(a), C.byref(b), C.byref(c))
    if (a.value, b.value, c.value) != (C.sizeof(HlvaeDims), C.sizeof(HlvaeVar), C.sizeof(HlvaeWs)):
        raise HlvaeError("struct layout mismatch between include/hlvae_hip.h and hl-vae_amd/_lib.py: "
                         f"{(a.value, b.value, c.value)} vs {(C.sizeof(HlvaeDims), C.sizeof(HlvaeVar), C.sizeof(HlvaeWs))}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().hlvae_last_error()
        raise HlvaeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    return None if t is None else _vp(t.data_ptr())
