"""ctypes binding of libcerberus_hip.so (the C-ABI declared in include/cerberus_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("CDET_LIB_PATH", _HERE / "libcerberus_hip.so"))  # override: profiling builds (tools/)

BF16, F16, F32, U8 = 0, 1, 2, 3
ACT_NONE, ACT_SILU = 0, 1
CONV_FWD, CONV_DGRAD = 0, 1

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [(n, i32) for n in (
        "N", "Hs", "Ws", "Cs", "Hd", "Wd", "Cd", "kh", "kw", "stride", "pad", "mode", "dtype", "out_dtype", "act",
        "src_ld", "src_coff", "dst_ld", "dst_coff", "res_ld", "res_coff", "accumulate")]


class BnFold(C.Structure):
    """cdet_bn_fold (include/cerberus_hip.h): lives in DEVICE memory, read by the producing kernels at their very end."""
    _fields_ = [("tickets", vp), ("cl_sums", vp), ("totals", vp), ("mean", vp), ("invstd", vp), ("running_mean", vp), ("running_var", vp),
                ("dgamma", vp), ("dbeta", vp), ("inv_count", C.c_double), ("unbias", C.c_double), ("eps", f32), ("momentum", f32),
                ("accumulate", i32), ("nrows", i32), ("C", i32), ("ncl", i32)]


class BnRunningItem(C.Structure):
    _fields_ = [("totals", vp), ("running_mean", vp), ("running_var", vp), ("inv_count", C.c_double), ("unbias", C.c_double),
                ("momentum", f32), ("C", i32)]


BN_FOLD_TICKET_WORDS = 8 * (1 + 128)


class LossDesc(C.Structure):
    _fields_ = [("N", i32), ("nc", i32), ("n_max", i32), ("hw", i32 * 6), ("stride", f32 * 3), ("gain_box", f32),
                ("gain_cls", f32), ("gain_dfl", f32), ("grad_scale", f32), ("dtype", i32), ("grad_dtype", i32),
                ("f_ld", i32), ("topk", i32), ("alpha", f32), ("beta", f32), ("grad_scale_dev", vp)]


class NmsDesc(C.Structure):
    _fields_ = [("N", i32), ("nc", i32), ("A", i32), ("dtype", i32), ("conf_thres", f32), ("iou_thres", f32),
                ("agnostic", i32), ("multi_label", i32), ("max_det", i32), ("max_nms", i32), ("max_cand", i32),
                ("classes", vp), ("n_classes", i32)]


class PackItem(C.Structure):
    _fields_ = [("w_oihw", vp), ("w_fwd", vp), ("w_dgrad", vp), ("O", i32), ("O_pad", i32), ("I", i32), ("kh", i32), ("kw", i32),
                ("first_block", i32), ("n_blocks", i32)]


class PackTiledItem(C.Structure):
    _fields_ = [("w_oihw", vp), ("w_fwd", vp), ("w_dgrad", vp), ("O", i32), ("I", i32), ("kh", i32), ("kw", i32),
                ("first_block", i32), ("n_blocks", i32)]


class LetterboxItem(C.Structure):
    _fields_ = [("img", vp), ("h", i32), ("w", i32), ("pitch", i32), ("new_w", i32), ("new_h", i32), ("top", i32), ("left", i32), ("area", i32)]


class AugTile(C.Structure):
    _fields_ = [("img", vp), ("h0", i32), ("w0", i32), ("pitch", i32), ("h", i32), ("w", i32), ("x1a", i32), ("y1a", i32), ("x2a", i32),
                ("y2a", i32), ("x1b", i32), ("y1b", i32), ("reserved", i32)]


class AugSample(C.Structure):
    _fields_ = [("tiles", AugTile * 8), ("minv", C.c_double * 18), ("mix_ratio", C.c_double), ("n_mosaic", i32), ("flipud", i32),
                ("fliplr", i32), ("use_hsv", i32), ("canvas", i32), ("perspective", i32), ("lut", C.c_uint8 * 768)]


class MergeDesc(C.Structure):
    _fields_ = [("N", i32), ("T", i32), ("max_det", i32), ("iou_thres", f32), ("rows", vp * 8), ("counts", vp * 8), ("cls_offset", i32 * 8)]


class MatchDesc(C.Structure):
    _fields_ = [("N", i32), ("max_det", i32), ("T", i32), ("max_labels", i32)]


class WgradItem(C.Structure):
    _fields_ = [("x", vp), ("dy", vp), ("dw", vp), ("src_ld", i32), ("src_coff", i32)]


class CatSrc(C.Structure):
    _fields_ = [("x", vp), ("ld", i32), ("coff", i32), ("C", i32), ("upsample", i32)]


class ParamSlot(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("mom", vp), ("ema", vp), ("n", i64), ("group", i32), ("weight_decay", f32),
                ("inv_div", f32), ("first_step", i32)]


_SIGS = {
    "cdet_version": (i32, []),
    "cdet_last_error": (C.c_char_p, []),
    "cdet_device_info": (i32, [C.POINTER(i32)]),
    "cdet_set_switch": (i32, [C.c_char_p, i32]),
    "cdet_get_switch": (i32, [C.c_char_p, C.POINTER(i32)]),
    "cdet_active_switches": (i32, [C.c_char_p, i32]),
    "cdet_has_experiments": (i32, []),
    "cdet_conv2d_stat_blocks": (i32, [C.POINTER(ConvDesc)]),
    "cdet_conv2d": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]),
    "cdet_pack_weight": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    "cdet_packed_weight_elems": (i64, [i32, i32, i32, i32, i32]),
    "cdet_pack_weights_batched": (i32, [vp, i32, i32, i32, vp]),
    "cdet_conv2d_tiled_ok": (i32, [C.POINTER(ConvDesc)]),
    "cdet_conv2d_tiled_stat_blocks": (i32, [C.POINTER(ConvDesc)]),
    "cdet_conv2d_tiled": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]),
    "cdet_conv2d_tiled_cat_ok": (i32, [C.POINTER(ConvDesc), C.POINTER(CatSrc), i32]),
    "cdet_conv2d_tiled_cat": (i32, [C.POINTER(ConvDesc), C.POINTER(CatSrc), i32, vp, vp, vp, vp, vp, vp]),
    "cdet_conv2d_s2_tiled_ok": (i32, [C.POINTER(ConvDesc)]),
    "cdet_conv2d_s2_tiled_stat_blocks": (i32, [C.POINTER(ConvDesc)]),
    "cdet_conv2d_s2_tiled": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]),
    "cdet_conv2d_s2_tiled_dgrad": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]),
    "cdet_tiled_weight_elems": (i64, [i32, i32, i32, i32]),
    "cdet_pack_weights_tiled": (i32, [vp, i32, i32, i32, vp]),
    "cdet_pack_weight_tiled": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "cdet_conv2d_wgrad_ws_elems": (i64, [C.POINTER(ConvDesc)]),
    "cdet_conv2d_wgrad": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, i32, vp]),
    "cdet_conv2d_wgrad_groupable": (i32, [C.POINTER(ConvDesc)]),
    "cdet_conv2d_wgrad_grouped_ws_elems": (i64, [C.POINTER(ConvDesc), i32]),
    "cdet_conv2d_wgrad_grouped": (i32, [C.POINTER(ConvDesc), vp, i32, vp, i32, vp]),
    "cdet_stem_conv": (i32, [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "cdet_stem_conv1_ok": (i32, [i32, i32, i32, i32, i32, i32, i32, i32, i32]),
    "cdet_stem_conv1_pack_elems": (i64, [i32]),
    "cdet_stem_conv1_pack": (i32, [vp, vp, i32, i32, vp]),
    "cdet_stem_conv1": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cdet_image_to_nhwc8": (i32, [vp, i32, vp, i32, i32, i32, i32, vp]),
    "cdet_stem_conv_stat_blocks": (i32, [i32, i32, i32]),
    "cdet_stem_conv_wgrad_ws_elems": (i64, [i32, i32, i32]),
    "cdet_stem_conv_wgrad": (i32, [vp, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp]),
    "cdet_fold_padded_wgrad": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "cdet_letterbox_batch": (i32, [vp, i32, vp, i32, i32, i32, i32, vp]),
    "cdet_mosaic_augment_batch": (i32, [vp, i32, vp, i32, vp]),
    "cdet_bn_finalize": (i32, [vp, i32, i32, i64, f32, f32, vp, vp, vp, vp, vp]),
    "cdet_bn_fold_cl_doubles": (i64, [i32, i32]),
    "cdet_conv2d_tiled_bn_ok": (i32, [C.POINTER(ConvDesc)]),
    "cdet_conv2d_tiled_bn": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp]),
    "cdet_conv2d_s2_tiled_bn_ok": (i32, [C.POINTER(ConvDesc)]),
    "cdet_conv2d_s2_tiled_bn": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp]),
    "cdet_bn_running_update": (i32, [vp, i32, i32, vp]),
    "cdet_bn_silu_bwd_reduce_fold": (i32, [vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp]),
    "cdet_bn_silu_fwd": (i32, [vp, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp, i32, i32, i64, i32, i32, vp]),
    "cdet_bn_bwd_blocks": (i32, [i64]),
    "cdet_bn_silu_bwd_reduce": (i32, [vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "cdet_bn_silu_bwd_apply": (i32, [vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, i32, i32,
                                     i64, i32, i32, i64, vp]),
    "cdet_bn_silu_bwd_apply_add": (i32, [vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, i32, i32,
                                         i64, i32, i32, i64, vp, i32, i32, vp]),
    "cdet_bn_bwd_sums": (i32, [vp, i32, i32, vp, vp, vp, i32, vp]),
    "cdet_copy_channels": (i32, [vp, i32, i32, vp, i32, i32, i64, i32, i32, i32, vp]),
    "cdet_colsum": (i32, [vp, i32, i32, i64, i32, i32, i32, vp, i32, vp, vp]),
    "cdet_add_channels": (i32, [vp, i32, i32, vp, i32, i32, vp, i32, i32, i64, i32, i32, vp]),
    "cdet_upsample2": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cdet_upsample2_bwd": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cdet_sppf_pool": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cdet_sppf_pool_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cdet_split3": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "cdet_epilogue_f32": (i32, [vp, i32, i32, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp, i32, i32, i64, i32, vp]),
    "cdet_bn_train_f32_ws_doubles": (i64, [i32]),
    "cdet_bn_train_f32": (i32, [vp, i32, i32, i64, i32, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cdet_bn_silu_bwd_f32": (i32, [vp, i32, i32, vp, i32, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "cdet_colsum_f32": (i32, [vp, i32, i32, i64, i32, vp, vp, vp]),
    "cdet_add_f32": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cdet_maxpool_bwd_f32": (i32, [vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cdet_maxpool_f32": (i32, [vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "cdet_detect_decode": (i32, [vp, vp, vp, C.POINTER(i32), C.POINTER(f32), i32, i32, i32, i32, vp, i32, vp]),
    "cdet_pad_targets": (i32, [vp, vp, vp, i32, i32, i32, f32, f32, vp, vp, vp]),
    "cdet_det_loss_ws_bytes": (i64, [C.POINTER(LossDesc)]),
    "cdet_det_loss": (i32, [C.POINTER(LossDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "cdet_nms_ws_bytes": (i64, [C.POINTER(NmsDesc)]),
    "cdet_match_predictions": (i32, [C.POINTER(MatchDesc), vp, vp, vp, vp, vp, vp, vp]),
    "cdet_merge_tasks": (i32, [C.POINTER(MergeDesc), vp, vp, vp, vp]),
    "cdet_nms_batched": (i32, [C.POINTER(NmsDesc), vp, vp, vp, vp, vp]),
    "cdet_nms_batched_idx": (i32, [C.POINTER(NmsDesc), vp, vp, vp, vp, vp, vp]),
    "cdet_peer_alloc": (i32, [i64, C.POINTER(vp)]),
    "cdet_peer_free": (i32, [vp]),
    "cdet_peer_export": (i32, [vp, vp]),
    "cdet_peer_import": (i32, [vp, C.POINTER(vp)]),
    "cdet_peer_close": (i32, [vp]),
    "cdet_peer_allreduce": (i32, [vp, i32, vp, i32, i32, i64, i64, C.c_uint32, vp, i32, vp]),
    "cdet_grad_sqnorm": (i32, [vp, i32, vp, vp, vp]),
    "cdet_accumulate_clear": (i32, [vp, vp, i64, vp]),
    "cdet_sgd_ema_step": (i32, [vp, i32, vp, f32, C.POINTER(f32), i32, f32, f32, vp, vp, vp]),
    "cdet_scaler_update": (i32, [vp, vp, f32, f32, i32, vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGS.keys())
_lib = None


class CdetError(RuntimeError):
    pass


def load():
    """Load the shared library (once). Raises CdetError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise CdetError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C {_HERE / 'csrc'}`). cerberusdet_amd has no CPU fallback by design.")
    # torch first: it brings its own libamdhip64.so.7, and the extension must resolve its HIP runtime to THAT copy (the same
    # runtime instance that owns torch's streams and allocations). Loaded the other way round, /opt/rocm's copy comes in as a
    # second runtime and every launch fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401

    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # second handle on the tiled convolution for the launch lists' data-gradient entries (same symbol; a distinct __name__ lets
    # the per-entry-point timing in bench.py / tools tell the two uses apart)
    alias = lib._FuncPtr(("cdet_conv2d_tiled", lib))
    alias.restype, alias.argtypes = _SIGS["cdet_conv2d_tiled"]
    alias.__name__ = "cdet_conv2d_tiled_dgrad"
    lib.cdet_conv2d_tiled_dgrad = alias
    if lib.cdet_version() != 1:
        raise CdetError(f"ABI version mismatch: library {lib.cdet_version()}, binding 1")
    _lib = lib
    return lib


def check(code, what=""):
    if code != 0:
        msg = load().cdet_last_error().decode(errors="replace")
        raise CdetError(f"{what} failed ({code}): {msg}")


def is_built() -> bool:
    return LIB_PATH.exists()


SWITCH_DEFAULT = -2147483648


def set_switch(name: str, value=None) -> None:
    """Pin a kernel-form switch of the library (csrc/switches.h; `name` = "conv_pp" or "CDET_CONV_PP"); None restores the default. The library
    reads its CDET_* environment ONCE at load -- tests and A/B tools that compare two forms inside one process go through here."""
    check(load().cdet_set_switch(name.encode(), SWITCH_DEFAULT if value is None else int(value)), "cdet_set_switch")


def get_switch(name: str):
    v = i32(0)
    check(load().cdet_get_switch(name.encode(), C.byref(v)), "cdet_get_switch")
    return None if v.value == SWITCH_DEFAULT else v.value


def active_switches() -> str:
    buf = C.create_string_buffer(1024)
    load().cdet_active_switches(buf, 1024)
    return buf.value.decode()


class switches:
    """Context manager: `with _lib.switches(conv_pp=0, halo_ng=1): ...` pins switches for the block and restores what was there before."""

    def __init__(self, **kw):
        self.kw = kw
        self.old = {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_switch(k)
            set_switch(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:  # (was "not pinned": restoring the DEFAULT is the same thing only for switches whose default is auto)
                load().cdet_set_switch(k.encode(), SWITCH_DEFAULT)
            else:
                set_switch(k, v)
        return False
