"""ctypes binding of libmmfe.so (the C ABI declared in include/mmfe.h).

The shared library is the only compute path of this package: if it cannot be loaded, or no
MI355X-class device is visible, every transform raises.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional, Sequence

import numpy as np

_LIB_NAME = "libmmfe.so"
_lib = None
_lib_lock = threading.Lock()

# --- constants mirrored from include/mmfe.h -------------------------------------------------------
ABI_VERSION = 7
KIND_FBANK, KIND_LOGSPEC, KIND_STFT_MAG, KIND_FRAME, KIND_RESAMPLE = 1, 2, 3, 4, 5
PAD_NONE, PAD_ZERO, PAD_REFLECT = 0, 1, 2
FRAMES_SNIP, FRAMES_CENTER, FRAMES_SCIPY, FRAMES_CEIL = 0, 1, 2, 3
LOG_NONE, LOG_LN_MAX, LOG_LN_ADD, LOG_LOG10_MAX = 0, 1, 2, 3
LAYOUT_TIME_MAJOR, LAYOUT_FEAT_MAJOR = 0, 1
OUT_F32, OUT_F16, OUT_BF16 = 0, 1, 2
IN_F32, IN_S16, IN_F64, IN_F16, IN_BF16 = 0, 1, 2, 3, 4
NOALLOC_MAX_ROWS = 65536

EXPORTED_SYMBOLS = (
    "mmfe_abi_version", "mmfe_last_error", "mmfe_device_count", "mmfe_spec_out_shape",
    "mmfe_plan_create", "mmfe_plan_destroy", "mmfe_run", "mmfe_run_noalloc", "mmfe_run_slices", "mmfe_run_ragged", "mmfe_run_slices_ragged", "mmfe_feature_sums",
    "mmfe_plan_kernel",
    "mmfe_patch_embed_create", "mmfe_patch_embed_destroy", "mmfe_patch_embed_shape", "mmfe_patch_embed_run", "mmfe_patch_embed_update",
)


class MmfeError(RuntimeError):
    """A call into libmmfe.so failed; the message is mmfe_last_error()."""


class MmfeSpec(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("kind", ctypes.c_int32),
        ("n_fft", ctypes.c_int32),
        ("win_len", ctypes.c_int32),
        ("win_off", ctypes.c_int32),
        ("hop", ctypes.c_int32),
        ("pad_mode", ctypes.c_int32),
        ("frames_mode", ctypes.c_int32),
        ("n_mel", ctypes.c_int32),
        ("remove_dc", ctypes.c_int32),
        ("preemph", ctypes.c_float),
        ("power", ctypes.c_int32),
        ("spec_scale", ctypes.c_float),
        ("log_kind", ctypes.c_int32),
        ("log_floor", ctypes.c_float),
        ("out_shift", ctypes.c_float),
        ("out_scale", ctypes.c_float),
        ("out_frames", ctypes.c_int32),
        ("pad_value", ctypes.c_float),
        ("drop_last", ctypes.c_int32),
        ("clamp_below_max", ctypes.c_float),
        ("layout", ctypes.c_int32),
        ("window_f64", ctypes.c_int32),
        ("out_dtype", ctypes.c_int32),
        ("in_dtype", ctypes.c_int32),
    ]


def library_path() -> str:
    """libmmfe.so next to this file; MMFE_LIBRARY_PATH selects another build of the same ABI (kernel tuning)."""
    override = os.environ.get("MMFE_LIBRARY_PATH")
    if override:
        return override
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def bind_library(path: str):
    """Load one build of libmmfe.so from `path` and declare the prototypes (no caching)."""
    if not os.path.exists(path):
        raise MmfeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C multi_modal_amd/csrc`. There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # missing ROCm runtime etc.
        raise MmfeError(f"cannot load {path}: {e}") from e
    c_i64, c_f_p = ctypes.c_int64, ctypes.c_void_p
    lib.mmfe_abi_version.restype = ctypes.c_int
    lib.mmfe_abi_version.argtypes = []
    lib.mmfe_last_error.restype = ctypes.c_char_p
    lib.mmfe_last_error.argtypes = []
    lib.mmfe_device_count.restype = ctypes.c_int
    lib.mmfe_device_count.argtypes = []
    lib.mmfe_spec_out_shape.restype = ctypes.c_int
    lib.mmfe_spec_out_shape.argtypes = [ctypes.POINTER(MmfeSpec), c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]
    lib.mmfe_plan_create.restype = ctypes.c_int
    lib.mmfe_plan_create.argtypes = [ctypes.POINTER(MmfeSpec), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.mmfe_plan_destroy.restype = ctypes.c_int
    lib.mmfe_plan_destroy.argtypes = [ctypes.c_void_p]
    lib.mmfe_run.restype = ctypes.c_int
    lib.mmfe_run.argtypes = [ctypes.c_void_p, c_f_p, c_i64, c_i64, c_i64, c_f_p, ctypes.c_void_p]
    lib.mmfe_run_noalloc.restype = ctypes.c_int
    lib.mmfe_run_noalloc.argtypes = [ctypes.c_void_p, c_f_p, c_i64, c_i64, c_i64, c_f_p, ctypes.c_void_p]
    lib.mmfe_run_slices.restype = ctypes.c_int
    lib.mmfe_run_slices.argtypes = [ctypes.c_void_p, c_f_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    c_i64, c_i64, c_f_p, ctypes.c_void_p]
    lib.mmfe_run_ragged.restype = ctypes.c_int
    lib.mmfe_run_ragged.argtypes = [ctypes.c_void_p, c_f_p, c_i64, ctypes.c_void_p, c_i64, c_i64, c_f_p, ctypes.c_void_p]
    lib.mmfe_run_slices_ragged.restype = ctypes.c_int
    lib.mmfe_run_slices_ragged.argtypes = [ctypes.c_void_p, c_f_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           c_i64, c_i64, c_f_p, ctypes.c_void_p]
    lib.mmfe_feature_sums.restype = ctypes.c_int
    lib.mmfe_feature_sums.argtypes = [c_f_p, c_i64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.mmfe_plan_kernel.restype = ctypes.c_char_p
    lib.mmfe_plan_kernel.argtypes = [ctypes.c_void_p]
    lib.mmfe_patch_embed_create.restype = ctypes.c_int
    lib.mmfe_patch_embed_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.mmfe_patch_embed_destroy.restype = ctypes.c_int
    lib.mmfe_patch_embed_destroy.argtypes = [ctypes.c_void_p]
    lib.mmfe_patch_embed_shape.restype = ctypes.c_int
    lib.mmfe_patch_embed_shape.argtypes = [ctypes.c_void_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]
    lib.mmfe_patch_embed_update.restype = ctypes.c_int
    lib.mmfe_patch_embed_update.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.mmfe_patch_embed_run.restype = ctypes.c_int
    lib.mmfe_patch_embed_run.argtypes = [ctypes.c_void_p, c_f_p, c_i64, c_i64, c_f_p, c_i64, ctypes.c_void_p]
    if lib.mmfe_abi_version() != ABI_VERSION:
        raise MmfeError(f"{path} has ABI version {lib.mmfe_abi_version()}, the Python side expects {ABI_VERSION}")
    return lib


def load_library():
    """Load libmmfe.so once and declare the prototypes.  Raises MmfeError when it is missing."""
    global _lib
    with _lib_lock:
        if _lib is None:
            _lib = bind_library(library_path())
        return _lib


def last_error() -> str:
    return load_library().mmfe_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise MmfeError(f"{what} failed ({rc}): {last_error()}")


def make_spec(**kw) -> MmfeSpec:
    s = MmfeSpec()
    s.struct_size = ctypes.sizeof(MmfeSpec)
    defaults = dict(win_off=0, pad_mode=PAD_NONE, frames_mode=FRAMES_SNIP, n_mel=0, remove_dc=0, preemph=0.0, power=2,
                    spec_scale=1.0, log_kind=LOG_NONE, log_floor=0.0, out_shift=0.0, out_scale=1.0, out_frames=0,
                    pad_value=0.0, drop_last=0, clamp_below_max=0.0, layout=LAYOUT_TIME_MAJOR, n_fft=0, window_f64=0, out_dtype=OUT_F32,
                    in_dtype=IN_F32)
    defaults.update(kw)
    for k, v in defaults.items():
        setattr(s, k, v)
    return s


def spec_out_shape(spec: MmfeSpec, n_samples: int):
    """(frames, features) of one row; pure host arithmetic (works without a GPU)."""
    lib = load_library()
    fr, ft = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib.mmfe_spec_out_shape(ctypes.byref(spec), int(n_samples), ctypes.byref(fr), ctypes.byref(ft)), "mmfe_spec_out_shape")
    return int(fr.value), int(ft.value)


class Plan:
    """Owns one mmfe_plan (device-resident constant tables for one transform on one device)."""

    def __init__(self, spec: MmfeSpec, window: np.ndarray, band_start: Optional[Sequence[int]] = None,
                 band_len: Optional[Sequence[int]] = None, band_weights: Optional[np.ndarray] = None, device: int = 0):
        self._lib = load_library()
        self.spec = spec
        self.device = int(device)
        win = np.ascontiguousarray(window, dtype=np.float64 if spec.window_f64 else np.float32)
        if win.shape != (spec.win_len,):
            raise ValueError(f"window must have {spec.win_len} taps, got {win.shape}")
        bs = bl = bw = None
        if spec.n_mel > 0 and spec.kind != KIND_FRAME:
            bs = np.ascontiguousarray(band_start, dtype=np.int32)
            bl = np.ascontiguousarray(band_len, dtype=np.int32)
            bw = np.ascontiguousarray(band_weights, dtype=np.float32)
            if bs.shape != (spec.n_mel,) or bl.shape != (spec.n_mel,) or bw.shape != (int(bl.sum()),):
                raise ValueError("inconsistent mel band tables")
        handle = ctypes.c_void_p()
        rc = self._lib.mmfe_plan_create(
            ctypes.byref(spec), win.ctypes.data_as(ctypes.c_void_p),
            None if bs is None else bs.ctypes.data_as(ctypes.c_void_p),
            None if bl is None else bl.ctypes.data_as(ctypes.c_void_p),
            None if bw is None else bw.ctypes.data_as(ctypes.c_void_p),
            self.device, ctypes.byref(handle))
        check(rc, "mmfe_plan_create")
        self._handle = handle

    def out_shape(self, n_samples: int):
        return spec_out_shape(self.spec, n_samples)

    def run_ptr(self, in_ptr: int, in_stride: int, n_rows: int, n_samples: int, out_ptr: int, stream: int = 0) -> None:
        check(self._lib.mmfe_run(self._handle, in_ptr, int(in_stride), int(n_rows), int(n_samples), out_ptr,
                                 ctypes.c_void_p(stream) if stream else None), "mmfe_run")

    def run_noalloc_ptr(self, in_ptr: int, in_stride: int, n_rows: int, n_samples: int, out_ptr: int, stream: int = 0) -> None:
        """mmfe_run_noalloc: as run_ptr, with the clamp scratch in the library's static device array (no allocation)."""
        check(self._lib.mmfe_run_noalloc(self._handle, in_ptr, int(in_stride), int(n_rows), int(n_samples), out_ptr,
                                         ctypes.c_void_p(stream) if stream else None), "mmfe_run_noalloc")

    def run_slices_ptr(self, recs_ptr: int, rec_start_ptr: int, rec_index_ptr: int, offsets_ptr: int, n_rows: int,
                       n_samples: int, out_ptr: int, stream: int = 0) -> None:
        check(self._lib.mmfe_run_slices(self._handle, recs_ptr, rec_start_ptr, rec_index_ptr, offsets_ptr, int(n_rows),
                                        int(n_samples), out_ptr, ctypes.c_void_p(stream) if stream else None),
              "mmfe_run_slices")

    def run_ragged_ptr(self, in_ptr: int, in_stride: int, row_len_ptr: int, n_rows: int, n_samples: int, out_ptr: int,
                       stream: int = 0) -> None:
        check(self._lib.mmfe_run_ragged(self._handle, in_ptr, int(in_stride), row_len_ptr, int(n_rows), int(n_samples), out_ptr,
                                        ctypes.c_void_p(stream) if stream else None), "mmfe_run_ragged")

    def run_slices_ragged_ptr(self, recs_ptr: int, rec_start_ptr: int, rec_index_ptr: int, offsets_ptr: int, row_len_ptr: int,
                              n_rows: int, n_samples: int, out_ptr: int, stream: int = 0) -> None:
        check(self._lib.mmfe_run_slices_ragged(self._handle, recs_ptr, rec_start_ptr, rec_index_ptr, offsets_ptr, row_len_ptr,
                                               int(n_rows), int(n_samples), out_ptr, ctypes.c_void_p(stream) if stream else None),
              "mmfe_run_slices_ragged")

    @property
    def kernel(self) -> str:
        """Name of the kernel family serving this plan (diagnostic)."""
        return self._lib.mmfe_plan_kernel(self._handle).decode()

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.mmfe_plan_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def feature_sums_ptr(x_ptr: int, n: int, sums_ptr: int, device: int, stream: int = 0) -> None:
    lib = load_library()
    check(lib.mmfe_feature_sums(x_ptr, int(n), sums_ptr, int(device), ctypes.c_void_p(stream) if stream else None),
          "mmfe_feature_sums")


class PatchEmbedHandle:
    """Owns one mmfe_patch_embed (the 16 x 16 / stride-16 patch convolution's weights in matrix-pipe operand order on one device)."""

    def __init__(self, height: int, width: int, weight: np.ndarray, bias: Optional[np.ndarray], compute_dtype: int, out_dtype: int,
                 device: int = 0, patch: int = 16, in_dtype: int = IN_F32):
        self._lib = load_library()
        w = np.ascontiguousarray(weight, dtype=np.float32)
        if w.ndim != 2 or w.shape[1] != patch * patch:
            raise ValueError(f"weight must be (embed_dim, {patch * patch}), got {w.shape}")
        b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        if b is not None and b.shape != (w.shape[0],):
            raise ValueError(f"bias must have {w.shape[0]} entries, got {b.shape}")
        handle = ctypes.c_void_p()
        check(self._lib.mmfe_patch_embed_create(int(height), int(width), int(patch), int(w.shape[0]), w.ctypes.data_as(ctypes.c_void_p),
                                                None if b is None else b.ctypes.data_as(ctypes.c_void_p), int(in_dtype), int(compute_dtype),
                                                int(out_dtype), int(device), ctypes.byref(handle)), "mmfe_patch_embed_create")
        self._handle = handle
        self.device = int(device)
        self.out_dtype = int(out_dtype)
        n, e = ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.mmfe_patch_embed_shape(self._handle, ctypes.byref(n), ctypes.byref(e)), "mmfe_patch_embed_shape")
        self.n_patches, self.embed_dim = int(n.value), int(e.value)

    def run_ptr(self, x_ptr: int, x_batch_stride: int, n_batch: int, out_ptr: int, out_batch_stride: int, stream: int = 0) -> None:
        check(self._lib.mmfe_patch_embed_run(self._handle, x_ptr, int(x_batch_stride), int(n_batch), out_ptr, int(out_batch_stride),
                                             ctypes.c_void_p(stream) if stream else None), "mmfe_patch_embed_run")

    def update_ptr(self, weight_ptr: int, bias_ptr: int, stream: int = 0) -> None:
        """New weights from device memory ((E, 256) float32 contiguous; bias_ptr 0 = no bias), re-packed by a kernel on `stream`."""
        check(self._lib.mmfe_patch_embed_update(self._handle, weight_ptr, bias_ptr if bias_ptr else None,
                                                ctypes.c_void_p(stream) if stream else None), "mmfe_patch_embed_update")

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.mmfe_patch_embed_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
