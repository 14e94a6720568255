"""ctypes binding of libgprx.so (include/gprx.h).  The only way the package reaches the GPU.

There is no CPU fallback: if the shared library is missing or a call fails, an exception is
raised (``GprxLibraryError`` / the mapped Python exception), never a silent numpy path.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

import os

# (GPRX_LIBRARY: another build of the same library, e.g. an experiment compiled with -D flags -- development A/B runs only)
LIB_PATH = Path(os.environ["GPRX_LIBRARY"]) if os.environ.get("GPRX_LIBRARY") else Path(__file__).resolve().parent / "libgprx.so"

GPRX_OK, GPRX_EINVAL, GPRX_ENOTPD, GPRX_EHIP, GPRX_ENOMEM, GPRX_ESTATE, GPRX_ERCCL = range(7)
UNIQUE_ID_BYTES = 128
TRAIN_VARIANCE, TRAIN_LENGTHSCALE, TRAIN_NOISE, TRAIN_Z = 1, 2, 4, 8
GEMM_C_LOWER, GEMM_A_LOWER, GEMM_A_UPPER, GEMM_B_LOWER, GEMM_B_UPPER = 1, 2, 4, 8, 16

DISTANCE_FORMS = {"difference": 0, "expanded": 1}
KERNEL_IDS = {"RBF": 0, "Matern12": 1, "Matern32": 2, "Matern52": 3, "Exponential": 4}


class GprxLibraryError(RuntimeError):
    """libgprx.so is missing or unusable."""


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p
_i64 = C.c_int64

# name -> (restype, argtypes); every symbol declared in include/gprx.h
PROTOTYPES = {
    "gprx_version": (C.c_int, []),
    "gprx_last_error": (C.c_char_p, [_vp]),
    "gprx_device_count": (C.c_int, [_ip]),
    "gprx_create": (C.c_int, [C.c_int, _i64, C.c_int, _i64, C.c_int, C.c_int, C.POINTER(_vp)]),
    "gprx_destroy": (C.c_int, [_vp]),
    "gprx_set_stream": (C.c_int, [_vp, _vp]),
    "gprx_synchronize": (C.c_int, [_vp]),
    "gprx_set_distance_form": (C.c_int, [_vp, C.c_int]),
    "gprx_set_data": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "gprx_objective": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _dp, _vp]),
    "gprx_factorize": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _vp]),
    "gprx_factorize_many": (C.c_int, [C.c_int, C.POINTER(_vp), _vp, _vp, C.c_int, _vp]),
    "gprx_factorize_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp]),
    "gprx_select_slot": (C.c_int, [_vp, C.c_int]),
    "gprx_last_batch_ms": (C.c_int, [_vp, _dp]),
    "gprx_predict_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, C.c_int]),
    "gprx_predict_batch_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, C.c_int]),
    "gprx_predict": (C.c_int, [_vp, _vp, _i64, _vp, _vp, C.c_int]),
    "gprx_predict_dev": (C.c_int, [_vp, _vp, _i64, _vp, _vp, C.c_int]),
    "gprx_last_timings": (C.c_int, [_vp, _dp]),
    "gprx_set_profiling": (C.c_int, [_vp, C.c_int]),
    "gprx_last_profile": (C.c_int, [_vp, _dp]),
    "gprx_last_kernel_build": (C.c_int, [_vp, _dp, _dp]),
    "gprx_last_cell_kernel": (C.c_int, [_vp, _dp, _dp, _dp]),
    "gprx_predict_batch_t": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, C.c_int]),
    "gprx_objective_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp]),
    "gprx_adam_batch": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "gprx_comm_runtime_check": (C.c_int, [C.c_int]),
    "gprx_comm_unique_id": (C.c_int, [_vp]),
    "gprx_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "gprx_comm_destroy": (C.c_int, [_vp]),
    "gprx_comm_last_error": (C.c_char_p, [_vp]),
    "gprx_comm_rank": (C.c_int, [_vp, _ip, _ip]),
    "gprx_comm_all_gather": (C.c_int, [_vp, _vp, _vp, _i64]),
    "gprx_comm_gather": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int]),
    "gprx_comm_all_reduce_max": (C.c_int, [_vp, _vp, _i64]),
    "gprx_comm_all_gather_host": (C.c_int, [_vp, _vp, _vp, _i64]),
    "gprx_comm_barrier": (C.c_int, [_vp]),
    "gprx_comm_synchronize": (C.c_int, [_vp]),
    "gprx_mem_info": (C.c_int, [C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "gprx_cell_bytes": (C.c_int, [_vp, C.c_int, C.POINTER(_i64)]),
    "gprx_dev_malloc": (C.c_int, [C.c_int, _i64, C.POINTER(_vp)]),
    "gprx_dev_free": (C.c_int, [C.c_int, _vp]),
    "gprx_memcpy_h2d": (C.c_int, [C.c_int, _vp, _vp, _i64]),
    "gprx_memcpy_d2h": (C.c_int, [C.c_int, _vp, _vp, _i64]),
    "gprx_pca_create": (C.c_int, [C.c_int, _i64, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.POINTER(_vp)]),
    "gprx_pca_destroy": (C.c_int, [_vp]),
    "gprx_pca_last_error": (C.c_char_p, [_vp]),
    "gprx_pca_transform": (C.c_int, [_vp, _vp, _i64, _vp]),
    "gprx_pca_reverse": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "gprx_pca_transform_dev": (C.c_int, [_vp, _vp, _i64, _vp]),
    "gprx_pca_reverse_dev": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "gprx_pca_synchronize": (C.c_int, [_vp]),
    "gprx_pca_to_depth_dev": (C.c_int, [_vp, _vp, _i64, C.c_int]),
    "gprx_pca_sqrt_dev": (C.c_int, [_vp, _vp, _i64]),
    "gprx_pca_transpose_dev": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "gprx_metrics": (C.c_int, [C.c_int, _vp, _vp, _vp, _i64, _i64, C.c_int, C.c_double, _vp, _vp, _vp, C.POINTER(C.c_uint64)]),
    "gprx_metrics_dev": (C.c_int, [C.c_int, _vp, _vp, _vp, _i64, _i64, C.c_int, C.c_double, _vp, _vp, _vp, C.POINTER(C.c_uint64)]),
    "gprx_kmeans_pp": (C.c_int, [C.c_int, _vp, _i64, C.c_int, _vp, C.c_int, C.c_int, _i64, _vp, _vp]),
    "gprx_kmeans_lloyd": (C.c_int, [C.c_int, _vp, _i64, C.c_int, _vp, C.c_int, C.c_double, C.c_int, _vp, _ip, _ip]),
    "gprx_gather_rows": (C.c_int, [C.c_int, _vp, _i64, _i64, _vp, _vp]),
    "gprx_kmat": (C.c_int, [C.c_int, C.c_int, _vp, _i64, _vp, _i64, C.c_int, _vp, C.c_double, C.c_double, _vp, _i64, _i64, _i64, C.c_int]),
    "gprx_gemm": (C.c_int, [C.c_int, C.c_int, C.c_int, _i64, _i64, _i64, C.c_double, _vp, _i64, _vp, _i64, C.c_double, _vp, _i64, C.c_int, C.c_int]),
    "gprx_potrf": (C.c_int, [C.c_int, _vp, _i64, _i64, _i64, _vp, _ip]),
    "gprx_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "gprx_set_handle_tuning": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "gprx_mfma_f64_peak": (C.c_int, [C.c_int, _dp]),
    "gprx_exp_probe": (C.c_int, [C.c_int, C.c_int, _vp, _i64, _vp]),
}

_lib = None


def load():
    """Load libgprx.so once and attach prototypes.  Raises GprxLibraryError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise GprxLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  gpras_amd has no CPU fallback."
        )
    try:
        lib = C.CDLL(str(LIB_PATH))
    except OSError as exc:  # missing ROCm runtime etc.
        raise GprxLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise GprxLibraryError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error(handle=None) -> str:
    msg = load().gprx_last_error(handle)
    return msg.decode() if msg else ""


def check(rc: int, handle=None) -> None:
    """Map a gprx status to the exception the reference's caller would have seen."""
    if rc == GPRX_OK:
        return
    msg = last_error(handle)
    if rc == GPRX_EINVAL:
        raise ValueError(msg)
    if rc == GPRX_ENOTPD:
        raise np.linalg.LinAlgError(msg)  # tensorflow raises InvalidArgumentError from Cholesky here
    if rc == GPRX_ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(f"libgprx error {rc}: {msg}")


def as_f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a: np.ndarray):
    # (`a.ctypes.data_as(c_void_p)` costs 2.2 us per array, this 1.1 -- five arrays per batched evaluation; a c_void_p rather than the bare
    # int so that a call through a prototype-less function object still passes 64 bits)
    return _vp(a.ctypes.data)


class DeviceBuffer:
    """Owning wrapper of a device allocation made through gprx_dev_malloc."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device = device
        self.nbytes = int(nbytes)
        p = _vp()
        check(load().gprx_dev_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p

    @classmethod
    def from_array(cls, a, device: int = 0) -> "DeviceBuffer":
        a = as_f64(a)
        buf = cls(a.nbytes, device)
        check(load().gprx_memcpy_h2d(device, buf.ptr, ptr(a), a.nbytes))
        return buf

    def to_array(self, shape) -> np.ndarray:
        out = np.empty(shape, dtype=np.float64)
        assert out.nbytes <= self.nbytes
        check(load().gprx_memcpy_d2h(self.device, ptr(out), self.ptr, out.nbytes))
        return out

    def at(self, offset_elems: int):
        return _vp(self.ptr.value + 8 * int(offset_elems))

    def free(self):
        if self.ptr is not None and self.ptr.value:
            load().gprx_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
