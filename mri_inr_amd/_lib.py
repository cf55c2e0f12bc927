"""ctypes binding of libmsiren.so (C ABI: include/msiren.h).

The library is built in-tree (``mri_inr_amd/libmsiren.so``) by ``__graft_entry__.build()`` /
``make -C mri_inr_amd/csrc``.  There is no CPU fallback: if the shared object is missing or no
gfx950 device is visible, importing the model still works (so CPU-only hosts can parse configs
and inspect symbols) but every compute entry point raises.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSIREN_LIB") or os.path.join(_HERE, "libmsiren.so")  # MSIREN_LIB: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "msiren.h")

ABI_VERSION = 3
ACT_SINE, ACT_MORLET = 0, 1
PREC_F32, PREC_BF16, PREC_F16X3, PREC_F16 = 0, 1, 2, 3
E_INVALID, E_STATE, E_SHAPE, E_HIP, E_NOMEM, E_RANGE = -1, -2, -3, -4, -5, -6
COMM_ID_BYTES = 128


class MsirenConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("dim_in", C.c_int32),
        ("dim_hidden", C.c_int32),
        ("dim_out", C.c_int32),
        ("num_layers", C.c_int32),
        ("latent_dim", C.c_int32),
        ("w0", C.c_float),
        ("w0_initial", C.c_float),
        ("use_bias", C.c_int32),
        ("activation", C.c_int32),
        ("outer_patch_size", C.c_int32),
        ("inner_patch_size", C.c_int32),
        ("siren_patch_size", C.c_int32),
        ("residual", C.c_int32),
        ("precision", C.c_int32),
        ("device", C.c_int32),
        ("reserved", C.c_int32 * 4),
    ]


_fp = C.POINTER(C.c_float)
_vp = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32

# name -> (restype, argtypes); every symbol include/msiren.h declares
PROTOTYPES = {
    "msiren_abi_version": (C.c_int, []),
    "msiren_last_error": (C.c_char_p, []),
    "msiren_device_count": (C.c_int, [C.POINTER(_i32)]),
    "msiren_create": (C.c_int, [C.POINTER(MsirenConfig), C.POINTER(_vp)]),
    "msiren_destroy": (C.c_int, [_vp]),
    "msiren_set_tensor": (C.c_int, [_vp, C.c_char_p, _vp, C.c_size_t]),
    "msiren_get_tensor": (C.c_int, [_vp, C.c_char_p, _vp, C.c_size_t]),
    "msiren_commit_weights": (C.c_int, [_vp]),
    "msiren_weights_blob_size": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "msiren_weights_export": (C.c_int, [_vp, _vp, C.c_size_t]),
    "msiren_weights_import": (C.c_int, [_vp, _vp, C.c_size_t]),
    "msiren_forward_mods": (C.c_int, [_vp, _vp, _i64, _vp]),
    "msiren_forward_mods_dev": (C.c_int, [_vp, _vp, _i64, _vp]),
    "msiren_encode_tiles": (C.c_int, [_vp, _vp, _i64, _vp]),
    "msiren_encode_tiles_dev": (C.c_int, [_vp, _vp, _i64, _vp]),
    "msiren_modulate": (C.c_int, [_vp, _vp, _i64, _vp]),
    "msiren_modulate_dev": (C.c_int, [_vp, _vp, _i64, _vp]),
    "msiren_forward_latent": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "msiren_forward_latent_dev": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "msiren_forward_tiles": (C.c_int, [_vp, _vp, _i64, _vp]),
    "msiren_forward_tiles_dev": (C.c_int, [_vp, _vp, _i64, _vp]),
    "msiren_reconstruct_slices": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "msiren_reconstruct_slices_dev": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "msiren_reconstruct_tiles_dev": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "msiren_recon_shape": (C.c_int, [_vp, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "msiren_image_to_patches_dev": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "msiren_weighted_fold_dev": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "msiren_patches_to_image_dev": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "msiren_black_patch_flags_dev": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "msiren_gather_rows_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "msiren_scatter_rows_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "msiren_sync": (C.c_int, [_vp]),
    "msiren_set_streams": (C.c_int, [_vp, _i32]),
    "msiren_dev_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "msiren_dev_free": (C.c_int, [_vp, _vp]),
    "msiren_host_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "msiren_host_free": (C.c_int, [_vp, _vp]),
    "msiren_host_range_kind": (C.c_int, [_vp, C.c_size_t, C.POINTER(_i32)]),
    "msiren_runtime_info": (C.c_int, [C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.c_char_p, C.c_size_t]),
    "msiren_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "msiren_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "msiren_timer_start": (C.c_int, [_vp]),
    "msiren_timer_stop": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "msiren_profile_enable": (C.c_int, [_vp, _i32]),
    "msiren_profile_read": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(C.c_double)]),
    "msiren_profile_read_kernel": (C.c_int, [_vp, _i32, C.c_char_p, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(_i64)]),
    "msiren_last_trunk_kernel": (C.c_int, [_vp, C.c_char_p]),
    "msiren_device_info": (C.c_int, [_vp, C.c_char_p, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_uint64)]),
    "msiren_device_pci": (C.c_int, [_vp, C.c_char_p]),
    "msiren_flops_per_coord": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "msiren_range_events": (C.c_int, [_vp, C.POINTER(_i64)]),
    "msiren_mfma_sustained_probe": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "msiren_comm_unique_id": (C.c_int, [_vp, C.c_size_t]),
    "msiren_comm_init_rank": (C.c_int, [_vp, _vp, C.c_size_t, _i32, _i32]),
    "msiren_comm_init_all": (C.c_int, [C.POINTER(_vp), _i32]),
    "msiren_broadcast_weights": (C.c_int, [_vp, _i32]),
    "msiren_broadcast_weights_all": (C.c_int, [C.POINTER(_vp), _i32, _i32]),
    "msiren_comm_barrier": (C.c_int, [_vp]),
    "msiren_comm_allreduce_max_f64": (C.c_int, [_vp, C.POINTER(C.c_double), _i32]),
    "msiren_comm_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "msiren_comm_destroy": (C.c_int, [_vp]),
    "msiren_trunk_timeline": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "msiren_f16x3_timeline": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "msiren_f16x3w_timeline": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
}

_lib = None
_lock = threading.Lock()


class MsirenError(RuntimeError):
    """HIP/runtime failure inside libmsiren (E_HIP, E_STATE, E_NOMEM)."""


class MsirenRangeError(MsirenError, FloatingPointError):
    """E_RANGE: an operand left the domain of the split-fp16 trunk.  Kept for source compatibility: since round 4 the library
    re-runs such launches on the exact-fp32 trunk on the stream itself and never returns the code (include/msiren.h, "Domain guard")."""


def build(verbose: bool = False) -> str:
    """Compile libmsiren.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-j", str(min(8, os.cpu_count() or 1)), "-C", os.path.join(_HERE, "csrc")]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libmsiren.so failed:\n" + res.stdout[-4000:])
    return LIB_PATH


def load():
    """dlopen the in-tree library and attach prototypes.  Raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        _init_torch_runtime_first()
        if not os.path.exists(LIB_PATH):
            raise MsirenError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C mri_inr_amd/csrc`.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError => header/library out of sync
            fn.restype = res
            fn.argtypes = args
        if lib.msiren_abi_version() != ABI_VERSION:
            raise MsirenError(f"libmsiren ABI {lib.msiren_abi_version()} != binding {ABI_VERSION}")
        _lib = lib
        return lib


def _init_torch_runtime_first():
    """PyTorch-ROCm wheels bundle their own HIP/HSA runtime (``torch/lib/libamdhip64.so``) under the SAME soname this
    library links (``libamdhip64.so.7``).  A process has one of them: whichever is mapped first serves every later
    request for that soname.  So in a host that imported torch first (the reference's own program does:
    test_mod_siren.py imports torch at the top) libmsiren's HIP calls run on torch's bundled runtime -- ROCm 7.0 in
    this image -- and in a torch-free host on the system one (7.2): ``runtime_info()`` / ``msiren_runtime_info`` name
    the file that is mapped.  Both are supported and measured (profiles/r6/02_*).  The one
    ordering that does not work is libmsiren first and torch afterwards: torch then finds the system runtime under its
    soname and ``torch.cuda`` reports no GPUs.  Hence: if the host program has already imported torch, let it bring its
    runtime up before we dlopen ours.  Nothing is imported here -- a torch-free host is unaffected."""
    import sys

    torch = sys.modules.get("torch")
    if torch is None:
        return
    try:
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def runtime_info() -> dict:
    """The HIP runtime libmsiren is bound to in this process (msiren_runtime_info): version, the version the library was
    built against, the driver version, and the path of the mapped libamdhip64."""
    lib = load()
    rv, ba, dv = _i32(), _i32(), _i32()
    buf = C.create_string_buffer(1024)
    lib.msiren_runtime_info(C.byref(rv), C.byref(ba), C.byref(dv), buf, 1024)

    def fmt(v):  # HIP_VERSION = major * 10^7 + minor * 10^5 + patch
        return f"{v // 10000000}.{v // 100000 % 100}.{v % 100000}" if v else None

    path = buf.value.decode()
    return {"hip_runtime_version": fmt(rv.value), "built_against_hip": fmt(ba.value), "hip_driver_version": fmt(dv.value),
            "libamdhip64": path, "torch_bundled": "/torch/lib/" in path}


def last_error() -> str:
    msg = load().msiren_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int):
    """Map C status codes onto the exceptions the reference's Python would raise."""
    if rc == 0:
        return
    msg = last_error()
    if rc in (E_INVALID,):
        raise ValueError(msg)
    if rc == E_RANGE:
        raise MsirenRangeError(msg)
    if rc == E_SHAPE:
        # torch's load_state_dict raises RuntimeError("... size mismatch ...")
        raise RuntimeError(msg)
    raise MsirenError(msg)


def device_count() -> int:
    n = _i32(0)
    load().msiren_device_count(C.byref(n))
    return int(n.value)
