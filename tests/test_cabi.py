"""The C-ABI library loads and exports every symbol include/msiren.h declares (no GPU needed),
and the product path fails loudly -- no CPU fallback -- when there is no gfx950 device."""
import ctypes
import os
import re

import numpy as np
import pytest

from mri_inr_amd import _lib


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def declared_symbols():
    text = open(_lib.HEADER_PATH).read()
    return sorted(set(re.findall(r"MSIREN_API\s+[\w\s\*]+?\b(msiren_\w+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/msiren.h but not exported"
    assert set(names) == set(_lib.PROTOTYPES), set(names) ^ set(_lib.PROTOTYPES)


def test_abi_version_and_struct_layout(lib):
    assert lib.msiren_abi_version() == _lib.ABI_VERSION
    # 16 x 4-byte scalars + reserved[4]: must match the C struct (no padding surprises)
    assert ctypes.sizeof(_lib.MsirenConfig) == 20 * 4


def test_no_device_fails_loudly(lib):
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    from mri_inr_amd import ModulatedSiren

    m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
    with pytest.raises(_lib.MsirenError):
        m.forward_mods(np.zeros((5, 1, 256), np.float32))
    with pytest.raises(_lib.MsirenError):
        m(np.zeros((1, 32, 32), np.float32))
    with pytest.raises(_lib.MsirenError):
        m.to("cpu")


def test_invalid_config_rejected_before_any_device_call(lib):
    cfg = _lib.MsirenConfig()
    cfg.abi_version = 999
    h = ctypes.c_void_p()
    assert lib.msiren_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.E_INVALID
    assert b"ABI" in lib.msiren_last_error()
    cfg.abi_version = _lib.ABI_VERSION
    cfg.dim_in, cfg.dim_hidden, cfg.dim_out, cfg.num_layers, cfg.latent_dim = 3, 256, 1, 5, 256
    assert lib.msiren_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.E_INVALID
    assert b"dim_in" in lib.msiren_last_error()
    cfg.dim_in, cfg.dim_out = 2, 2
    assert lib.msiren_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.E_INVALID
    assert b"dim_out" in lib.msiren_last_error()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under mri_inr_amd/ may reference it."""
    root = os.path.dirname(_lib.__file__)
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "from oracle" not in src and "import oracle" not in src, f
    # ... nor the evaluation driver, the diagnostic tools or the examples (bench.py uses it as checker / CPU baseline
    # only, __graft_entry__.smoke() as checker)
    repo = os.path.dirname(root)
    extra = [os.path.join(repo, "test_mod_siren.py")]
    for sub in ("tools", "examples"):
        extra += [os.path.join(repo, sub, f) for f in os.listdir(os.path.join(repo, sub)) if f.endswith((".py", ".c", ".hip"))]
    for path in extra:
        src = open(path).read()
        assert "from oracle" not in src and "import oracle" not in src, path


def test_runtime_info_names_the_hip_runtime_the_library_is_bound_to():
    """msiren_runtime_info (no handle, no device needed): the library links libamdhip64.so.7 by soname and a PyTorch-ROCm wheel bundles a
    libamdhip64.so of the same soname -- in a process that imported torch first every HIP call of libmsiren runs on torch's bundled
    runtime, in a torch-free process on the system one.  Both orders in child processes: the reported path is the file that is mapped
    (/proc/self/maps), there is exactly one libamdhip64 in the process, and the two orders really do end up on different files here."""
    import json
    import subprocess
    import sys

    prog = ("import sys, json\n"
            "if sys.argv[1] == 'torch':\n    import torch\n"
            "from mri_inr_amd import _lib\n"
            "info = _lib.runtime_info()\n"
            "mapped = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
            "print(json.dumps({'info': info, 'mapped': mapped}))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for mode in ("plain", "torch"):
        r = subprocess.run([sys.executable, "-c", prog, mode], cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got[mode] = json.loads(r.stdout.strip().splitlines()[-1])
        info, mapped = got[mode]["info"], got[mode]["mapped"]
        assert len(mapped) == 1, mapped                                         # one HIP runtime per process
        assert os.path.realpath(info["libamdhip64"]) == os.path.realpath(mapped[0]), (info, mapped)
        assert info["hip_runtime_version"] and info["built_against_hip"]
    assert got["plain"]["info"]["torch_bundled"] is False
    try:
        import torch  # noqa: F401
    except Exception:
        return
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):   # a ROCm wheel: the torch-first process is on the bundled runtime
        assert got["torch"]["info"]["torch_bundled"] is True and os.path.realpath(got["torch"]["info"]["libamdhip64"]) == os.path.realpath(tl)
