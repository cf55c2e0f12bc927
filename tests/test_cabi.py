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
