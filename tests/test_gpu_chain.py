"""The Linear layers between the encoder's convolutions and the trunk as ONE launch (mri_inr_amd/csrc/modulator_chain.hip.h:
conv3, Linear(64, Z) and the Modulator layers; reference: `self.modulator(self.encoder(tiles))`,
src/networks/modulated_siren.py:446, 325-343, src/networks/encoding/siren_encoder.py:503-512) against one launch per layer:
same tiles, same MFMA chains, same reduction order -- the modulations, hence the outputs, must be the same bits."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import load_golden, nerr
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn

pytestmark = pytest.mark.gpu


def make(sd, chain, *, L=5, Z=256, spin=None, **kw):
    keys = ("MSIREN_CHAIN", "MSIREN_CHAIN_SPIN")
    old = {k: os.environ.get(k) for k in keys}
    os.environ["MSIREN_CHAIN"] = "1" if chain else "0"   # read once, at msiren_create
    if spin is not None:
        os.environ["MSIREN_CHAIN_SPIN"] = str(spin)
    try:
        m = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=L, latent_dim=Z, w0=1.0, w0_initial=30.0, use_bias=True,
                           dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None, outer_patch_size=32,
                           inner_patch_size=16, siren_patch_size=24, device="cuda:0", activation="sine", **kw)
        m.load_state_dict(sd)
        m.to("cuda").eval()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return m


def chain_info(m):
    active, events = C.c_int32(), C.c_int64()
    _lib.check(m._lib.msiren_chain_info(m._h, C.byref(active), C.byref(events)))
    return active.value, events.value


def tiles_dev(m, tiles):
    """msiren_forward_tiles_dev on resident tiles: the single-stream device path."""
    B = tiles.shape[0]
    d_t = m.device_array(tiles.shape).copy_from(tiles)
    d_o = m.device_array((B, 24, 24))
    _lib.check(m._lib.msiren_forward_tiles_dev(m._h, d_t.ptr, B, d_o.ptr))
    m.sync()
    return d_o.numpy()


def latent_dev(m, z):
    B = z.shape[0]
    d_z = m.device_array(z.shape).copy_from(z)
    d_o = m.device_array((B, 24, 24))
    d_m = m.device_array((m.num_layers, B, 256))
    _lib.check(m._lib.msiren_forward_latent_dev(m._h, d_z.ptr, B, d_o.ptr, d_m.ptr))
    m.sync()
    return d_o.numpy(), d_m.numpy()


@pytest.fixture(scope="module")
def pair():
    sd = syn.make_state_dict(seed=7, trained_like=True)
    return make(sd, True), make(sd, False)


@pytest.mark.parametrize("B", [1, 5, 16, 17, 32, 33, 47, 48, 49, 100, 255, 400, 401, 513, 1000, 3200])
def test_chain_tiles_bit_identical_to_per_layer_launches(pair, B):
    """Every cluster shape: one group, ragged last group, odd group counts (a single group left over after the pairs),
    more groups than clusters x 2 (several rounds per stage), the fused small-batch encoder (B < 48) in front."""
    on, off = pair
    assert chain_info(on) == (1, 0) and chain_info(off)[0] == 0
    tiles = np.random.default_rng(B).random((B, 32, 32), dtype=np.float32)
    a, b = tiles_dev(on, tiles), tiles_dev(off, tiles)
    assert np.isfinite(a).all() and np.array_equal(a, b)
    assert chain_info(on) == (1, 0)   # no silent retreat to the per-layer launches


@pytest.mark.parametrize("B", [1, 16, 31, 400, 777])
def test_chain_modulations_bit_identical(pair, B):
    on, off = pair
    z = np.random.default_rng(1000 + B).standard_normal((B, 256)).astype(np.float32)
    (oa, ma), (ob, mb) = latent_dev(on, z), latent_dev(off, z)
    assert np.array_equal(ma, mb) and np.array_equal(oa, ob)
    assert (ma >= 0).all() and ma.any()   # ReLU outputs, not a buffer nobody wrote


def test_chain_against_reference_fixture():
    """The reference's own modulations for its own latents (tests/golden, generated from the reference)."""
    g = load_golden("trunk_sine.npz")
    if "modulator_latent" not in g:
        pytest.skip("fixture holds no latent")
    sd = syn.make_state_dict(seed=7)
    on = make(sd, True)
    _, mods = latent_dev(on, np.ascontiguousarray(g["modulator_latent"], np.float32))
    assert nerr(mods, g["modulator_mods"]) <= 1e-5


def test_chain_many_launches_of_changing_shape(pair):
    """Hundreds of launches with different stage lists (tiles / latent entry) and batch sizes, each with an epoch of its own
    over the same exchange buffers: a granule of an earlier launch must never pass for a current one."""
    on, off = pair
    rng = np.random.default_rng(5)
    ref = {}
    for it in range(240):
        B = int(rng.choice([1, 16, 48, 100, 400]))
        if B not in ref:
            t = rng.random((B, 32, 32), dtype=np.float32)
            z = rng.standard_normal((B, 256)).astype(np.float32)
            ref[B] = (t, tiles_dev(off, t), z, latent_dev(off, z)[1])
        t, out, z, mods = ref[B]
        if it % 3 == 2:
            assert np.array_equal(latent_dev(on, z)[1], mods), (it, B)
        else:
            assert np.array_equal(tiles_dev(on, t), out), (it, B)
    assert chain_info(on) == (1, 0)


def test_chain_in_the_masked_slice_pipeline(pair):
    """Black-tile plan: the row count comes from the device (plan[0]); chain and per-layer launches see the same rows."""
    on, off = pair
    img = syn.make_slice(3)
    img[:100] = 0.0   # whole tiles black
    a, b = on.reconstruct(img), off.reconstruct(img)
    assert np.array_equal(a, b) and np.isfinite(a).all()
    assert chain_info(on) == (1, 0)


def test_chain_other_depth_and_latent_size():
    sd = syn.make_state_dict(seed=3, num_layers=3, latent_dim=128) if "latent_dim" in syn.make_state_dict.__code__.co_varnames else None
    if sd is None:
        pytest.skip("synthetic state_dict has a fixed latent size")
    on, off = make(sd, True, L=3, Z=128), make(sd, False, L=3, Z=128)
    tiles = np.random.default_rng(9).random((130, 32, 32), dtype=np.float32)
    assert np.array_equal(tiles_dev(on, tiles), tiles_dev(off, tiles))
    assert chain_info(on) == (1, 0)


def test_chain_that_gives_up_is_loud_then_per_layer():
    """MSIREN_CHAIN_SPIN=0: a workgroup gives up at the first poll that finds the previous stage unfinished.  The device
    call's sync reports it; the handle goes back to one launch per layer and is right from then on; a host-pointer call
    re-runs itself and returns the right result."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    off = make(sd, False)
    tiles = np.random.default_rng(2).random((400, 32, 32), dtype=np.float32)
    ref = tiles_dev(off, tiles)
    m = make(sd, True, spin=0)
    with pytest.raises(_lib.MsirenError, match="hand-off"):
        for _ in range(20):   # an idle chip may finish a stage before the first poll now and then
            tiles_dev(m, tiles)
    active, events = chain_info(m)
    assert active == 0 and events >= 1
    assert np.array_equal(tiles_dev(m, tiles), ref)
    m2 = make(sd, True, spin=0)
    small = tiles[:100]   # the host call runs on one stream: chain first, then its own re-run with a launch per layer
    for _ in range(20):
        assert np.array_equal(m2(small), ref[:100])
    assert chain_info(m2)[1] >= 1
