"""The weight-stationary split-fp16 trunk (mri_inr_amd/csrc/siren_trunk_f16x3w.hip.h; selected with MSIREN_F16_WS=1 when a
handle is created) against the reference fixtures, the fp64 oracle and the register-resident kernel it is an alternative to:
same arithmetic (SirenNet.forward, src/networks/modulated_siren.py:215-233), another data flow."""
import os

import numpy as np
import pytest

from conftest import load_golden, nerr, rms
from mri_inr_amd import ModulatedSiren, synthetic as syn
from oracle import siren_oracle as orc

pytestmark = pytest.mark.gpu


def make(sd, ws, *, L=5, S=24, act="sine", **kw):
    old = {k: os.environ.get(k) for k in ("MSIREN_F16_WS", "MSIREN_F16_HALF")}
    os.environ["MSIREN_F16_WS"] = "1" if ws else "0"   # read once, at msiren_create
    os.environ["MSIREN_F16_HALF"] = "0"                # no half-unit instance: the kernel under test takes every batch size
    try:
        m = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=L, latent_dim=256, w0=kw.pop("w0", 1.0),
                           w0_initial=kw.pop("w0_initial", 30.0), use_bias=kw.pop("use_bias", True), dropout=0.1, modulate=True,
                           encoder_type="custom", encoder_path=None, outer_patch_size=32, inner_patch_size=16,
                           siren_patch_size=S, device="cuda:0", activation=act, precision="f16x3")
        m.load_state_dict(sd)
        m.to("cuda").eval()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    return m


def close(a, b, tol):
    assert a.shape == b.shape and np.isfinite(a).all()
    e = nerr(a, b)
    assert e <= tol, e
    return e


@pytest.mark.parametrize("act", ["sine", "morlet"])
def test_ws_trunk_vs_reference_fixtures_and_the_register_resident_kernel(act):
    g = load_golden(f"trunk_{act}.npz")
    sd = syn.make_state_dict(seed=7)
    w, n = make(sd, True, act=act), make(sd, False, act=act)
    cases = {
        "uniform_B1": syn.make_mods(31, 5, 1, 256),
        "uniform_B64": syn.make_mods(32, 5, 64, 256),
        "sparse_B16": syn.make_mods(33, 5, 16, 256, lo=0.0, hi=2.0, zero_fraction=0.5),
        "modulator_B16": g["modulator_mods"],
    }
    for name, mods in cases.items():
        out = w.forward_mods(mods)
        ref = g[name]
        o = out.reshape(out.shape[0], -1)
        assert nerr(o, ref) <= 1e-4 and rms(o, ref) <= 1e-5, (name, nerr(o, ref), rms(o, ref))
        # same arithmetic in the same order (the canonical last_layer sum: siren_trunk_f16x3n.hip.h): same bits
        assert np.array_equal(out, n.forward_mods(mods)), name


@pytest.mark.parametrize("B", [1, 2, 3, 5, 7, 28, 29, 57, 64, 113, 400, 401, 1000])
def test_ws_trunk_batch_sizes_and_schedules(B):
    """Every shape of schedule (passes of 4, 3 and 2 units, padded last pass, grids below the CU count) against the fp64
    oracle, and each patch independent of the batch it travels in (bit-exact)."""
    sd = syn.make_state_dict(seed=11, trained_like=True)
    w = make(sd, True)
    mods = syn.make_mods(100 + B, 5, B, 256)
    out = w.forward_mods(mods)
    sel = np.unique(np.r_[0, B - 1, np.random.default_rng(B).integers(0, B, 6)])
    ref = orc.siren_forward(sd, mods[:, sel], num_layers=5, dtype=np.float64).reshape(-1, 24, 24)
    close(out[sel], ref, 1e-4)
    # the same patches alone (other schedules, other positions in a pass): same bits
    alone = w.forward_mods(mods[:, sel])
    assert np.array_equal(alone, out[sel])


@pytest.mark.parametrize("L,S", [(3, 24), (4, 10), (5, 7), (5, 33)])
def test_ws_trunk_other_depths_and_ragged_patches(L, S):
    sd = syn.make_state_dict(seed=5, num_layers=L, siren_patch_size=S)
    w = make(sd, True, L=L, S=S)
    mods = syn.make_mods(9, L, 21, 256)
    out = w.forward_mods(mods)
    ref = orc.siren_forward(sd, mods, num_layers=L, dtype=np.float64).reshape(-1, S, S)
    close(out, ref, 1e-4)


def test_ws_forward_tiles_and_slice_pipeline_with_black_tiles():
    sd = syn.make_state_dict(seed=7, trained_like=True)
    w, n = make(sd, True), make(sd, False)
    tiles = np.random.default_rng(3).random((130, 32, 32), dtype=np.float32)
    assert np.array_equal(w(tiles), n(tiles))
    img = syn.make_slice(5, brain_mask=True)          # black corner tiles: the device-side plan path
    assert np.array_equal(w.reconstruct(img), n.reconstruct(img))
    imgs = np.stack([syn.make_slice(k, brain_mask=(k % 2 == 0)) for k in range(3)])
    assert np.array_equal(w.reconstruct(imgs), n.reconstruct(imgs))


def test_every_f16x3_instance_gives_the_same_bits():
    """Default selection (half-unit instance for small batches, weight-stationary for single-stream launches, register-
    resident with two streams) against the forced kernels: a patch comes out bit for bit the same from all of them."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    mods = syn.make_mods(8, 5, 70, 256)
    w, n = make(sd, True), make(sd, False)
    ref = w.forward_mods(mods)
    assert np.array_equal(ref, n.forward_mods(mods))
    d = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=5, latent_dim=256, w0=1.0, w0_initial=30.0, use_bias=True,
                       dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None, outer_patch_size=32,
                       inner_patch_size=16, siren_patch_size=24, device="cuda:0", activation="sine", precision="f16x3")
    d.load_state_dict(sd)
    d.to("cuda").eval()
    assert np.array_equal(d.forward_mods(mods), ref)                 # weight-stationary by default
    assert np.array_equal(d.forward_mods(mods[:, :9]), ref[:9])      # half-unit instance
    from mri_inr_amd import _lib
    _lib.check(d._lib.msiren_set_streams(d._h, 2))                   # two streams: the register-resident trunk
    assert np.array_equal(d.forward_mods(mods), ref)
    _lib.check(d._lib.msiren_set_streams(d._h, 1))


# ---- domain guard of the split-fp16 trunk (include/msiren.h: MSIREN_E_RANGE, msiren_range_events) ----------------------------
def _range_events(m):
    import ctypes as C
    from mri_inr_amd import _lib
    n = C.c_int64()
    _lib.check(m._lib.msiren_range_events(m._h, C.byref(n)))
    return n.value


@pytest.mark.parametrize("forced", ["default", "ws", "n"])
@pytest.mark.parametrize("wscale,mod", [(1e-3, 1e3), (1e-3, 1e5), (1.0, 1e3), (1.0, 1e5), (1e3, 1e3), (1e3, 1e5), (1.0, 3e7)])
def test_f16x3_domain_identical_to_fp32_outside_it(wscale, mod, forced):
    """Modulations are ReLU outputs of trained weights: unbounded in principle.  The split-fp16 trunk carries activation x
    modulation x 2^-a (the next layer's weight scale) in fp16: beyond 65504 it would return inf / NaN where the reference's
    fp32 (modulated_siren.py:215-233) does not.  Behind every f16x3 trunk launch the library enqueues the exact-fp32 trunk as
    a conditional launch on the same stream: whatever the magnitudes, on the synchronous AND the asynchronous entry points,
    with one stream or two, the output buffer holds what the exact-fp32 trunk returns (flagged launches, bit for bit) or meets
    the gate on its own -- never inf, NaN or garbage, and a plain sync() succeeds."""
    from mri_inr_amd import _lib

    L, B = 5, 40
    sd = syn.make_state_dict(seed=11, with_encoder=False)
    sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
    for l in range(1, L):
        sd[f"net.layers.{l}.weight"] = (sd[f"net.layers.{l}.weight"] * np.float32(wscale)).astype(np.float32)
    kw = dict(dim_in=2, dim_hidden=256, dim_out=1, num_layers=L, latent_dim=256, w0=1.0, w0_initial=30.0, use_bias=True, dropout=0.1,
              modulate=True, encoder_type="other", encoder_path=None, outer_patch_size=32, inner_patch_size=16, siren_patch_size=24,
              device="cuda:0", activation="sine")
    env = {"default": {}, "ws": {"MSIREN_F16_WS": "1", "MSIREN_F16_HALF": "0"}, "n": {"MSIREN_F16_WS": "0", "MSIREN_F16_HALF": "0"}}[forced]
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        m = ModulatedSiren(**kw, precision="f16x3")
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    f = ModulatedSiren(**kw, precision="fp32")
    for mm in (m, f):
        mm.load_state_dict({k: v for k, v in sd.items() if k in mm.state_dict()}, strict=False)
        mm.to("cuda").eval()
    mods = (syn.make_mods(5, L, B, 256) * np.float32(mod)).astype(np.float32)
    want = f.forward_mods(mods)                      # exact-fp32 trunk: finite for finite inputs
    assert np.isfinite(want).all()
    e0 = _range_events(m)
    got = m.forward_mods(mods)                       # host-pointer call
    assert np.isfinite(got).all()
    flagged = _range_events(m) > e0
    if flagged:
        assert np.array_equal(got, want)
    else:                                            # inside the domain: the usual gate against fp64 (relaxed to the fp32 trunk's own distance)
        ref = orc.siren_forward(sd, mods, num_layers=L, dtype=np.float64).reshape(-1, 24, 24)
        assert nerr(got, ref) <= max(1e-4, 3 * nerr(want, ref)), (nerr(got, ref), nerr(want, ref))
    # the asynchronous entry points: same buffer contents after a plain sync; in-domain calls in between are untouched
    small = syn.make_mods(6, L, 7, 256)
    small_ref = m.forward_mods(small)
    d_m, d_s = m.device_array(mods.shape).copy_from(mods), m.device_array(small.shape).copy_from(small)
    for streams in (1, 2):
        _lib.check(m._lib.msiren_set_streams(m._h, streams))
        d_o, d_so = [m.device_array((B, 24, 24)) for _ in range(3)], [m.device_array((7, 24, 24)) for _ in range(3)]
        e1 = _range_events(m)
        for k in range(3):
            _lib.check(m._lib.msiren_forward_mods_dev(m._h, d_m.ptr, B, d_o[k].ptr))
            _lib.check(m._lib.msiren_forward_mods_dev(m._h, d_s.ptr, 7, d_so[k].ptr))
        m.sync()                                     # no error: the flag is informational
        assert (_range_events(m) > e1) == flagged
        for o, so in zip(d_o, d_so):
            assert np.array_equal(o.numpy(), got)
            assert np.array_equal(so.numpy(), small_ref)
    _lib.check(m._lib.msiren_set_streams(m._h, 1))


def test_f16x3_domain_nan_modulation_stays_where_the_reference_has_it():
    sd = syn.make_state_dict(seed=7)
    m = make(sd, True)
    mods = syn.make_mods(3, 5, 33, 256)
    mods[2, 17, 100] = np.nan
    from mri_inr_amd import _lib
    d_m = m.device_array(mods.shape).copy_from(mods)
    d_o = m.device_array((33, 24, 24))
    e0 = _range_events(m)
    _lib.check(m._lib.msiren_forward_mods_dev(m._h, d_m.ptr, 33, d_o.ptr))
    m.sync()
    assert _range_events(m) == e0 + 1
    out = d_o.numpy()                                # conditional fp32 launch: NaN where the reference would have it (patch 17 only)
    assert np.array_equal(m.forward_mods(mods), out, equal_nan=True)
    bad = ~np.isfinite(out).reshape(33, -1).all(axis=1)
    assert bad[17] and bad.sum() == 1
