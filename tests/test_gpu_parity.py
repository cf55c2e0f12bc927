"""GPU parity: the HIP path (through the C ABI) against the reference-generated golden fixtures
and against the CPU oracle on seeded inputs.

Gate: max|gpu - ref| / max|ref| <= 1e-4 and RMS <= 1e-5 for the
fp32 configurations.  The oracle is the checker only; nothing here feeds it into the product.
"""
import json
import os

import numpy as np
import pytest

from conftest import load_golden, nerr, rms
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn
from oracle import siren_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-4
RMS_TOL = 1e-5


def make_model(sd, *, H=256, L=5, Z=256, S=24, act="sine", use_bias=True, **kw):
    m = ModulatedSiren(dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=Z, w0=kw.pop("w0", 1.0),
                       w0_initial=kw.pop("w0_initial", 30.0), use_bias=use_bias, dropout=0.1, modulate=True,
                       encoder_type="custom", encoder_path=None, outer_patch_size=32, inner_patch_size=16,
                       siren_patch_size=S, device="cuda", activation=act, **kw)
    m.load_state_dict(sd)
    m.to("cuda")
    m.eval()
    return m


def make_with_env(sd, env, **kw):
    """make_model under environment knobs: they are read once, at msiren_create."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return make_model(sd, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def check(out, ref, tol=TOL, rtol=RMS_TOL):
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert out.dtype == np.float32
    assert np.isfinite(out).all()
    e, r = nerr(out, ref), rms(out, ref)
    assert e <= tol and r <= rtol, (e, r)
    return e


PRECISIONS = ["fp32", "f16x3"]  # exact-fp32 MFMA trunk / split-fp16 trunk: every reference fixture meets both


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("act", ["sine", "morlet"])
def test_trunk_vs_reference_fixtures(act, prec):
    """SirenNet.forward (modulated_siren.py:215-233) at the YAML shape against the reference's outputs, on the
    exact-fp32 kernel (siren_trunk_f32_kernel<256,...>) and on the split-fp16 kernel: same gate."""
    g = load_golden(f"trunk_{act}.npz")
    sd = syn.make_state_dict(seed=7)
    m = make_model(sd, act=act, precision=prec)
    cases = {
        "uniform_B1": syn.make_mods(31, 5, 1, 256),
        "uniform_B64": syn.make_mods(32, 5, 64, 256),
        "sparse_B16": syn.make_mods(33, 5, 16, 256, lo=0.0, hi=2.0, zero_fraction=0.5),
        "modulator_B16": g["modulator_mods"],
    }
    for name, mods in cases.items():
        out = m.forward_mods(mods)
        assert out.shape == (mods.shape[1], 24, 24)
        check(out.reshape(out.shape[0], -1), g[name])
    # the modulator's tuple form is accepted as well
    out2 = m.forward_mods(tuple(cases["uniform_B64"][l] for l in range(5)))
    assert np.array_equal(out2, m.forward_mods(cases["uniform_B64"]))


@pytest.mark.parametrize("act", ["sine", "morlet"])
def test_tiny_vs_reference_fixture(act):
    g = load_golden(f"tiny_{act}.npz")
    meta = json.loads(str(g["meta"]))
    sd = syn.make_state_dict(seed=meta["seed"], dim_hidden=meta["H"], num_layers=meta["L"],
                             latent_dim=meta["Z"], siren_patch_size=meta["S"])
    m = make_model(sd, H=meta["H"], L=meta["L"], Z=meta["Z"], S=meta["S"], act=act, precision="fp32")  # H = 32: fp32 trunk
    mods = syn.make_mods(meta["mods_seed"], meta["L"], meta["B"], meta["H"])
    out = m.forward_mods(mods)
    check(out.reshape(meta["B"], -1), g["out"])


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("preset", ["default", "trained"])
@pytest.mark.parametrize("act", ["sine", "morlet"])
def test_forward_tiles_and_latent_vs_reference(preset, act, prec):
    g = load_golden(f"forward_{preset}_{act}.npz")
    sd = syn.make_state_dict(seed=7, trained_like=(preset == "trained"))
    m = make_model(sd, act=act, precision=prec)
    tiles = np.random.default_rng(42).random((8, 32, 32), dtype=np.float32)
    out = m(tiles)
    check(out, g["out"])
    out_l = m.forward_latent(g["latent"])
    check(out_l, g["out"])
    out_m = m.forward_mods(g["mods"])
    check(out_m, g["out"])


@pytest.mark.parametrize("prec", PRECISIONS)
def test_slice_reconstruction_vs_reference(prec):
    g = load_golden("slice_recon.npz")
    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = make_model(sd, precision=prec)
    img = syn.make_slice(0, 160, 128, brain_mask=True)
    rec = m.reconstruct(img)
    check(rec, g["image"][0])
    rec2 = m.reconstruct(np.stack([img, syn.make_slice(1, 160, 128)]))
    assert rec2.shape == (2, 160, 128)
    assert np.array_equal(rec2[0], rec)


@pytest.mark.parametrize("H,L,S,B,act,bias", [
    (256, 5, 24, 400, "sine", True),     # config 2: one 320x320 slice
    (256, 5, 24, 7, "morlet", True),
    (256, 1, 24, 3, "sine", True),       # no hidden layer at all
    (256, 2, 24, 3, "sine", False),      # use_bias=False
    (100, 3, 10, 5, "sine", True),       # padded hidden width, ragged last coordinate chunk
    (128, 4, 9, 2, "morlet", True),
    (384, 3, 12, 2, "sine", True),
    (512, 3, 8, 2, "sine", True),
])
def test_trunk_vs_oracle_shapes(H, L, S, B, act, bias):
    sd = syn.make_state_dict(seed=3, dim_hidden=H, num_layers=L, siren_patch_size=S, use_bias=bias, with_encoder=False)
    sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
    m = ModulatedSiren(dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=256, w0=1.0, w0_initial=30.0,
                       use_bias=bias, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=S, device="cuda", activation=act,
                       precision="fp32")
    m.load_state_dict(sd, strict=False)
    m.to("cuda")
    mods = syn.make_mods(5, L, B, H)
    out = m.forward_mods(mods)
    ref64 = orc.siren_forward(sd, mods, num_layers=L, activation=act, siren_patch_size=S, dtype=np.float64)
    ref32 = orc.siren_forward(sd, mods, num_layers=L, activation=act, siren_patch_size=S)
    e64 = check(out.reshape(B, -1), ref64)
    e32 = nerr(ref32, ref64)
    # the kernel should sit at the fp32 noise floor, not merely under the gate
    assert e64 <= max(10 * e32, 2e-5), (e64, e32)


def test_nonunit_frequencies():
    sd = syn.make_state_dict(seed=9, w0=2.0)
    m = make_model(sd, w0=2.0, w0_initial=10.0, act="morlet")
    mods = syn.make_mods(6, 5, 4, 256)
    out = m.forward_mods(mods)
    ref = orc.siren_forward(sd, mods, num_layers=5, w0=2.0, w0_initial=10.0, activation="morlet", dtype=np.float64)
    check(out.reshape(4, -1), ref)


def test_empty_batch_and_errors():
    sd = syn.make_state_dict(seed=7)
    m = make_model(sd)
    assert m.forward_mods(np.zeros((5, 0, 256), np.float32)).shape == (0, 24, 24)
    assert m(np.zeros((0, 32, 32), np.float32)).shape == (0, 24, 24)
    with pytest.raises(ValueError):
        m.forward_mods(np.zeros((4, 2, 256), np.float32))
    with pytest.raises(ValueError):
        m(np.zeros((2, 24, 24), np.float32))
    bad = dict(sd)
    bad["net.layers.1.weight"] = np.zeros((256, 255), np.float32)
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(bad)
    bad = dict(sd)
    del bad["net.last_layer.bias"]
    with pytest.raises(RuntimeError, match="Missing key"):
        m.load_state_dict(bad)
    with pytest.raises(ValueError):
        ModulatedSiren(dim_in=3, dim_hidden=256, dim_out=1, num_layers=5, latent_dim=256, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda", activation="sine")


def test_torch_device_tensors_roundtrip():
    import torch

    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = make_model(sd)
    tiles = np.random.default_rng(1).random((5, 32, 32), dtype=np.float32)
    ref = m(tiles)
    t = torch.from_numpy(tiles).cuda()
    out = m(t)
    assert out.is_cuda and out.shape == (5, 24, 24)
    assert np.array_equal(out.cpu().numpy(), ref)
    out_cpu = m(torch.from_numpy(tiles))
    assert not out_cpu.is_cuda and np.array_equal(out_cpu.numpy(), ref)


def test_linearity_free_property_full_size():
    """Size-independent property at config-2 size: patches are independent, so evaluating a batch
    equals evaluating any split of it, and permuting patches permutes outputs (bit-exact)."""
    sd = syn.make_state_dict(seed=7)
    m = make_model(sd)
    mods = syn.make_mods(77, 5, 400, 256)
    full = m.forward_mods(mods)
    a = m.forward_mods(mods[:, :150])
    b = m.forward_mods(mods[:, 150:])
    assert np.array_equal(full, np.concatenate([a, b], 0))
    perm = np.random.default_rng(0).permutation(400)
    assert np.array_equal(m.forward_mods(mods[:, perm]), full[perm])
    assert np.abs(full).max() <= 1.0


def test_config3_full_size_batch_invariance():
    """BASELINE configs[2] size (64 slices = 25 600 tiles in one call), default (f16x3, work-queue) trunk:
    the batch is 64 shuffled copies of one slice's 400 tiles, so every output must equal -- bit for bit --
    the output of the same tile in the 400-tile call, which itself is checked against the fp64 oracle."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = make_model(sd, precision="auto")
    img = syn.make_slice(3)
    tiles, _ = orc.image_to_patches(img, 32, 16)
    base = m(tiles)
    ref = orc.modulated_siren_forward(sd, tiles[:48], num_layers=5, dtype=np.float64)
    check(base[:48], ref)
    idx = np.concatenate([np.random.default_rng(k).permutation(400) for k in range(64)])
    big = m(tiles[idx])
    assert big.shape == (25600, 24, 24)
    assert np.array_equal(big, base[idx])
    # the slice pipeline at the same size: 64 slices in one call == each slice on its own
    imgs = np.stack([syn.make_slice(k, brain_mask=(k % 2 == 0)) for k in range(64)])
    rec = m.reconstruct(imgs)
    assert rec.shape == (64, 320, 320)
    for k in (0, 1, 31, 63):
        assert np.array_equal(rec[k], m.reconstruct(imgs[k]))


@pytest.mark.parametrize("H,Z,L,B", [(256, 256, 5, 400), (256, 256, 5, 3), (64, 48, 3, 17), (100, 24, 2, 5)])
def test_modulator_kernels_vs_oracle(H, Z, L, B):
    """latent -> mods on the device (MFMA kernel when H, Z are multiples of 16, VALU kernel otherwise)."""
    from mri_inr_amd import _lib

    sd = syn.make_state_dict(seed=13, dim_hidden=H, num_layers=L, latent_dim=Z, siren_patch_size=8,
                             modulator_bias_center=0.3, with_encoder=False)
    m = ModulatedSiren(dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=Z, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=8, device="cuda", activation="sine")
    m.load_state_dict(sd, strict=False)
    m.to("cuda")
    z = np.random.default_rng(2).standard_normal((B, Z)).astype(np.float32)
    out = np.empty((B, 8, 8), np.float32)
    mods = np.empty((L, B, H), np.float32)
    _lib.check(m._lib.msiren_forward_latent(m._h, z.ctypes.data, B, out.ctypes.data, mods.ctypes.data))
    ref_mods = orc.modulator_forward(sd, z, num_layers=L, dtype=np.float64)
    assert nerr(mods, ref_mods) < 1e-5
    assert (mods >= 0).all()
    ref = orc.siren_forward(sd, ref_mods, num_layers=L, siren_patch_size=8, dtype=np.float64)
    check(out.reshape(B, -1), ref)


@pytest.mark.parametrize("B", [1, 50])
def test_encoder_with_a_latent_size_the_mfma_kernels_do_not_take(B):
    """latent_dim = 24 (not a multiple of 16): the fused per-tile encoder kernel (VALU; encoder_modulator.hip.h) and the VALU Modulator
    kernel are what runs -- the only shapes they still serve since round 6.  Latent and output against the fp64 oracle
    (siren_encoder.py:503-512, 565-577; modulated_siren.py:325-343)."""
    H, Z, L = 100, 24, 3
    sd = syn.make_state_dict(seed=17, dim_hidden=H, num_layers=L, latent_dim=Z, trained_like=True)
    m = make_model(sd, H=H, L=L, Z=Z)
    tiles = np.random.default_rng(B).random((B, 32, 32), dtype=np.float32)
    z = m.encoder(tiles)
    assert z.shape == (B, Z) and nerr(z, orc.encoder_forward(sd, tiles, dtype=np.float64)) < 1e-5
    check(m(tiles), orc.modulated_siren_forward(sd, tiles, num_layers=L, dtype=np.float64))
    assert np.array_equal(m.encoder(tiles[:1]), z[:1])            # one kernel at every batch size: same bits


@pytest.mark.parametrize("L,S,B", [(5, 24, 400), (5, 24, 7), (2, 24, 5), (3, 10, 9), (4, 24, 1030), (6, 8, 33), (8, 24, 50),
                                   (11, 24, 20)])
def test_f16x3_trunk_vs_oracle_shapes(L, S, B):
    sd = syn.make_state_dict(seed=4, num_layers=L, siren_patch_size=S, with_encoder=False)
    sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
    m = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=L, latent_dim=256, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=S, device="cuda", activation="sine",
                       precision="f16x3")
    m.load_state_dict(sd, strict=False)
    m.to("cuda")
    mods = syn.make_mods(5, L, B, 256)
    out = m.forward_mods(mods)
    ref64 = orc.siren_forward(sd, mods, num_layers=L, siren_patch_size=S, dtype=np.float64)
    e64 = check(out.reshape(B, -1), ref64)
    e32 = nerr(orc.siren_forward(sd, mods, num_layers=L, siren_patch_size=S), ref64)
    assert e64 <= max(10 * e32, 2e-5), (e64, e32)
    # determinism + batch-split equivariance (persistent grid: unit -> wave mapping changes with B)
    assert np.array_equal(out, m.forward_mods(mods))
    k = B // 3
    if k:
        assert np.array_equal(out[:k], m.forward_mods(mods[:, :k]))


@pytest.mark.parametrize("act", ["sine", "morlet"])
def test_half_unit_trunk_instance_is_bit_identical(act):
    """Small batches run the half-unit instance of the split-fp16 trunk (16 coordinates per wave,
    siren_trunk_f16x3h.hip.h); larger ones the 32-coordinate kernel.  Same arithmetic in the same order: a patch
    must come out bit for bit the same whichever instance evaluated it (and both meet the reference fixture)."""
    g = load_golden(f"trunk_{act}.npz")
    sd = syn.make_state_dict(seed=7)
    m = make_model(sd, act=act, precision="f16x3")
    mods = syn.make_mods(32, 5, 64, 256)                      # the "uniform_B64" fixture case: 64 x 18 units > 512
    big = m.forward_mods(mods)
    check(big.reshape(64, -1), g["uniform_B64"])
    for n in (1, 5, 28):                                      # <= 28 patches: all units fit one round as half-units
        small = m.forward_mods(mods[:, :n])
        assert np.array_equal(small, big[:n]), n
    one = m.forward_mods(syn.make_mods(31, 5, 1, 256))        # BASELINE configs[0]: a single tile
    check(one.reshape(1, -1), g["uniform_B1"])


@pytest.mark.parametrize("wscale,mlo,mhi", [(1.0, 0.5, 1.5), (0.25, 0.5, 1.5), (4.0, 0.05, 0.15), (1.0, 0.0005, 0.002),
                                            (0.5, 5.0, 20.0), (2.0, 0.5, 1.5)])
def test_f16x3_scale_folding_over_weight_and_modulation_magnitudes(wscale, mlo, mhi):
    """The split-fp16 trunk scales each hidden layer's weights by 2^a (rms|W'| ~ 0.1) and undoes it on the activation side
    through the previous layer's modulation row, so both fp16 `lo` parts sit at the edge of the subnormal range.  Small /
    large weights and tiny / large modulations move operands deeper into it (or towards overflow): the kernel must stay
    at the fp32 noise floor of the same model (reference arithmetic: modulated_siren.py:215-233 in fp32 vs fp64)."""
    L, B = 5, 40
    sd = syn.make_state_dict(seed=11, with_encoder=False)
    sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
    for l in range(1, L):
        sd[f"net.layers.{l}.weight"] = (sd[f"net.layers.{l}.weight"] * np.float32(wscale)).astype(np.float32)
    m = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=L, latent_dim=256, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda", activation="sine",
                       precision="f16x3")
    m.load_state_dict(sd, strict=False)
    m.to("cuda")
    mods = syn.make_mods(5, L, B, 256, lo=mlo, hi=mhi)
    out = m.forward_mods(mods).reshape(B, -1)
    ref64 = orc.siren_forward(sd, mods, num_layers=L, dtype=np.float64)
    ref32 = orc.siren_forward(sd, mods, num_layers=L)
    assert np.isfinite(out).all()
    scale = max(np.abs(ref64).max(), 1e-30)
    e64, e32 = np.abs(out - ref64).max() / scale, np.abs(ref32 - ref64).max() / scale
    assert e64 <= max(10 * e32, 2e-5), (e64, e32)
    assert e64 <= 1e-4 or e32 > 2e-5, (e64, e32)   # the gate, unless fp32 itself is already that far from fp64 here


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_f16x3_log_uniform_modulations_within_a_row_and_outlier_weight_rows(seed):
    """The shape in which ONE power-of-two scale per hidden layer helps least (round-4 review): modulations drawn log-uniformly
    from 1e-6 ... 1e3 WITHIN each row -- tiny and large side by side, so no single exponent suits a row -- and hidden weight
    matrices with a few rows 100 x larger than the rest (the layer's scale is set by them; every other row sits 100 x deeper in
    fp16's subnormal range).  Gate as test_f16x3_scale_folding_...: within the larger of 10 x the distance of the reference's
    own fp32 arithmetic (modulated_siren.py:215-233) from fp64, and 2e-5; rows that leave the fp16 domain are repaired to the
    exact-fp32 trunk's bits by the conditional launch, so the output is finite either way."""
    L, B = 5, 48
    rng = np.random.default_rng(100 + seed)
    sd = syn.make_state_dict(seed=11 + seed, with_encoder=False)
    sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
    for l in range(1, L):
        w = sd[f"net.layers.{l}.weight"].copy()
        w[rng.choice(256, 4, replace=False)] *= np.float32(100.0)
        sd[f"net.layers.{l}.weight"] = w
    kw = dict(dim_in=2, dim_hidden=256, dim_out=1, num_layers=L, latent_dim=256, w0=1.0, w0_initial=30.0, use_bias=True, dropout=0.1,
              modulate=True, encoder_type="custom", encoder_path=None, outer_patch_size=32, inner_patch_size=16, siren_patch_size=24,
              device="cuda", activation="sine")
    m, f = ModulatedSiren(**kw, precision="f16x3"), ModulatedSiren(**kw, precision="fp32")
    for mm in (m, f):
        mm.load_state_dict(sd, strict=False)
        mm.to("cuda")
    for hi in (0.0, 3.0):   # 1e-6 ... 1 (inside the fp16 domain) and 1e-6 ... 1e3 (rows beyond 65 504 / 2^a: repaired)
        mods = (10.0 ** rng.uniform(-6.0, hi, (L, B, 256))).astype(np.float32)
        out = m.forward_mods(mods).reshape(B, -1)
        ref64 = orc.siren_forward(sd, mods, num_layers=L, dtype=np.float64)
        ref32 = orc.siren_forward(sd, mods, num_layers=L)
        assert np.isfinite(out).all()
        scale = max(np.abs(ref64).max(), 1e-30)
        e64, e32 = np.abs(out - ref64).max() / scale, np.abs(ref32 - ref64).max() / scale
        assert e64 <= max(10 * e32, 2e-5), (hi, e64, e32)
        # per row as well: a row of small outputs is not hidden behind the batch's largest
        rows = np.abs(out - ref64).max(axis=1) / np.maximum(np.abs(ref64).max(axis=1), 1e-30)
        rows32 = np.abs(ref32 - ref64).max(axis=1) / np.maximum(np.abs(ref64).max(axis=1), 1e-30)
        assert (rows <= np.maximum(10 * rows32, 5e-5)).all(), (hi, rows.max(), rows32.max())
        exact = f.forward_mods(mods).reshape(B, -1)
        assert np.abs(out - exact).max() / scale <= max(10 * e32, 2e-5)


def test_f16x3_full_forward_matches_fp32_path():
    sd = syn.make_state_dict(seed=7, trained_like=True)
    m32 = make_model(sd, precision="fp32")
    m16 = make_model(sd, precision="f16x3")
    assert m32._config().precision != m16._config().precision
    tiles = np.random.default_rng(3).random((50, 32, 32), dtype=np.float32)
    a, b = m32(tiles), m16(tiles)
    ref = orc.modulated_siren_forward(sd, tiles, num_layers=5, dtype=np.float64)
    assert nerr(a, ref) < 1e-4 and nerr(b, ref) < 1e-4
    assert nerr(b, a) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("L", [3, 4, 6])
def test_two_streams_at_other_depths_same_bits_as_one_stream(L):
    """Depths other than 5 run the loop form of the register-resident trunk in two-stream mode (with the ring of 4: the ring-of-3
    instance of that form spills); one stream: the weight-stationary trunk (3..5) or the same loop form.  Same bits."""
    from mri_inr_amd import _lib

    sd = syn.make_state_dict(seed=5, num_layers=L, trained_like=True)
    m = make_model(sd, L=L, precision="f16x3")
    tiles = [np.random.default_rng(10 * L + s).random((60 + 97 * s, 32, 32), dtype=np.float32) for s in range(4)]
    d_in = [m.device_array(t.shape).copy_from(t) for t in tiles]
    d_out = [m.device_array((t.shape[0], 24, 24)) for t in tiles]
    for a, b, t in zip(d_in, d_out, tiles):
        _lib.check(m._lib.msiren_forward_tiles_dev(m._h, a.ptr, t.shape[0], b.ptr))
    m.sync()
    one = [b.numpy() for b in d_out]
    _lib.check(m._lib.msiren_set_streams(m._h, 2))
    for _ in range(2):
        for a, b, t in zip(d_in, d_out, tiles):
            _lib.check(m._lib.msiren_forward_tiles_dev(m._h, a.ptr, t.shape[0], b.ptr))
    m.sync()
    for r, b in zip(one, d_out):
        assert np.isfinite(r).all() and np.array_equal(r, b.numpy())


def test_two_stream_pipelining_is_bit_identical():
    """msiren_set_streams(2): consecutive async calls alternate streams; results must not change."""
    from mri_inr_amd import _lib

    sd = syn.make_state_dict(seed=7, trained_like=True)
    for prec in ("fp32", "f16x3"):
        m = make_model(sd, precision=prec)
        tiles = [np.random.default_rng(s).random((37 + 11 * s, 32, 32), dtype=np.float32) for s in range(6)]
        ref = [m(t) for t in tiles]
        img = syn.make_slice(5, brain_mask=True)
        ref_img = m.reconstruct(img)
        d_in = [m.device_array(t.shape).copy_from(t) for t in tiles]
        d_out = [m.device_array((t.shape[0], 24, 24)) for t in tiles]
        for streams in (2, 3):   # (3: round 5 -- the rotation, the cut of a large call and host calls from any position of it)
            _lib.check(m._lib.msiren_set_streams(m._h, streams))
            for _ in range(3):
                for a, b, t in zip(d_in, d_out, tiles):
                    _lib.check(m._lib.msiren_forward_tiles_dev(m._h, a.ptr, t.shape[0], b.ptr))
            m.sync()
            for r, b in zip(ref, d_out):
                assert np.array_equal(r, b.numpy())
            # host-pointer entry points stay self-contained (copies and kernels on one stream) in this mode
            for k in range(4):
                assert np.array_equal(m.reconstruct(img), ref_img)
                assert np.array_equal(m(tiles[0]), ref[0])
                _lib.check(m._lib.msiren_forward_tiles_dev(m._h, d_in[k].ptr, tiles[k].shape[0], d_out[k].ptr))   # (moves the rotation on)
            m.sync()
        assert m._lib.msiren_set_streams(m._h, 4) != 0 and m._lib.msiren_set_streams(m._h, 0) != 0
        _lib.check(m._lib.msiren_set_streams(m._h, 1))
        # a large synchronous host call cuts itself in two halves over the two streams (copy overlap):
        # same bits as the whole batch through the device entry point
        big = np.random.default_rng(9).random((401, 32, 32), dtype=np.float32)
        d_big, d_res = m.device_array(big.shape).copy_from(big), m.device_array((401, 24, 24))
        _lib.check(m._lib.msiren_forward_tiles_dev(m._h, d_big.ptr, 401, d_res.ptr))
        m.sync()
        for _ in range(2):
            assert np.array_equal(m(big), d_res.numpy())


@pytest.mark.parametrize("H,L,Z,act", [(512, 10, 128, "sine"), (256, 4, 64, "morlet")])
def test_residual_variant_vs_own_oracle(H, L, Z, act):
    """BASELINE config 5 shape (10 x 512, latent 128) with the build-defined residual semantics
    x_{l+1} = x_l + mod_l * act(W_l x_l + b_l), l >= 1.  PARITY UNPINNED against the reference (its
    residual branch is not in the container): checked against this build's own oracle only."""
    sd = syn.make_state_dict(seed=21, dim_hidden=H, num_layers=L, latent_dim=Z, siren_patch_size=24, with_encoder=False)
    sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
    m = ModulatedSiren(dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=Z, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda", activation=act,
                       residual=True)
    m.load_state_dict(sd, strict=False)
    m.to("cuda")
    mods = syn.make_mods(8, L, 6, H, lo=0.1, hi=0.6)
    out = m.forward_mods(mods)
    ref = orc.siren_forward(sd, mods, num_layers=L, activation=act, residual=True, dtype=np.float64)
    ref32 = orc.siren_forward(sd, mods, num_layers=L, activation=act, residual=True)
    e = nerr(out.reshape(6, -1), ref)
    assert e <= max(1e-4, 10 * nerr(ref32, ref)), (e, nerr(ref32, ref))


@pytest.mark.parametrize("prec,res,act,tol", [("bf16", True, "sine", 6e-2), ("bf16", False, "sine", 6e-2),
                                               ("f16", True, "sine", 8e-3), ("f16", False, "morlet", 8e-3)])
def test_config5_16bit_trunk_vs_own_oracle(prec, res, act, tol):
    """BASELINE config 5: 10 x 512, latent 128, 16-bit operands / fp32 accumulate, register-resident trunk.
    PARITY UNPINNED against the reference (residual branch not in the container); the tolerance is the
    operand format's: bf16 has an 8-bit significand (2^-9 per rounding, ~10 roundings deep), fp16 11 bits."""
    H, L, Z, B = 512, 10, 128, 9
    sd = syn.make_state_dict(seed=21, dim_hidden=H, num_layers=L, latent_dim=Z, with_encoder=False)
    sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
    m = ModulatedSiren(dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=Z, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda", activation=act,
                       residual=res, precision=prec)
    m.load_state_dict(sd, strict=False)
    m.to("cuda")
    mods = syn.make_mods(8, L, B, H, lo=0.1, hi=0.6)
    out = m.forward_mods(mods)
    ref = orc.siren_forward(sd, mods, num_layers=L, activation=act, residual=res, dtype=np.float64)
    e = nerr(out.reshape(B, -1), ref)
    assert np.isfinite(out).all() and e <= tol, e
    assert np.array_equal(out, m.forward_mods(mods))
    # the fp32 trunk on the same model is the high-precision cross-check of the 16-bit one
    m32 = ModulatedSiren(dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=Z, w0=1.0, w0_initial=30.0,
                         use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                         outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda", activation=act,
                         residual=res)
    m32.load_state_dict(sd, strict=False)
    m32.to("cuda")
    assert nerr(m32.forward_mods(mods).reshape(B, -1), ref) < 1e-4


@pytest.mark.parametrize("res,L,mod", [(True, 10, 3e5), (True, 10, 1.0), (False, 4, 1e6), (True, 2, 2e5)])
def test_f16_single_product_domain_identical_to_fp32_outside_it(res, L, mod):
    """MSIREN_PREC_F16 (H = 512, fp16 operands, one product): activations, modulations and residual sums are rounded to fp16, and
    beyond 65 504 that is inf -- NaN one sine later -- where the reference's fp32 arithmetic stays finite.  A launch that stores a
    non-finite output raises its stream's flag, and the exact-fp32 trunk, enqueued behind it as a conditional launch, redoes
    the batch: outside the domain the buffer holds the exact-fp32 trunk's bits, inside it the fp16 kernel's own (both the
    weight-stationary kernel and, at num_layers = 2, the register-resident one), on the synchronous and the asynchronous API."""
    H, Z, B = 512, 128, 11
    sd = syn.make_state_dict(seed=27, dim_hidden=H, num_layers=L, latent_dim=Z, with_encoder=False)
    sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
    kw = dict(dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=Z, w0=1.0, w0_initial=30.0, use_bias=True, dropout=0.1,
              modulate=True, encoder_type="custom", encoder_path=None, outer_patch_size=32, inner_patch_size=16, siren_patch_size=24,
              device="cuda", activation="sine", residual=res)
    m, f = ModulatedSiren(**kw, precision="f16"), ModulatedSiren(**kw)
    for mm in (m, f):
        mm.load_state_dict(sd, strict=False)
        mm.to("cuda")
    small = syn.make_mods(8, L, B, H, lo=0.1, hi=0.6)
    mods = (small * np.float32(mod)).astype(np.float32)
    want = f.forward_mods(mods)
    assert np.isfinite(want).all()
    got = m.forward_mods(mods)
    inside = m.forward_mods(small)
    ref_small = orc.siren_forward(sd, small, num_layers=L, residual=res, dtype=np.float64)
    assert np.isfinite(inside).all() and nerr(inside.reshape(B, -1), ref_small) <= 8e-3
    if mod > 100:
        assert np.array_equal(got, want)          # repaired: bit for bit the exact-fp32 trunk
    else:
        assert np.array_equal(got, inside)        # inside the domain nothing is redone
        assert not np.array_equal(got, want)
    # asynchronous entry point, flagged and clean launches interleaved on both streams
    _lib_ = __import__("mri_inr_amd")._lib
    _lib_.check(m._lib.msiren_set_streams(m._h, 2))
    d_big, d_small = m.device_array(mods.shape).copy_from(mods), m.device_array(small.shape).copy_from(small)
    outs = [m.device_array((B, 24, 24)) for _ in range(6)]
    for k in range(6):
        _lib_.check(m._lib.msiren_forward_mods_dev(m._h, (d_big if k % 2 == 0 else d_small).ptr, B, outs[k].ptr))
    m.sync()
    for k in range(6):
        assert np.array_equal(outs[k].numpy(), got if k % 2 == 0 else inside), k


@pytest.mark.parametrize("L,B", [(2, 5), (3, 5), (11, 3)])
def test_16bit_trunk_depth_range_vs_own_oracle(L, B):
    """The single-product trunk at its depth limits: num_layers = 2 runs the register-resident kernel (the weight-stationary
    layer pipeline needs a hidden layer in front of the final one), 3 the shortest weight-stationary pipeline, 11 the deepest
    model whose tables fit the 160 KB LDS; 12 must be refused at load time, not at the first launch."""
    H, Z = 512, 128

    def build(depth):
        sd = syn.make_state_dict(seed=23, dim_hidden=H, num_layers=depth, latent_dim=Z, with_encoder=False)
        sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
        m = ModulatedSiren(dim_in=2, dim_hidden=H, dim_out=1, num_layers=depth, latent_dim=Z, w0=1.0, w0_initial=30.0,
                           use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                           outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda", activation="sine",
                           residual=True, precision="bf16")
        m.load_state_dict(sd, strict=False)
        m.to("cuda")
        return sd, m

    sd, m = build(L)
    mods = syn.make_mods(9, L, B, H, lo=0.1, hi=0.6)
    out = m.forward_mods(mods)
    assert m.last_trunk_kernel().startswith("siren_trunk_x1n_kernel" if L == 2 else "siren_trunk_x1w_kernel"), m.last_trunk_kernel()
    ref = orc.siren_forward(sd, mods, num_layers=L, activation="sine", residual=True, dtype=np.float64)
    e = nerr(out.reshape(B, -1), ref)
    assert np.isfinite(out).all() and e <= 6e-2, e
    if L == 11:
        with pytest.raises(ValueError, match="num_layers"):   # MSIREN_E_INVALID, with the library's message
            build(12)


def test_config5_batch_invariance_over_pass_shapes():
    """A tile's output must not depend on the batch it came in (modulated_siren.py:435-457: patches are independent) -- bit for
    bit, whatever mix of 4-unit passes, 2-unit tail passes (x1w_schedule) and grid (x1w_balanced_grid) the launch is made of:
    1 tile = 18 units (4 + 1 passes), 57 tiles = 1026 units (one full round + a single 2-unit pass), 58 = 1044 (+ 10 of them),
    129 = 2322 units (two full rounds + 137 tail passes), 460 tiles = 8 rounds + a tail; one and two streams."""
    H, L, Z, B = 512, 10, 128, 460
    sd = syn.make_state_dict(seed=21, dim_hidden=H, num_layers=L, latent_dim=Z, with_encoder=False)
    sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
    m = ModulatedSiren(dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=Z, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda", activation="sine",
                       residual=True, precision="bf16")
    m.load_state_dict(sd, strict=False)
    m.to("cuda")
    mods = syn.make_mods(31, L, B, H, lo=0.1, hi=0.6)
    whole = m.forward_mods(mods)
    assert m.last_trunk_kernel().startswith("siren_trunk_x1w_kernel") and np.isfinite(whole).all()
    ref = orc.siren_forward(sd, mods[:, :3], num_layers=L, activation="sine", residual=True, dtype=np.float64)
    assert nerr(whole[:3].reshape(3, -1), ref) <= 6e-2
    from mri_inr_amd import _lib
    for streams in (1, 2, 3):
        _lib.check(m._lib.msiren_set_streams(m._h, streams))
        for b in (1, 2, 7, 57, 58, 129, 256, 257):
            part = m.forward_mods(np.ascontiguousarray(mods[:, :b]))
            assert np.array_equal(part, whole[:b]), (streams, b)
            tail = m.forward_mods(np.ascontiguousarray(mods[:, B - b:]))
            assert np.array_equal(tail, whole[B - b:]), (streams, b)


def test_default_precision_is_the_fast_exact_trunk():
    """precision="auto" (the default) selects the f16x3 trunk where supported and must agree with the
    explicit choices bit for bit; unsupported shapes silently use the fp32 trunk."""
    sd = syn.make_state_dict(seed=7)
    mods = syn.make_mods(3, 5, 12, 256)
    a = make_model(sd).forward_mods(mods)
    assert np.array_equal(a, make_model(sd, precision="f16x3").forward_mods(mods))
    sd2 = syn.make_state_dict(seed=7, dim_hidden=128, num_layers=3, with_encoder=False)
    sd2 = {k: v for k, v in sd2.items() if not k.startswith("modulator")}
    outs = []
    for prec in ("auto", "fp32"):
        m = ModulatedSiren(dim_in=2, dim_hidden=128, dim_out=1, num_layers=3, latent_dim=256, w0=1.0, w0_initial=30.0,
                           use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                           outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda",
                           activation="sine", precision=prec)
        m.load_state_dict(sd2, strict=False)
        m.to("cuda")
        outs.append(m.forward_mods(syn.make_mods(3, 3, 4, 128)))
    assert np.array_equal(outs[0], outs[1])


def test_tiling_kernels_vs_reference_fixtures():
    """image_to_patches / weighted fold on the device against outputs of the reference's tiling.py."""
    import ctypes as C

    from mri_inr_amd import _lib

    g = load_golden("tiling.npz")
    sd = syn.make_state_dict(seed=7)
    m = make_model(sd)
    for name, (hh, ww) in (("320x320", (320, 320)), ("70x50", (70, 50))):
        img = syn.make_slice(3, hh, ww, brain_mask=(name == "320x320"))
        nv, nh = C.c_int32(), C.c_int32()
        _lib.check(m._lib.msiren_recon_shape(m._h, hh, ww, C.byref(nv), C.byref(nh)))
        assert (nv.value, nh.value) == tuple(g[f"info_{name}"])
        n = nv.value * nh.value
        d_img = m.device_array((1, hh, ww)).copy_from(img[None])
        d_p = m.device_array((n, 32, 32))
        _lib.check(m._lib.msiren_image_to_patches_dev(m._h, d_img.ptr, 1, hh, ww, d_p.ptr))
        m.sync()
        assert np.array_equal(d_p.numpy(), g[f"patches_{name}"])  # byte moving: bit-exact
        rec = np.random.default_rng(5).random((n, 24, 24), dtype=np.float32)
        d_r = m.device_array(rec.shape).copy_from(rec)
        d_o = m.device_array((1, nv.value * 16, nh.value * 16))
        _lib.check(m._lib.msiren_weighted_fold_dev(m._h, d_r.ptr, 1, nv.value, nh.value, d_o.ptr))
        m.sync()
        assert nerr(d_o.numpy()[0], g[f"wfold_{name}"][0]) < 1e-6
    # reflect padding needs pad < dim, as F.pad does
    d_small = m.device_array((1, 6, 6))
    d_ps = m.device_array((1, 32, 32))
    with pytest.raises(ValueError):
        _lib.check(m._lib.msiren_image_to_patches_dev(m._h, d_small.ptr, 1, 6, 6, d_ps.ptr))


def test_black_patch_semantics_match_reference():
    """Zero tiles are skipped by the reference and re-inserted as zeros WITH their fold weight
    (tiling.py:287-301, :117-118): neighbours are pulled down.  Known answer from SURVEY.md §8(f)."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = make_model(sd)
    img = syn.make_slice(0, 160, 128, brain_mask=True)
    rec = m.reconstruct(img)
    ref = orc.reconstruct_slice(sd, img, num_layers=5, dtype=np.float64)
    check(rec, ref.astype(np.float32), tol=1e-4)
    assert np.all(rec[:8, :8] == 0.0)  # the corner patch is black: exactly zero there


def test_deep_model_falls_back_to_fp32_trunk():
    L = 14  # beyond the f16x3 kernel's LDS budget: the library must use the fp32 trunk silently
    sd = syn.make_state_dict(seed=2, num_layers=L, with_encoder=False)
    sd = {k: v for k, v in sd.items() if not k.startswith("modulator")}
    m = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=L, latent_dim=256, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda", activation="sine")
    m.load_state_dict(sd, strict=False)
    m.to("cuda")
    mods = syn.make_mods(5, L, 3, 256, lo=0.3, hi=0.9)
    ref = orc.siren_forward(sd, mods, num_layers=L, dtype=np.float64)
    e32 = nerr(orc.siren_forward(sd, mods, num_layers=L), ref)
    e = nerr(m.forward_mods(mods).reshape(3, -1), ref)
    assert e <= max(1e-4, 10 * e32), (e, e32)


def test_rccl_broadcast_then_forward_single_rank():
    """The N > 1 bench path with the real backend: init RCCL ("nccl"), broadcast the weight blob through a
    device tensor, load it into the HIP library of the same process and evaluate.  world_size 1 (one
    card here); the 2-rank logic is covered on gloo in test_dist_gloo.py."""
    import subprocess
    import sys
    import textwrap

    script = textwrap.dedent("""
        import os, sys, numpy as np
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
        import torch, torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from mri_inr_amd import ModulatedSiren, synthetic as syn
        from mri_inr_amd.dist import broadcast_state_dict
        from oracle import siren_oracle as orc
        sd0 = syn.make_state_dict(seed=3, trained_like=True)
        sd = broadcast_state_dict(sd0, src=0, device=torch.device("cuda", 0))
        assert all(np.array_equal(sd[k], sd0[k]) for k in sd0)
        m = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=5, latent_dim=256, w0=1.0, w0_initial=30.0,
                           use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                           outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda:0",
                           activation="sine")
        m.load_state_dict(sd); m.to("cuda:0").eval()
        tiles = np.random.default_rng(0).random((16, 32, 32), dtype=np.float32)
        out = np.asarray(m(tiles))
        ref = orc.modulated_siren_forward(sd, tiles, num_layers=5, dtype=np.float64)
        err = float(np.abs(out - ref).max() / np.abs(ref).max())
        t = torch.tensor([err], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
        dist.destroy_process_group()
        print("NERR", float(t.item()))
        sys.exit(0 if float(t.item()) < 1e-4 else 1)
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", script], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "NERR" in r.stdout


def test_integration_md_ctypes_stub_runs_as_written():
    """INTEGRATION.md §2 is the binding a maintainer would paste: execute that code block verbatim in a fresh
    interpreter (no torch, no mri_inr_amd package -- only ctypes + numpy + libmsiren.so) and check its output."""
    import re
    import subprocess
    import sys
    import textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"## 2\..*?```python\n(.*?)```", doc, re.S).group(1)
    assert "msiren_forward_tiles" in block and "msiren_create" in block
    prologue = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        import importlib.util
        spec = importlib.util.spec_from_file_location("syn", %r)   # synthetic weights only (pure numpy)
        syn = importlib.util.module_from_spec(spec); spec.loader.exec_module(syn)
        class _T:                                   # the two tensor methods the stub uses
            def __init__(self, a): self.a = a
            def cpu(self): return self
            def numpy(self): return self.a
        sd = syn.make_state_dict(seed=7, trained_like=True)
        state_dict = {k: _T(v) for k, v in sd.items()}
        tiles = np.random.default_rng(5).random((9, 32, 32), dtype=np.float32)
    """) % (root, os.path.join(root, "mri_inr_amd", "synthetic.py"))
    epilogue = textwrap.dedent("""
        assert "torch" not in sys.modules and "mri_inr_amd" not in sys.modules
        from oracle import siren_oracle as orc
        ref = orc.modulated_siren_forward(sd, tiles, num_layers=5, dtype=np.float64)
        err = float(np.abs(out - ref).max() / np.abs(ref).max())
        print("NERR", err)
        sys.exit(0 if err < 1e-4 else 1)
    """)
    r = subprocess.run([sys.executable, "-c", prologue + block + epilogue], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "NERR" in r.stdout


def test_tiling_kernels_ragged_sizes_vs_oracle():
    """image_to_patches / weighted fold / whole-slice pipeline for image sizes that are not multiples of the
    stride, several slices per call, down to the smallest size reflect padding accepts -- against the oracle
    (itself pinned to the reference's tiling.py by tests/golden/tiling.npz).  Byte moving is bit-exact."""
    import ctypes as C

    from mri_inr_amd import _lib

    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = make_model(sd)
    rng = np.random.default_rng(21)
    for n, hh, ww in ((1, 16, 25), (2, 33, 47), (3, 64, 48), (1, 100, 321), (2, 319, 35), (1, 24, 400)):
        imgs = rng.random((n, hh, ww), dtype=np.float32)
        imgs[:, : hh // 3, : ww // 3] = 0.0  # some black tiles
        nv, nh = C.c_int32(), C.c_int32()
        _lib.check(m._lib.msiren_recon_shape(m._h, hh, ww, C.byref(nv), C.byref(nh)))
        ref_p, info = zip(*(orc.image_to_patches(im, 32, 16) for im in imgs))
        assert (nv.value, nh.value) == tuple(np.ravel(info[0]))
        per = nv.value * nh.value
        d_img = m.device_array(imgs.shape).copy_from(imgs)
        d_p = m.device_array((n * per, 32, 32))
        _lib.check(m._lib.msiren_image_to_patches_dev(m._h, d_img.ptr, n, hh, ww, d_p.ptr))
        m.sync()
        assert np.array_equal(d_p.numpy(), np.concatenate(ref_p, 0)), (n, hh, ww)
        rec = rng.random((n * per, 24, 24), dtype=np.float32)
        d_r = m.device_array(rec.shape).copy_from(rec)
        d_o = m.device_array((n, nv.value * 16, nh.value * 16))
        _lib.check(m._lib.msiren_weighted_fold_dev(m._h, d_r.ptr, n, nv.value, nh.value, d_o.ptr))
        m.sync()
        ref_o = np.stack([orc.patches_to_image_weighted_average(rec[k * per:(k + 1) * per], info[k], 24, 16) for k in range(n)])
        assert nerr(d_o.numpy(), ref_o) < 2e-6, (n, hh, ww)
        # the whole pipeline, black tiles included
        got = m.reconstruct(imgs)
        ref = np.stack([orc.reconstruct_slice(sd, im, num_layers=5, dtype=np.float64) for im in imgs])
        assert got.shape == ref.shape and nerr(got, ref) < 1e-4, (n, hh, ww, nerr(got, ref))
    # too small for reflect padding (F.pad raises in the reference): refused, not mis-tiled
    for bad in ((8, 40), (17, 17)):
        with pytest.raises((ValueError, RuntimeError)):
            m.reconstruct(np.ones(bad, np.float32))


def test_two_handles_from_two_threads():
    """thread-compatible, re-entrant across handles.  Two models with different weights (and
    different trunks: f16x3 work-queue kernel / fp32 kernel) are driven concurrently from two host threads
    (ctypes drops the GIL inside the library); every result equals the model's own single-threaded output."""
    import threading

    models, tiles, refs = [], [], []
    for seed, prec in ((7, "f16x3"), (8, "fp32")):
        sd = syn.make_state_dict(seed=seed, trained_like=True)
        m = make_model(sd, precision=prec)
        t = np.random.default_rng(seed).random((300 + seed, 32, 32), dtype=np.float32)
        models.append(m)
        tiles.append(t)
        refs.append(m(t))
    errors = []

    def work(i):
        try:
            for _ in range(15):
                out = models[i](tiles[i])
                if not np.array_equal(out, refs[i]):
                    errors.append((i, float(np.abs(out - refs[i]).max())))
                img = models[i].reconstruct(syn.make_slice(i, brain_mask=True))
                if img.shape != (320, 320) or not np.isfinite(img).all():
                    errors.append((i, "reconstruct"))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors[:5]


def test_two_threads_share_one_pageable_input_array():
    """Two handles in two threads are handed THE SAME pageable input array (and slices of it that lie inside the other thread's range).
    Inputs are only read: the library leaves the caller's memory as it is (no registration, no page-locking -- include/msiren.h), each call
    copies its tiles through the runtime, the outputs are blocks of each model's own page-locked pool written in place.  Every output
    equals the single-threaded one."""
    import threading

    sd = syn.make_state_dict(seed=7, trained_like=True)
    shared = np.random.default_rng(77).random((400, 32, 32), dtype=np.float32)
    models = [make_model(sd, precision="f16x3") for _ in range(2)]
    ref = models[0](shared)
    assert np.array_equal(models[1](shared), ref)
    errors = []

    def work(i):
        try:
            for k in range(40):
                lo = (37 * (k + i)) % 200 if k % 3 else 0          # the whole array, and slices of it that lie inside the other thread's range
                x = shared[lo:] if k % 3 else shared
                out = models[i](x)
                if not np.array_equal(out, ref[lo:]):
                    errors.append((i, k, float(np.abs(out - ref[lo:]).max())))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors[:5]
    assert _range_kind(models[0], shared) == 0          # still ordinary pageable memory: nothing of the caller's was page-locked


def _range_kind(m, a):
    """msiren_host_range_kind of a numpy array's bytes: 0 pageable, 1 inside one page-locked allocation, 2 page-locked in part."""
    import ctypes

    k = ctypes.c_int32(-1)
    _lib.check(m._lib.msiren_host_range_kind(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), ctypes.byref(k)))
    return k.value


def _forward_into(m, x, out):
    import ctypes

    fp = ctypes.POINTER(ctypes.c_float)
    assert x.flags.c_contiguous and out.flags.c_contiguous and x.dtype == out.dtype == np.float32
    _lib.check(m._lib.msiren_forward_tiles(m._h, x.ctypes.data_as(fp), x.shape[0], out.ctypes.data_as(fp)))


def test_windows_of_one_array_from_two_threads():
    """The parallel-for over slices: two handles in two threads work on WINDOWS of one input array and one output array -- adjacent ones
    (disjoint bytes that share a page at the seam) and input windows that overlap in part.  All of it pageable memory that the library only
    copies from / to.  Every output equals the single-threaded one; no call fails."""
    import threading

    sd = syn.make_state_dict(seed=7, trained_like=True)
    big_in = np.random.default_rng(78).random((1300, 32, 32), dtype=np.float32)
    models = [make_model(sd, precision="f16x3") for _ in range(2)]
    ref = models[0](big_in)
    big_out = np.zeros((1300, 24, 24), np.float32)
    errors = []

    def work(i):
        try:
            for k in range(36):
                n = 400 if k % 4 else 100
                out = big_out[400 * i:400 * i + n]                  # adjacent windows of the shared output array
                if k % 3 == 0:      # adjacent windows of the input as well
                    lo_in = 400 * i
                elif k % 3 == 1:    # input windows that overlap in part: [100, 500) and [300, 700)
                    lo_in = 100 + 200 * i
                elif i == 0:        # a wide window (an output buffer of its own) ...
                    lo_in, n = 0, 800
                    out = np.empty((n, 24, 24), np.float32)
                else:               # ... and a moving one inside it
                    lo_in = 17 * (k % 11)
                _forward_into(models[i], big_in[lo_in:lo_in + n], out)
                if not np.array_equal(out, ref[lo_in:lo_in + n]):
                    errors.append((i, k, lo_in, n, float(np.abs(out - ref[lo_in:lo_in + n]).max())))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors[:5]


def test_buffers_page_locked_in_part_go_through_a_bounce_buffer():
    """Deterministic form of the above: the CALLER page-locks rows [0, 300) of an input and of an output array (hipHostRegister), then calls on
    rows [200, 600): the first byte of either buffer is page-locked, the last is not (msiren_host_range_kind: 2).  Neither a kernel nor a
    runtime copy is safe on such a range (hipMemcpy refuses a range that leaves the registration it begins in): it goes through a bounce
    buffer.  Same bits as on untouched arrays; fully page-locked and fully pageable windows of the same arrays work as ever.  Then the
    case both of whose ENDS are page-locked while the middle is not (two registrations, pageable rows in between): the device address of
    page-locked memory equals its host address here, so the ends alone prove nothing -- the allocation that holds the first byte must
    hold the last (hipPointerGetAttribute RANGE_START_ADDR / RANGE_SIZE); a kernel let loose on that range would fault on the pageable rows."""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = make_model(sd, precision="f16x3")
    x = np.random.default_rng(79).random((700, 32, 32), dtype=np.float32)
    ref = m(x.copy())
    out = np.zeros((700, 24, 24), np.float32)
    assert hip.hipHostRegister(ctypes.c_void_p(x.ctypes.data), ctypes.c_size_t(300 * 32 * 32 * 4), ctypes.c_uint(0)) == 0
    assert hip.hipHostRegister(ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(300 * 24 * 24 * 4), ctypes.c_uint(0)) == 0
    try:
        assert [_range_kind(m, x[a:b]) for a, b in ((200, 600), (0, 300), (50, 250), (300, 700), (0, 700))] == [2, 1, 1, 0, 2]
        assert _range_kind(m, out[0:300]) == 1 and _range_kind(m, out[100:301]) == 2 and _range_kind(m, out[300:]) == 0
        pool_block = m.pinned_empty((64, 32, 32))
        assert _range_kind(m, pool_block) == 1 and _range_kind(m, pool_block[3:40]) == 1       # (what the in-place default path rests on)
        del pool_block
        for lo, hi in ((200, 600), (0, 300), (300, 700), (250, 314), (0, 700)):
            out[:] = 0
            _forward_into(m, x[lo:hi], out[lo:hi])
            assert np.array_equal(out[lo:hi], ref[lo:hi]), (lo, hi)
            assert not out[:lo].any() and not out[hi:].any()
        # input in part, output pageable; and the other way round
        o2 = np.empty((400, 24, 24), np.float32)
        _forward_into(m, x[200:600], o2)
        assert np.array_equal(o2, ref[200:600])
        x2 = x[200:600].copy()
        out[:] = 0
        _forward_into(m, x2, out[200:600])
        assert np.array_equal(out[200:600], ref[200:600])
        # the other synchronous entry points copy through the runtime: the same rule (HostSrc / HostDst)
        z = m.encoder(x2)
        assert np.array_equal(m.encoder(x[200:600]), z)
        d = m.device_array((400, 32, 32)).copy_from(x[200:600])
        assert np.array_equal(d.numpy(), x2)
    finally:
        assert hip.hipHostUnregister(ctypes.c_void_p(x.ctypes.data)) == 0
        assert hip.hipHostUnregister(ctypes.c_void_p(out.ctypes.data)) == 0
    out[:] = 0
    _forward_into(m, x[200:600], out[200:600])
    assert np.array_equal(out[200:600], ref[200:600])
    # two registrations with pageable rows between them: rows [0, 100) and [200, 300) page-locked, the call on rows [50, 250)
    row = 32 * 32 * 4
    for lo in (0, 200):
        assert hip.hipHostRegister(ctypes.c_void_p(x.ctypes.data + lo * row), ctypes.c_size_t(100 * row), ctypes.c_uint(0)) == 0
    try:
        assert _range_kind(m, x[50:250]) == 2 and _range_kind(m, x[0:100]) == 1 and _range_kind(m, x[200:300]) == 1 and _range_kind(m, x[100:200]) == 0
        o3 = m.pinned_empty((200, 24, 24))
        _forward_into(m, x[50:250], o3)
        assert np.array_equal(o3, ref[50:250])
    finally:
        for lo in (0, 200):
            assert hip.hipHostUnregister(ctypes.c_void_p(x.ctypes.data + lo * row)) == 0


@pytest.mark.parametrize("start", ["0xFFFFF800", "0x7FFFF800"])
def test_pass_counter_wraps_safely(start):
    """The persistent trunks' pass counter is never reset (the host tells each launch the value it will find), so
    after ~2^32 passes (minutes of continuous use) it wraps.  Started just below 2^32 / 2^31 (test knob
    MSIREN_QUEUE_START, read when a handle's queue is created) the next launches cross the boundary: results must
    not change, and nothing may hang."""
    import subprocess
    import sys
    import textwrap

    script = textwrap.dedent("""
        import numpy as np
        from mri_inr_amd import ModulatedSiren, synthetic as syn
        from oracle import siren_oracle as orc
        sd = syn.make_state_dict(seed=7, trained_like=True)
        m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine")
        m.load_state_dict(sd); m.to("cuda")
        tiles = np.random.default_rng(2).random((400, 32, 32), dtype=np.float32)
        ref = orc.modulated_siren_forward(sd, tiles[:32], num_layers=5, dtype=np.float64)
        first = m(tiles)                       # 2 x 900 passes per call: the counter wraps within two calls
        assert np.abs(first[:32] - ref).max() / np.abs(ref).max() < 1e-4
        for _ in range(6):
            assert np.array_equal(m(tiles), first)
        print("WRAP OK")
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MSIREN_QUEUE_START=start)
    r = subprocess.run([sys.executable, "-c", script], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "WRAP OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("inner,S", [(8, 16), (16, 32), (32, 32), (16, 20)])
def test_slice_pipeline_other_tiling_geometries(inner, S):
    """The YAML surface lets inner_patch_size / siren_patch_size vary (outer stays 32: the custom encoder is
    hard-wired to 32x32 tiles).  Tiling, black filter, forward and weighted fold against the oracle."""
    sd = syn.make_state_dict(seed=5, siren_patch_size=S, trained_like=True)
    m = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=5, latent_dim=256, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=inner, siren_patch_size=S, device="cuda", activation="sine")
    m.load_state_dict(sd)
    m.to("cuda").eval()
    img = syn.make_slice(2, 96, 80, brain_mask=True)
    got = m.reconstruct(img)
    ref = orc.reconstruct_slice(sd, img, num_layers=5, outer=32, inner=inner, siren_patch_size=S, dtype=np.float64)
    assert got.shape == ref.shape
    assert nerr(got, ref) < 1e-4, nerr(got, ref)
    # the fold kernel alone against the REFERENCE's output for this geometry (tests/golden/tiling_geometries.npz)
    import ctypes as C

    from mri_inr_amd import _lib

    g = load_golden("tiling_geometries.npz")
    nv, nh = (int(v) for v in g[f"info_{inner}_{S}"])
    rec = np.random.default_rng(6).random((nv * nh, S, S), dtype=np.float32)
    d_r = m.device_array(rec.shape).copy_from(rec)
    d_o = m.device_array((1, nv * inner, nh * inner))
    _lib.check(m._lib.msiren_weighted_fold_dev(m._h, d_r.ptr, 1, nv, nh, d_o.ptr))
    m.sync()
    assert nerr(d_o.numpy()[0], g[f"wfold_{inner}_{S}"]) < 2e-6


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("name", ["w0", "nobias", "small", "morlet_w0", "deep"])
def test_model_variants_vs_reference_fixtures(name, prec):
    """The HIP path on hyper-parameters off the YAML defaults, against outputs of the reference itself
    (tests/golden/model_variants.npz): trunk on seeded modulations and full forward on seeded tiles.  Both trunk
    arithmetics ("small" is 3 x 128: the library serves it with the fp32 trunk whatever is asked)."""
    from test_oracle_golden import _variant

    g = load_golden("model_variants.npz")
    v, sd, mods, tiles = _variant(name, g)
    m = ModulatedSiren(dim_in=2, dim_hidden=v["H"], dim_out=1, num_layers=v["L"], latent_dim=v["Z"], w0=v["w0"],
                       w0_initial=v["w0_initial"], use_bias=v["use_bias"], dropout=0.1, modulate=True,
                       encoder_type="custom", encoder_path=None, outer_patch_size=32, inner_patch_size=16,
                       siren_patch_size=v["S"], device="cuda", activation=v["activation"], precision=prec)
    m.load_state_dict(sd)
    m.to("cuda").eval()
    kw = dict(num_layers=v["L"], w0=v["w0"], w0_initial=v["w0_initial"], activation=v["activation"],
              siren_patch_size=v["S"], dtype=np.float64)
    for got, ref, truth in ((m.forward_mods(mods).reshape(6, -1), g[f"{name}_trunk"], orc.siren_forward(sd, mods, **kw)),
                            (m(tiles), g[f"{name}_forward"], orc.modulated_siren_forward(sd, tiles, **kw))):
        assert got.shape == ref.shape and got.dtype == np.float32 and np.isfinite(got).all()
        # The north star's tolerance against the reference's output -- unless the reference's own fp32 arithmetic is
        # further than that from the fp64 result (w0 = 2 doubles every hidden sine argument: torch-fp32 is 7-8e-5
        # away from fp64 there, 3e-6 on the other variants; two such results may differ by the sum) ...
        floor = nerr(ref, truth)
        assert nerr(got, ref) <= max(1e-4, 2.0 * floor), (nerr(got, ref), floor)
        # ... and at least about as close to the fp64 result as the reference is
        assert nerr(got, truth) <= max(2e-5, 1.5 * floor), (nerr(got, truth), floor)
        assert rms(got, ref) <= max(1e-5, 2.0 * rms(ref, truth)), (rms(got, ref), rms(ref, truth))


def test_tiling_roundtrip_property_on_device():
    """Size-independent property through the device kernels: tiles cut from one image by image_to_patches_dev are
    mutually consistent, so the weighted fold of their 24x24 centres returns the image on [0, H) x [0, W) -- at the
    BASELINE slice size, several slices per call, and ragged sizes."""
    import ctypes as C

    from mri_inr_amd import _lib

    sd = syn.make_state_dict(seed=7)
    m = make_model(sd)
    rng = np.random.default_rng(3)
    for n, hh, ww in ((1, 320, 320), (5, 320, 320), (2, 200, 136), (3, 37, 81)):
        imgs = rng.random((n, hh, ww), dtype=np.float32)
        nv, nh = C.c_int32(), C.c_int32()
        _lib.check(m._lib.msiren_recon_shape(m._h, hh, ww, C.byref(nv), C.byref(nh)))
        per = nv.value * nh.value
        d_img = m.device_array(imgs.shape).copy_from(imgs)
        d_p = m.device_array((n * per, 32, 32))
        _lib.check(m._lib.msiren_image_to_patches_dev(m._h, d_img.ptr, n, hh, ww, d_p.ptr))
        m.sync()
        centres = np.ascontiguousarray(d_p.numpy()[:, 4:28, 4:28])
        d_c = m.device_array(centres.shape).copy_from(centres)
        d_o = m.device_array((n, nv.value * 16, nh.value * 16))
        _lib.check(m._lib.msiren_weighted_fold_dev(m._h, d_c.ptr, n, nv.value, nh.value, d_o.ptr))
        m.sync()
        out = d_o.numpy()
        assert np.abs(out[:, :hh, :ww] - imgs).max() < 1e-6, (n, hh, ww)


def test_submodules_encoder_modulator_net_like_the_reference():
    """model.encoder(tiles), model.modulator(z), model.net(coords, mods): the reference's sub-modules (modulated_siren.py:404-425)
    against the oracle, and forward == net(grid, modulator(encoder(tiles))) bit for bit."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = make_model(sd)
    tiles = np.random.default_rng(12).random((70, 32, 32), dtype=np.float32)
    z = m.encoder(tiles)
    assert z.shape == (70, 256) and z.dtype == np.float32
    assert nerr(z, orc.encoder_forward(sd, tiles, dtype=np.float64)) < 1e-5
    mods = m.modulator(z)
    assert isinstance(mods, tuple) and len(mods) == 5 and all(t.shape == (70, 256) for t in mods)
    ref_mods = orc.modulator_forward(sd, z, num_layers=5, dtype=np.float64)
    assert nerr(np.stack(mods), ref_mods) < 1e-5 and all((t >= 0).all() for t in mods)
    coords = np.broadcast_to(np.asarray(m.grid, np.float32), (70,) + np.asarray(m.grid).shape)
    out = m.net(coords, mods)
    assert out.shape == (70, 576, 1)
    assert np.array_equal(out.reshape(70, 24, 24), m(tiles))      # the same three steps inside forward
    assert np.array_equal(m.net(None, mods), out)
    with pytest.raises(ValueError):
        m.net(coords * 0.5, mods)
    # small batch (fused per-tile encoder) and the empty batch
    z1 = m.encoder(tiles[:3])
    assert nerr(z1, orc.encoder_forward(sd, tiles[:3], dtype=np.float64)) < 1e-5
    assert m.encoder(tiles[:0]).shape == (0, 256)
    assert all(t.shape == (0, 256) for t in m.modulator(np.zeros((0, 256), np.float32)))
    torch = pytest.importorskip("torch")
    zt = m.encoder(torch.from_numpy(tiles))
    assert isinstance(zt, torch.Tensor) and np.array_equal(zt.numpy(), z)
