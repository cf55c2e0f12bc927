"""Large asynchronous calls overlap with themselves (mri_inr_amd/csrc/msiren.hip: forward_tiles_split), and the paths the
benchmark's timed region runs are checked against the reference's fixtures DIRECTLY (not through another path's output).

With MSIREN_SPLIT_MIN=3200 (round 4's default; off since round 5: with the one-launch prologue the uncut call is as fast or faster,
profiles/r5/07_*) a call of >= 3200 tiles on a one-stream handle is cut in two: the encoder + Modulator of most of the batch run on the handle's other stream
beside the register-resident trunk of the first part, the weight-stationary trunk of the rest follows.  Patches are
independent (modulated_siren.py:435-457) and the two trunks give the same bits, so nothing may change.
"""
import os

import numpy as np
import pytest

from conftest import load_golden
from mri_inr_amd import _lib, synthetic as syn
from oracle import siren_oracle as orc
from test_gpu_parity import check, make_model

pytestmark = pytest.mark.gpu


def make_with_env(sd, env, **kw):
    """The knobs are read once, at msiren_create."""
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


def run_dev(m, d_in, n, d_out):
    _lib.check(m._lib.msiren_forward_tiles_dev(m._h, d_in.ptr, n, d_out.ptr))


@pytest.mark.parametrize("act", ["sine", "morlet"])
def test_split_call_same_bits_as_uncut_call(act):
    sd = syn.make_state_dict(seed=7, trained_like=True)
    B = 3300
    tiles = np.random.default_rng(11).random((B, 32, 32), dtype=np.float32)
    whole = make_with_env(sd, {"MSIREN_SPLIT_MIN": 0}, act=act, precision="f16x3")
    d_in, d_out = whole.device_array(tiles.shape).copy_from(tiles), whole.device_array((B, 24, 24))
    run_dev(whole, d_in, B, d_out)
    whole.sync()
    ref = d_out.numpy()
    assert whole.last_trunk_kernel().startswith("siren_trunk_f16x3w_kernel")
    check(ref[:32], orc.modulated_siren_forward(sd, tiles[:32], num_layers=5, activation=act, dtype=np.float64))
    check(ref[-32:], orc.modulated_siren_forward(sd, tiles[-32:], num_layers=5, activation=act, dtype=np.float64))
    a = 1 if act == "morlet" else 0
    for env in ({"MSIREN_SPLIT_MIN": 3200}, {"MSIREN_SPLIT_MIN": 3200, "MSIREN_SPLIT_PCT": 30}, {"MSIREN_SPLIT_MIN": 3200, "MSIREN_SPLIT_PCT": 1}, {"MSIREN_SPLIT_MIN": 3300}):
        m = make_with_env(sd, env, act=act, precision="f16x3")
        d_i, d_o = m.device_array(tiles.shape).copy_from(tiles), [m.device_array((B, 24, 24)) for _ in range(2)]
        for streams in (1, 2):
            _lib.check(m._lib.msiren_set_streams(m._h, streams))
            _lib.check(m._lib.msiren_profile_enable(m._h, 1))
            for k in range(4):   # back to back: the second part's modulations cross streams, call after call
                run_dev(m, d_i, B, d_o[k & 1])
            m.sync()
            ks = {k["kernel"]: k for k in m.profile_kernels()}
            _lib.check(m._lib.msiren_profile_enable(m._h, 0))
            assert sum(k["coords"] for k in ks.values()) == 4 * B * 576 and all(k["launches"] == 4 for k in ks.values())
            if streams == 1:   # the call overlaps with itself
                assert set(ks) == {f"siren_trunk_f16x3n_kernel<{a},3,5>", f"siren_trunk_f16x3w_kernel<{a},4>"}, ks
                assert ks[f"siren_trunk_f16x3w_kernel<{a},4>"]["coords"] > ks[f"siren_trunk_f16x3n_kernel<{a},3,5>"]["coords"]
            else:              # two streams: consecutive calls overlap with each other, nothing is cut
                assert set(ks) == {f"siren_trunk_f16x3n_kernel<{a},3,5>"}, ks
            for o in d_o:
                assert np.array_equal(o.numpy(), ref), (env, streams)
        # the synchronous host call goes the same way
        assert np.array_equal(m(tiles), ref)
        # below the threshold nothing is cut
        _lib.check(m._lib.msiren_set_streams(m._h, 1))
        run_dev(m, d_i, 3100, d_o[0])
        m.sync()
        assert m.last_trunk_kernel() == f"siren_trunk_f16x3w_kernel<{a},4>"
        assert np.array_equal(d_o[0].numpy()[:3100], ref[:3100])


def test_split_is_not_taken_where_it_does_not_apply():
    """(with MSIREN_SPLIT_MIN=3200) Other depths, the exact-fp32 trunk and the masked slice pipeline (tile count known to the device only) run uncut."""
    tiles = np.random.default_rng(12).random((3300, 32, 32), dtype=np.float32)
    for L, prec in ((4, "f16x3"), (5, "fp32")):
        sd = syn.make_state_dict(seed=3, num_layers=L, trained_like=True)
        m = make_with_env(sd, {"MSIREN_SPLIT_MIN": 3200}, L=L, precision=prec)
        _lib.check(m._lib.msiren_profile_enable(m._h, 1))
        out = m(tiles)
        assert len(m.profile_kernels()) == 1
        check(out[:16], orc.modulated_siren_forward(sd, tiles[:16], num_layers=L, dtype=np.float64))
    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = make_with_env(sd, {"MSIREN_SPLIT_MIN": 3200})
    imgs = np.stack([syn.make_slice(k, brain_mask=True) for k in range(9)])
    _lib.check(m._lib.msiren_profile_enable(m._h, 1))
    rec = m.reconstruct(imgs)
    assert len(m.profile_kernels()) == 1
    assert np.array_equal(rec[8], m.reconstruct(imgs[8]))


@pytest.mark.parametrize("act", ["sine", "morlet"])
def test_two_stream_headline_path_vs_reference_fixture(act):
    """What bench.py's timed region runs: msiren_set_streams(h, 2), alternating msiren_forward_tiles_dev calls (the
    register-resident trunk beside the next call's encoder / Modulator) -- every output against the REFERENCE's."""
    g = load_golden(f"forward_trained_{act}.npz")
    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = make_model(sd, act=act, precision="f16x3")
    tiles = np.random.default_rng(42).random((8, 32, 32), dtype=np.float32)   # the fixture's input (oracle/gen_fixtures.py)
    big = np.concatenate([tiles] * 50)                                          # 400 tiles: the full-size kernel instances
    d_in, d_big = m.device_array(tiles.shape).copy_from(tiles), m.device_array(big.shape).copy_from(big)
    d_out = [m.device_array((8, 24, 24)) for _ in range(6)]
    d_bout = [m.device_array((400, 24, 24)) for _ in range(6)]
    _lib.check(m._lib.msiren_set_streams(m._h, 2))
    for k in range(6):
        run_dev(m, d_big, 400, d_bout[k])
        if k == 0:
            assert m.last_trunk_kernel() == f"siren_trunk_f16x3n_kernel<{1 if act == 'morlet' else 0},3,5>"
        run_dev(m, d_in, 8, d_out[k])
    m.sync()
    for o in d_out:
        check(o.numpy(), g["out"])
    for o in d_bout:
        got = o.numpy().reshape(50, 8, 24, 24)
        for r in (0, 17, 49):
            check(got[r], g["out"])


def test_full_slice_reconstruct_vs_fp64_oracle():
    """One full 320x320 masked slice through the device pipeline (tiles -> black filter -> forward -> weighted fold)
    against the oracle's float64 reconstruction (tiling.py:10-140,244-303 + modulated_siren.py:435-457)."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = make_model(sd)
    img = syn.make_slice(2, brain_mask=True)
    ref = orc.reconstruct_slice(sd, img, num_layers=5, dtype=np.float64)
    rec = m.reconstruct(img)
    check(rec, ref)
    _lib.check(m._lib.msiren_set_streams(m._h, 2))
    d_img, d_rec = m.device_array((1, 320, 320)).copy_from(img[None]), [m.device_array((1, 320, 320)) for _ in range(2)]
    for k in range(4):
        _lib.check(m._lib.msiren_reconstruct_slices_dev(m._h, d_img.ptr, 1, 320, 320, d_rec[k & 1].ptr))
    m.sync()
    for o in d_rec:
        assert np.array_equal(o.numpy()[0], rec)


@pytest.mark.parametrize("H,Z,L,B", [(256, 256, 5, 1100), (256, 256, 5, 1037), (64, 48, 3, 1030), (48, 16, 2, 1025), (512, 128, 10, 1056), (512, 128, 3, 300)])
def test_tiled_linear_layers_same_bits_as_the_16x16_kernel(H, Z, L, B):
    """Throughput sizes run conv3, Linear(64, Z) and the Modulator layers on 32 x 32 output tiles
    (linear_mfma_tile_kernel<2, 2>), latency sizes on 16 x 16: same MFMA chains, same K split, same reduction order --
    `self.modulator(self.encoder(tiles))` (modulated_siren.py:446) must not depend on the batch a tile came in."""
    kw = dict(dim_hidden=H, num_layers=L, latent_dim=Z)
    sd = syn.make_state_dict(seed=21, trained_like=True, **kw) if H != 512 else \
        syn.make_state_dict(seed=21, modulator_bias_center=0.25, encoder_gain=10.0, **kw)
    mk = dict(H=H, L=L, Z=Z, precision="fp32")
    small = make_with_env(sd, {"MSIREN_LINEAR_TILE_MIN": 0}, **mk)
    tiled = make_with_env(sd, {"MSIREN_LINEAR_TILE_MIN": 1}, **mk)
    auto = make_model(sd, **mk)   # default threshold: 1024 rows (256 for layers of >= 512 outputs)
    tiles = np.random.default_rng(B).random((B, 32, 32), dtype=np.float32)
    z = small.encoder(tiles)
    assert np.array_equal(tiled.encoder(tiles), z) and np.array_equal(auto.encoder(tiles), z)
    assert np.array_equal(auto.encoder(tiles[:70]), z[:70])          # (16 x 16 kernel in the same handle)
    check(z[:40], orc.encoder_forward(sd, tiles[:40], dtype=np.float64), tol=1e-5)
    mods = small.modulator(z)
    for a, b, c in zip(mods, tiled.modulator(z), auto.modulator(z)):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    ref = orc.modulator_forward(sd, z[:40].astype(np.float64), num_layers=L, dtype=np.float64)
    for a, r in zip(mods, ref):
        check(a[:40], r, tol=1e-5)
    # the masked slice pipeline hands the row count over on the device
    if H == 256:
        imgs = np.stack([syn.make_slice(k, brain_mask=True) for k in range(5)])
        assert np.array_equal(small.reconstruct(imgs), tiled.reconstruct(imgs))


def test_host_call_of_several_slices_pipelines_itself_same_bits():
    """A numpy -> numpy call of >= MSIREN_HOST_PIPE_MIN tiles (default 2400; 800 / 128 here) cuts itself into chunks over the handle's two streams (uploads and downloads beside the
    other chunk's kernels; msiren_forward_tiles_impl).  Patches are independent and every trunk / prologue instance gives the same
    bits, so nothing may change -- also with asynchronous *_dev work still pending on the helper stream of a two-stream handle."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    B = 1339
    tiles = np.random.default_rng(17).random((B, 32, 32), dtype=np.float32)
    one = make_with_env(sd, {"MSIREN_HOST_CHUNKS": 1, "MSIREN_SPLIT_MIN": 0}, precision="f16x3")
    ref = one(tiles)
    check(ref[:24], orc.modulated_siren_forward(sd, tiles[:24], num_layers=5, dtype=np.float64))
    for env in ({"MSIREN_HOST_PIPE_MIN": 800}, {"MSIREN_HOST_PIPE_MIN": 800, "MSIREN_HOST_FIRST": 56, "MSIREN_HOST_PIECE": 200}, {"MSIREN_HOST_PIPE_MIN": 128, "MSIREN_HOST_FIRST": 40}):
        m = make_with_env(sd, env, precision="f16x3")
        _lib.check(m._lib.msiren_profile_enable(m._h, 1))
        assert np.array_equal(m(tiles), ref), env
        ks = {k["kernel"]: k for k in m.profile_kernels()}
        assert set(ks) == {"siren_trunk_f16x3n_kernel<0,3,5>", "siren_trunk_f16x3w_kernel<0,4>"} and sum(k["coords"] for k in ks.values()) == B * 576, ks
        assert ks["siren_trunk_f16x3w_kernel<0,4>"]["launches"] == 1      # the last chunk
        assert np.array_equal(m(tiles[:100]), ref[:100])                    # (below the threshold: one chunk, buffers in place)
        # two-stream handle, un-synced device calls on both streams, then the host call
        _lib.check(m._lib.msiren_set_streams(m._h, 2))
        d_in = m.device_array((400, 32, 32)).copy_from(tiles[:400])
        d_out = [m.device_array((400, 24, 24)) for _ in range(3)]
        for k in range(3):
            run_dev(m, d_in, 400, d_out[k])
        assert np.array_equal(m(tiles), ref), env
        m.sync()
        for o in d_out:
            assert np.array_equal(o.numpy(), ref[:400])
        _lib.check(m._lib.msiren_set_streams(m._h, 1))


def test_out_of_domain_modulation_in_the_second_part_of_a_cut_call():
    """With MSIREN_SPLIT_MIN=3200 a >= 3200-tile call on a one-stream handle is cut in two (forward_tiles_split); a tile whose modulations leave the fp16
    domain lies in the SECOND part: that part's launch is repaired by the conditional exact-fp32 trunk (which reads mods2 with
    p.B = B1), the first part keeps its split-fp16 bits."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    B = 3300
    tiles = np.random.default_rng(23).random((B, 32, 32), dtype=np.float32)
    m = make_with_env(sd, {"MSIREN_SPLIT_MIN": 3200}, precision="f16x3")
    clean = m(tiles)
    bad = tiles.copy()
    bad[3000] *= 3e7          # latent ~1e7 -> modulations far beyond 65504
    z = m.encoder(bad)
    mods = np.stack(m.modulator(z), 0)
    assert np.abs(mods[:, 3000]).max() > 1e5 and np.isfinite(mods).all()
    B0 = (B * 12 // 100 + 15) // 16 * 16                      # forward_tiles_split's first part (MSIREN_SPLIT_PCT = 12)
    d_in, d_out = m.device_array(bad.shape).copy_from(bad), m.device_array((B, 24, 24))
    _lib.check(m._lib.msiren_profile_enable(m._h, 1))
    run_dev(m, d_in, B, d_out)
    m.sync()
    assert len(m.profile_kernels()) == 2                       # the call was cut
    got = d_out.numpy()
    exact = make_model(sd, precision="fp32")
    assert np.array_equal(got[:B0], clean[:B0])                                  # first part: untouched
    assert np.array_equal(got[B0:], exact.forward_mods(mods[:, B0:]))            # second part: the exact-fp32 trunk's bits
    assert np.isfinite(got).all()


def test_page_locked_buffers_same_bits_recycled_and_outlive_the_model():
    """msiren_host_alloc / model.pinned_empty / model.pin_outputs: with page-locked input and output the copies are asynchronous DMA
    without staging (one slice: 437 -> 394 us).  Same bits as the pageable call, at every size and through the cut call of
    several slices; the pool recycles blocks; arrays stay valid after their model is gone."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    tiles = np.random.default_rng(31).random((400, 32, 32), dtype=np.float32)
    m = make_with_env(sd, {"MSIREN_HOST_PIPE_MIN": 800}, precision="f16x3")
    ref = m(tiles)                                   # pageable in, pageable out: one chunk
    _lib.check(m._lib.msiren_profile_enable(m._h, 1))
    m(tiles)
    assert len(m.profile_kernels()) == 1
    pin = m.pinned_empty(tiles.shape)
    pin[...] = tiles
    m.pin_outputs(True)
    out = m(pin)
    assert np.array_equal(out, ref)
    big = m.pinned_empty((1000, 32, 32))
    big[...] = np.concatenate([tiles, tiles, tiles[:200]])
    _lib.check(m._lib.msiren_profile_enable(m._h, 1))
    got = m(big)
    assert len(m.profile_kernels()) == 2                                     # >= MSIREN_HOST_PIPE_MIN tiles: the call cuts itself
    _lib.check(m._lib.msiren_profile_enable(m._h, 0))
    assert np.array_equal(got[:400], ref) and np.array_equal(got[400:800], ref) and np.array_equal(got[800:], ref[:200])
    del got, big
    ptr0 = out.ctypes.data
    for n in (1, 100, 128, 399):                     # below / above the threshold, ragged
        assert np.array_equal(m(pin[:n]), ref[:n]), n
    assert np.array_equal(m(tiles), ref)             # pageable input with pinned output: one chunk, still right
    del out
    again = m(pin)
    assert again.ctypes.data == ptr0 or np.array_equal(again, ref)   # (the block came back from the pool)
    m.pin_outputs(False)
    keep = m(pin).copy()
    del m
    import gc
    gc.collect()
    assert np.array_equal(again, ref) and np.array_equal(keep, ref) and np.array_equal(pin, tiles)


def test_by_default_a_large_call_on_a_one_stream_handle_is_one_trunk_launch():
    """Round 5: MSIREN_SPLIT_MIN defaults to 0 -- behind the one-launch prologue the uncut call is as fast at 64 slices and faster at 8
    (profiles/r5/07_*)."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = make_model(sd, precision="f16x3")
    tiles = np.random.default_rng(3).random((3300, 32, 32), dtype=np.float32)
    d_in, d_out = m.device_array(tiles.shape).copy_from(tiles), m.device_array((3300, 24, 24))
    _lib.check(m._lib.msiren_profile_enable(m._h, 1))
    run_dev(m, d_in, 3300, d_out)
    m.sync()
    ks = m.profile_kernels()
    assert len(ks) == 1 and ks[0]["kernel"] == "siren_trunk_f16x3w_kernel<0,4>" and ks[0]["coords"] == 3300 * 576, ks
    cut = make_with_env(sd, {"MSIREN_SPLIT_MIN": 3200}, precision="f16x3")
    assert np.array_equal(d_out.numpy(), cut(tiles))


def test_host_slice_call_in_place_same_bits_as_staged_copies():
    """msiren_reconstruct_slices (numpy slice -> numpy reconstruction, the reference's metrics_error pattern): since round 5 the caller's
    buffers are page-locked for the call, the fold stores into the caller's array and the image arrives by DMA from the locked pages
    (MSIREN_RECON_ZC=5; 0 = staged copies, 3 = both buffers in place).  Same bits in every mode -- masked slices, several slices per call,
    sizes that are not a multiple of the stride, buffers that are page-locked in part by the caller."""
    import ctypes

    sd = syn.make_state_dict(seed=7, trained_like=True)
    imgs = np.stack([syn.make_slice(k, 320, 320, brain_mask=bool(k & 1)) for k in range(3)])
    odd = syn.make_slice(5, 200, 170)
    staged = make_with_env(sd, {"MSIREN_RECON_ZC": 0}, precision="f16x3")
    ref, ref_odd = staged.reconstruct(imgs), staged.reconstruct(odd)
    assert ref.shape == (3, 320, 320) and ref_odd.shape == (208, 176) and np.isfinite(ref).all()
    for zc in (5, 1, 3, 4):
        m = make_with_env(sd, {"MSIREN_RECON_ZC": zc}, precision="f16x3")
        assert np.array_equal(m.reconstruct(imgs), ref), zc
        assert np.array_equal(m.reconstruct(imgs[1]), ref[1]), zc
        assert np.array_equal(m.reconstruct(odd), ref_odd), zc
    # the caller page-locks the first slice of a stack and calls on slices 0..1 and 1..2: the first range is page-locked in part
    hip = ctypes.CDLL("libamdhip64.so")
    m = make_model(sd, precision="f16x3")
    stack = imgs.copy()
    assert hip.hipHostRegister(ctypes.c_void_p(stack.ctypes.data), ctypes.c_size_t(320 * 320 * 4), ctypes.c_uint(0)) == 0
    try:
        assert np.array_equal(m.reconstruct(stack[:2]), ref[:2])
        assert np.array_equal(m.reconstruct(stack[1:]), ref[1:])
        assert np.array_equal(m.reconstruct(stack[0]), ref[0])
    finally:
        assert hip.hipHostUnregister(ctypes.c_void_p(stack.ctypes.data)) == 0


def test_back_to_back_one_stream_calls_across_the_edge_of_the_fp16_domain():
    """One-stream handle, msiren_forward_tiles_dev calls back to back without a sync; out-of-domain batches and clean ones alternate, other entry
    points cut in between.  The conditional exact-fp32 launch of call k reads the stream's modulations, which call k+1's prologue overwrites:
    whatever the library does to get that launch off the critical path (round 5 tried a side stream: slower, profiles/r5/05_*), every buffer
    must hold the exact-fp32 bits for a flagged batch and the split-fp16 bits otherwise -- here against a second handle used synchronously."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    rng = np.random.default_rng(31)
    clean = rng.random((400, 32, 32), dtype=np.float32)
    bad = clean.copy()
    bad[37] *= 3e7                                   # latent ~1e7 -> modulations far beyond 65504
    small = clean[:7].copy()
    plain = make_model(sd, precision="f16x3")
    want = {"clean": plain(clean), "bad": plain(bad), "small": plain(small)}
    z = plain.encoder(clean[:64])
    want_latent = plain.forward_latent(z)
    img = syn.make_slice(4, 320, 320, brain_mask=True)
    want_img = plain.reconstruct(img)
    exact = make_model(sd, precision="fp32")
    assert np.array_equal(want["bad"], exact.forward_mods(np.stack(plain.modulator(plain.encoder(bad)), 0)))   # the whole flagged batch: fp32 bits
    m = make_model(sd, precision="f16x3")
    d_in = {k: m.device_array(v.shape).copy_from(v) for k, v in (("clean", clean), ("bad", bad), ("small", small))}
    d_z, d_img = m.device_array(z.shape).copy_from(z), m.device_array((1, 320, 320)).copy_from(img[None])
    seq = ["clean", "bad", "clean", "bad", "bad", "small", "clean", "small", "bad", "clean"]
    outs = [m.device_array((d_in[k].shape[0], 24, 24)) for k in seq]
    d_lat, d_rec = m.device_array((64, 24, 24)), m.device_array((1, 320, 320))
    for rep in range(2):
        for i, k in enumerate(seq):
            run_dev(m, d_in[k], d_in[k].shape[0], outs[i])
            if i == 3:    # another entry point in the middle of the run: it orders itself behind the conditional launch that is still aside
                _lib.check(m._lib.msiren_forward_latent_dev(m._h, d_z.ptr, 64, d_lat.ptr, None))
            if i == 6:
                _lib.check(m._lib.msiren_reconstruct_slices_dev(m._h, d_img.ptr, 1, 320, 320, d_rec.ptr))
        if rep == 0:
            m.sync()
    m.sync()
    for i, k in enumerate(seq):
        assert np.array_equal(outs[i].numpy(), want[k]), (i, k)
    assert np.array_equal(d_lat.numpy(), want_latent) and np.array_equal(d_rec.numpy()[0], want_img)
    # a host call right behind an asynchronous one
    run_dev(m, d_in["bad"], 400, outs[0])
    assert np.array_equal(m(clean), want["clean"])
    m.sync()
    assert np.array_equal(outs[0].numpy(), want["bad"])


def test_synchronous_host_call_checks_the_domain_flag_on_the_host():
    """A one-chunk msiren_forward_tiles call waits for its stream anyway: since round 5 its trunk raises the out-of-domain flag in host memory and
    the call looks at it after the wait -- no conditional launch per call; a flagged call runs the exact-fp32 trunk then (and downloads again
    where it copies).  Same buffers as with the conditional launch (MSIREN_HOST_CHECK=0): in place (400 tiles), with copies (48 tiles), on
    page-locked arrays; clean calls before and after a flagged one keep their split-fp16 bits; the event counter counts."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    rng = np.random.default_rng(33)
    clean = rng.random((400, 32, 32), dtype=np.float32)
    bad = clean.copy()
    bad[37] *= 3e7
    exact = make_model(sd, precision="fp32")
    old = make_with_env(sd, {"MSIREN_HOST_CHECK": 0}, precision="f16x3")
    mods_of = lambda x: np.stack(old.modulator(old.encoder(x)), 0)
    want_clean, want_bad, want_bad48 = old(clean), exact.forward_mods(mods_of(bad)), exact.forward_mods(mods_of(bad[:48]))
    assert np.array_equal(old(bad), want_bad) and np.array_equal(old(bad[:48]), want_bad48)
    m = make_model(sd, precision="f16x3")
    def events(mm):
        import ctypes as C

        n = C.c_int64()
        _lib.check(mm._lib.msiren_range_events(mm._h, C.byref(n)))
        return n.value

    e0 = events(m)
    assert np.array_equal(m(clean), want_clean) and events(m) == e0
    assert np.array_equal(m(bad), want_bad) and events(m) == e0 + 1            # in place
    assert np.array_equal(m(clean), want_clean) and events(m) == e0 + 1
    assert np.array_equal(m(bad[:48]), want_bad48) and events(m) == e0 + 2     # copies: downloaded again
    assert np.array_equal(m(clean[:48]), want_clean[:48])
    pin = m.pinned_empty(bad.shape)
    pin[...] = bad
    m.pin_outputs(True)
    assert np.array_equal(m(pin), want_bad) and events(m) == e0 + 3
    m.pin_outputs(False)
    # asynchronous calls keep the conditional launch; a host call right behind one
    d_in, d_out = m.device_array(bad.shape).copy_from(bad), m.device_array((400, 24, 24))
    run_dev(m, d_in, 400, d_out)
    assert np.array_equal(m(clean), want_clean)
    m.sync()
    assert np.array_equal(d_out.numpy(), want_bad)


def test_fused_tiling_flags_plan_same_bits_as_the_separate_kernels():
    """Round 5: image_to_patches + black_flags + compact_flags as one launch (the workgroup that draws the last ticket builds the plan) and the
    pass counter's reset inside the fold.  MSIREN_TILING_FUSED=0 = the separate kernels: the same reconstruction bit for bit -- masked and
    unmasked slices, several per call, odd sizes, all-black and no-black inputs, many calls in a row (the ticket counter is put back each time),
    one and two streams."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    imgs = np.stack([syn.make_slice(k, 320, 320, brain_mask=bool(k % 3)) for k in range(6)])
    odd = syn.make_slice(9, 200, 170, brain_mask=True)
    black = np.zeros((320, 320), np.float32)
    sep = make_with_env(sd, {"MSIREN_TILING_FUSED": 0}, precision="f16x3")
    m = make_with_env(sd, {"MSIREN_TILING_FUSED": 2}, precision="f16x3")      # (2: on the asynchronous API as well; the default fuses in host calls only)
    want, want_odd = sep.reconstruct(imgs), sep.reconstruct(odd)
    assert np.array_equal(make_model(sd, precision="f16x3").reconstruct(imgs), want)
    assert np.array_equal(sep.reconstruct(black), np.zeros_like(black))
    for rep in range(3):
        assert np.array_equal(m.reconstruct(imgs), want)
        assert np.array_equal(m.reconstruct(odd), want_odd)
        assert np.array_equal(m.reconstruct(black), np.zeros_like(black))
        for k in range(6):
            assert np.array_equal(m.reconstruct(imgs[k]), want[k]), k
    # asynchronous calls back to back on one and two streams, forward calls in between (they share the pass counter the fold resets)
    tiles = np.random.default_rng(41).random((100, 32, 32), dtype=np.float32)
    want_t = sep(tiles)
    d_i = m.device_array(imgs.shape).copy_from(imgs)
    d_t, d_to = m.device_array(tiles.shape).copy_from(tiles), [m.device_array((100, 24, 24)) for _ in range(4)]
    for streams in (1, 2, 3):
        _lib.check(m._lib.msiren_set_streams(m._h, streams))
        d_r = [m.device_array((1, 320, 320)) for _ in range(12)]
        for j in range(12):
            _lib.check(m._lib.msiren_reconstruct_slices_dev(m._h, d_i.ptr + (j % 6) * 320 * 320 * 4, 1, 320, 320, d_r[j].ptr))
            if j % 3 == 1:
                run_dev(m, d_t, 100, d_to[j // 3])
        m.sync()
        for j in range(12):
            assert np.array_equal(d_r[j].numpy()[0], want[j % 6]), (streams, j)
        for o in d_to:
            assert np.array_equal(o.numpy(), want_t)
    _lib.check(m._lib.msiren_set_streams(m._h, 1))
