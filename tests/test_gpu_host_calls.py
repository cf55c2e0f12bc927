"""Host-pointer calls, the paths the benchmark's timed region runs, and the same-bits properties between code paths that all ship:
one chunk in place / the pipelined cut of several slices; the host-side domain check of synchronous calls / the conditional launch of
asynchronous ones; the fused tiling kernel of synchronous slice calls / the separate kernels of asynchronous ones; the 16 x 16 / the
32 x 32-tile Linear kernels of the exact-fp32 prologue.  Round 6: the paths that had lost their A/Bs (the cut of one large *_dev call,
MSIREN_SPLIT_MIN; per-call page-locking; the knobs that selected losers) are gone from the library, and so are their tests; every
comparison here is between two paths the library takes by itself.
"""
import os

import numpy as np
import pytest

from conftest import load_golden
from mri_inr_amd import _lib, synthetic as syn
from oracle import siren_oracle as orc
from test_gpu_parity import check, make_model, make_with_env

pytestmark = pytest.mark.gpu


def run_dev(m, d_in, n, d_out):
    _lib.check(m._lib.msiren_forward_tiles_dev(m._h, d_in.ptr, n, d_out.ptr))

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
    """Exact-fp32 handles: throughput sizes (>= 1024 rows; >= 256 for layers of >= 512 outputs) run conv3, Linear(64, Z) and the
    Modulator layers on 32 x 32 output tiles (linear_mfma_tile_kernel<2, 2>), latency sizes on 16 x 16: same MFMA chains, same K split,
    same reduction order -- `self.modulator(self.encoder(tiles))` (modulated_siren.py:446) must not depend on the batch a tile came in.
    The whole batch (tiled kernel) against the same rows in pieces below the threshold (16 x 16 kernel), in one handle -- down to pieces
    of 1, 7 and 47 tiles: until round 6 an fp32 handle's encoder below 48 tiles was ONE fused per-tile kernel whose VALU sums ran in
    another order (same gate, other last bits); it now serves only latent sizes the MFMA kernels do not take (48-16-2 here: Z = 16 is a
    multiple of 16, so that case is on the MFMA path as well)."""
    kw = dict(dim_hidden=H, num_layers=L, latent_dim=Z)
    sd = syn.make_state_dict(seed=21, trained_like=True, **kw) if H != 512 else \
        syn.make_state_dict(seed=21, modulator_bias_center=0.25, encoder_gain=10.0, **kw)
    m = make_model(sd, H=H, L=L, Z=Z, precision="fp32")
    tiles = np.random.default_rng(B).random((B, 32, 32), dtype=np.float32)
    z = m.encoder(tiles)                                              # tiled kernels
    edges = [0, 1, 8, 55] + list(range(200, B - 100, 200)) + [B]      # pieces of 1, 7, 47 and 100..299 rows: below every threshold
    pieces = list(zip(edges[:-1], edges[1:]))
    z_small = np.concatenate([m.encoder(tiles[lo:hi]) for lo, hi in pieces])
    assert np.array_equal(z_small, z)
    check(z[:40], orc.encoder_forward(sd, tiles[:40], dtype=np.float64), tol=1e-5)
    assert np.array_equal(m.encoder(tiles[:40]), z[:40])
    mods = m.modulator(z)
    for l, a in enumerate(mods):
        small = np.concatenate([m.modulator(z[lo:hi])[l] for lo, hi in pieces])
        assert np.array_equal(a, small), l
    ref = orc.modulator_forward(sd, z[:40].astype(np.float64), num_layers=L, dtype=np.float64)
    for a, r in zip(mods, ref):
        check(a[:40], r, tol=1e-5)
    # the masked slice pipeline hands the row count over on the device: five slices at once (tiled) against one by one (16 x 16)
    if H == 256:
        imgs = np.stack([syn.make_slice(k, brain_mask=True) for k in range(5)])
        rec = m.reconstruct(imgs)
        for k in range(5):
            assert np.array_equal(m.reconstruct(imgs[k]), rec[k]), k

def test_host_call_of_several_slices_pipelines_itself_same_bits():
    """A numpy -> numpy call of >= MSIREN_HOST_PIPE_MIN tiles (default 2400; 800 / 128 here) cuts itself into chunks over the handle's two
    streams (uploads and downloads beside the other chunk's kernels; msiren_forward_tiles_impl, host_plan.h).  Patches are independent and
    every trunk / prologue instance gives the same bits, so nothing may change against the one-chunk call (the default below the threshold)
    -- also with asynchronous *_dev work still pending on the helper stream of a two-stream handle."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    B = 1339
    tiles = np.random.default_rng(17).random((B, 32, 32), dtype=np.float32)
    one = make_model(sd, precision="f16x3")               # 1339 < 2400: one chunk
    _lib.check(one._lib.msiren_profile_enable(one._h, 1))
    ref = one(tiles)
    assert [k["kernel"] for k in one.profile_kernels()] == ["siren_trunk_f16x3w_kernel<0,4>"]
    check(ref[:24], orc.modulated_siren_forward(sd, tiles[:24], num_layers=5, dtype=np.float64))
    for env in ({"MSIREN_HOST_PIPE_MIN": 800}, {"MSIREN_HOST_PIPE_MIN": 128}):
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

def test_page_locked_buffers_same_bits_recycled_and_outlive_the_model():
    """msiren_host_alloc / model.pinned_empty / model.pin_outputs: with page-locked input and output the kernels work on the caller's
    arrays in place.  Same bits as the pageable call, at every size and through the cut call of several slices; the pool recycles blocks.
    Arrays outlive their model: `del model` destroys the GPU handle AT ONCE (the pool holds no reference to the model: no cycle, no wait
    for a gc pass), the arrays stay readable, and their blocks are freed later through msiren_host_free(NULL, ptr)."""
    import ctypes
    import gc
    import weakref

    sd = syn.make_state_dict(seed=7, trained_like=True)
    tiles = np.random.default_rng(31).random((400, 32, 32), dtype=np.float32)
    m = make_with_env(sd, {"MSIREN_HOST_PIPE_MIN": 800}, precision="f16x3")
    ref = m(tiles)                                   # pageable in, pool block out: one chunk
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
    # the model goes, by reference count alone
    lib, cell, pool = m._lib, m._pinned._cell, weakref.ref(m._pinned)
    gc.disable()
    try:
        wm = weakref.ref(m)
        del m
        assert wm() is None and cell[0] is None      # __del__ has run: handle destroyed, the pool's handle cell emptied
    finally:
        gc.enable()
    k = ctypes.c_int32(-1)
    _lib.check(lib.msiren_host_range_kind(ctypes.c_void_p(again.ctypes.data), ctypes.c_size_t(again.nbytes), ctypes.byref(k)))
    assert k.value == 1                              # the block is still page-locked memory under the live array ...
    assert np.array_equal(again, ref) and np.array_equal(keep, ref) and np.array_equal(pin, tiles)
    addr, nbytes = again.ctypes.data, again.nbytes
    del again, pin, ref                              # ... and goes through msiren_host_free(NULL, ptr) when the last array does (ref is a pool block too)
    gc.collect()
    assert pool() is None
    _lib.check(lib.msiren_host_range_kind(ctypes.c_void_p(addr), ctypes.c_size_t(nbytes), ctypes.byref(k)))
    assert k.value == 0

def test_a_large_call_on_a_one_stream_handle_is_one_trunk_launch():
    """Behind the one-launch prologue the uncut call is as fast at 64 slices and faster at 8 (profiles/r5/07_*): round 6 deleted the cut."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = make_model(sd, precision="f16x3")
    tiles = np.random.default_rng(3).random((3300, 32, 32), dtype=np.float32)
    d_in, d_out = m.device_array(tiles.shape).copy_from(tiles), m.device_array((3300, 24, 24))
    _lib.check(m._lib.msiren_profile_enable(m._h, 1))
    run_dev(m, d_in, 3300, d_out)
    m.sync()
    ks = m.profile_kernels()
    assert len(ks) == 1 and ks[0]["kernel"] == "siren_trunk_f16x3w_kernel<0,4>" and ks[0]["coords"] == 3300 * 576, ks
    got = d_out.numpy()
    check(got[:32], orc.modulated_siren_forward(sd, tiles[:32], num_layers=5, dtype=np.float64))
    check(got[-32:], orc.modulated_siren_forward(sd, tiles[-32:], num_layers=5, dtype=np.float64))
    assert np.array_equal(m(tiles), got)                       # the host call of 3300 tiles pipelines itself: same bits

def test_host_slice_call_in_place_same_bits_as_staged_copies():
    """msiren_reconstruct_slices (numpy slice -> numpy reconstruction, the reference's metrics_error pattern): where the caller's
    reconstruction array is page-locked memory (the mirror's pool, its default) the fold stores into it; with pin_outputs(False) the
    reconstruction is downloaded.  Same bits -- masked slices, several slices per call, sizes that are not a multiple of the stride,
    buffers that are page-locked in part by the caller."""
    import ctypes

    sd = syn.make_state_dict(seed=7, trained_like=True)
    imgs = np.stack([syn.make_slice(k, 320, 320, brain_mask=bool(k & 1)) for k in range(3)])
    odd = syn.make_slice(5, 200, 170)
    m = make_model(sd, precision="f16x3")
    m.pin_outputs(False)
    ref, ref_odd = m.reconstruct(imgs), m.reconstruct(odd)          # staged copies
    assert ref.shape == (3, 320, 320) and ref_odd.shape == (208, 176) and np.isfinite(ref).all()
    m.pin_outputs(True)
    assert np.array_equal(m.reconstruct(imgs), ref)
    assert np.array_equal(m.reconstruct(imgs[1]), ref[1])
    assert np.array_equal(m.reconstruct(odd), ref_odd)
    # the caller page-locks the first slice of a stack and calls on slices 0..1 and 1..2: the first range is page-locked in part
    hip = ctypes.CDLL("libamdhip64.so")
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
    """A one-chunk msiren_forward_tiles call waits for its stream anyway: its trunk raises the out-of-domain flag in host memory and the
    call looks at it after the wait -- no conditional launch per call; a flagged call runs the exact-fp32 trunk then (and downloads again
    where it copies).  Same buffers as the asynchronous API gives with its conditional launch: in place (400 tiles), with copies
    (48 tiles), on page-locked arrays; clean calls before and after a flagged one keep their split-fp16 bits; the event counter counts."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    rng = np.random.default_rng(33)
    clean = rng.random((400, 32, 32), dtype=np.float32)
    bad = clean.copy()
    bad[37] *= 3e7
    exact = make_model(sd, precision="fp32")
    m = make_model(sd, precision="f16x3")

    def via_dev(x):      # the asynchronous entry point: conditional launch on the stream
        d_i, d_o = m.device_array(x.shape).copy_from(x), m.device_array((x.shape[0], 24, 24))
        run_dev(m, d_i, x.shape[0], d_o)
        m.sync()
        return d_o.numpy()

    mods_of = lambda x: np.stack(m.modulator(m.encoder(x)), 0)
    want_clean, want_bad, want_bad48 = via_dev(clean), exact.forward_mods(mods_of(bad)), exact.forward_mods(mods_of(bad[:48]))
    assert np.array_equal(via_dev(bad), want_bad) and np.array_equal(via_dev(bad[:48]), want_bad48)

    def events(mm):
        import ctypes as C

        n = C.c_int64()
        _lib.check(mm._lib.msiren_range_events(mm._h, C.byref(n)))
        return n.value

    e0 = events(m)
    assert np.array_equal(m(clean), want_clean) and events(m) == e0
    assert np.array_equal(m(bad), want_bad) and events(m) == e0 + 1            # in place
    assert np.array_equal(m(clean), want_clean) and events(m) == e0 + 1
    m.pin_outputs(False)
    assert np.array_equal(m(bad[:48]), want_bad48) and events(m) == e0 + 2     # copies: downloaded again
    assert np.array_equal(m(clean[:48]), want_clean[:48])
    m.pin_outputs(True)
    pin = m.pinned_empty(bad.shape)
    pin[...] = bad
    assert np.array_equal(m(pin), want_bad) and events(m) == e0 + 3
    # asynchronous calls keep the conditional launch; a host call right behind one
    d_in, d_out = m.device_array(bad.shape).copy_from(bad), m.device_array((400, 24, 24))
    run_dev(m, d_in, 400, d_out)
    assert np.array_equal(m(clean), want_clean)
    m.sync()
    assert np.array_equal(d_out.numpy(), want_bad)

def test_fused_tiling_flags_plan_same_bits_as_the_separate_kernels():
    """Synchronous host calls (msiren_reconstruct_slices) run image_to_patches + black_flags + compact_flags as ONE launch (the workgroup
    that draws the last ticket builds the plan) with the pass counter's reset inside the fold; asynchronous calls
    (msiren_reconstruct_slices_dev) keep the separate kernels (profiles/r5/13_*).  The same reconstruction bit for bit -- masked and
    unmasked slices, several per call, odd sizes, all-black and no-black inputs, many calls in a row (the ticket counter is put back each
    time), one to three streams, forward calls in between (they share the pass counter the fold resets)."""
    sd = syn.make_state_dict(seed=7, trained_like=True)
    imgs = np.stack([syn.make_slice(k, 320, 320, brain_mask=bool(k % 3)) for k in range(6)])
    odd = syn.make_slice(9, 200, 170, brain_mask=True)
    black = np.zeros((320, 320), np.float32)
    m = make_model(sd, precision="f16x3")

    def sep(x):          # the asynchronous entry point: separate kernels
        x3 = x if x.ndim == 3 else x[None]
        n, hh, ww = x3.shape
        d_i = m.device_array(x3.shape).copy_from(x3)
        d_r = m.device_array((n, (hh + 15) // 16 * 16, (ww + 15) // 16 * 16))
        _lib.check(m._lib.msiren_reconstruct_slices_dev(m._h, d_i.ptr, n, hh, ww, d_r.ptr))
        m.sync()
        r = d_r.numpy()
        return r if x.ndim == 3 else r[0]

    want, want_odd = sep(imgs), sep(odd)
    assert np.array_equal(sep(black), np.zeros_like(black))
    for rep in range(3):
        assert np.array_equal(m.reconstruct(imgs), want)                 # fused
        assert np.array_equal(m.reconstruct(odd), want_odd)
        assert np.array_equal(m.reconstruct(black), np.zeros_like(black))
        for k in range(6):
            assert np.array_equal(m.reconstruct(imgs[k]), want[k]), k
    tiles = np.random.default_rng(41).random((100, 32, 32), dtype=np.float32)
    want_t = m(tiles)
    d_i = m.device_array(imgs.shape).copy_from(imgs)
    d_t, d_to = m.device_array(tiles.shape).copy_from(tiles), [m.device_array((100, 24, 24)) for _ in range(4)]
    for streams in (1, 2, 3):
        _lib.check(m._lib.msiren_set_streams(m._h, streams))
        d_r = [m.device_array((1, 320, 320)) for _ in range(12)]
        for j in range(12):
            _lib.check(m._lib.msiren_reconstruct_slices_dev(m._h, d_i.ptr + (j % 6) * 320 * 320 * 4, 1, 320, 320, d_r[j].ptr))
            if j % 3 == 1:
                run_dev(m, d_t, 100, d_to[j // 3])
            if j == 7:
                assert np.array_equal(m.reconstruct(imgs[2]), want[2])   # a fused synchronous call in the middle of the queue
        m.sync()
        for j in range(12):
            assert np.array_equal(d_r[j].numpy()[0], want[j % 6]), (streams, j)
        for o in d_to:
            assert np.array_equal(o.numpy(), want_t)
    _lib.check(m._lib.msiren_set_streams(m._h, 1))
