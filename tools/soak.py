#!/usr/bin/env python3
"""Soak: random batch sizes, offsets and stream modes through the asynchronous API for a fixed time per configuration; every output
must equal, bit for bit, the rows of ONE reference run over the whole pool (a tile's output does not depend on the batch it came
in).  Exercises the pass queues (thousands of launches per handle), every trunk instance's size thresholds (half units, tails,
2-unit passes, the large-call split), the conditional fp32 launch and the two-stream hand-overs.

    python tools/soak.py [seconds per configuration = 60] [sine,morlet,config5_bf16]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mri_inr_amd import ModulatedSiren, _lib, synthetic as syn

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(2026)
POOL = 3400
tiles = rng.random((POOL, 32, 32), dtype=np.float32)
tiles[rng.random(POOL) < 0.05] *= 40.0   # a few tiles far outside what trained encoders see


def model(cfg):
    if cfg == "config5_bf16":
        sd = syn.make_state_dict(seed=21, dim_hidden=512, num_layers=10, latent_dim=128, modulator_bias_center=0.25, encoder_gain=10.0)
        m = ModulatedSiren(2, 512, 1, 10, 128, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine", residual=True, precision="bf16")
    else:
        sd = syn.make_state_dict(seed=7, trained_like=True)
        m = ModulatedSiren(2, 256, 1, 5, 256, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", cfg)
    m.load_state_dict(sd)
    m.to("cuda")
    return m


ok = True
for cfg in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("sine", "morlet", "config5_bf16")):
    m = model(cfg)
    d_t = m.device_array(tiles.shape).copy_from(tiles)
    d_ref = m.device_array((POOL, 24, 24))
    _lib.check(m._lib.msiren_forward_tiles_dev(m._h, d_t.ptr, POOL, d_ref.ptr))
    m.sync()
    ref = d_ref.numpy()
    assert np.isfinite(ref).all(), cfg
    # calls below 48 tiles get a reference of their own, computed in chunks of 47 (until round 5 the encoder of such calls was a fused kernel with
    # another summation order; since the one-launch prologue the two references are equal bit for bit -- printed below -- and stay as a check)
    ref_small = np.empty_like(ref)
    for o in range(0, POOL, 47):
        b = min(47, POOL - o)
        _lib.check(m._lib.msiren_forward_tiles_dev(m._h, d_t.ptr + o * 32 * 32 * 4, b, d_ref.ptr + o * 24 * 24 * 4))
    m.sync()
    ref_small = d_ref.numpy()
    print(f"{cfg}: fused-encoder path vs throughput path: max|diff| {np.abs(ref_small - ref).max():.3e}, rms {np.sqrt(np.mean((ref_small - ref) ** 2)):.3e}", flush=True)
    outs = [m.device_array((POOL, 24, 24)) for _ in range(6)]
    t_end, it, launches, bad = time.time() + secs, 0, 0, 0
    sizes = [1, 2, 3, 7, 8, 9, 28, 29, 57, 58, 100, 129, 256, 257, 400, 401, 1023, 1024, 1025, 3199, 3200, 3300]
    while time.time() < t_end:
        streams = int(rng.integers(1, 4))
        _lib.check(m._lib.msiren_set_streams(m._h, streams))
        pend = []
        for k in range(6):
            b = int(sizes[rng.integers(len(sizes))]) if rng.random() < 0.7 else int(rng.integers(1, POOL + 1))
            o = int(rng.integers(0, POOL - b + 1))
            _lib.check(m._lib.msiren_forward_tiles_dev(m._h, d_t.ptr + o * 32 * 32 * 4, b, outs[k].ptr))
            pend.append((k, o, b))
            launches += 1
        m.sync()
        for k, o, b in pend:
            got = outs[k].numpy()[:b]
            want = (ref_small if b < 48 else ref)[o:o + b]
            if not np.array_equal(got, want):
                bad += 1
                d = np.abs(got - want).max()
                print(f"  MISMATCH {cfg}: streams {streams} offset {o} size {b} max|diff| {d:.3e}", flush=True)
        it += 1
    print(f"{cfg}: {launches} calls in {secs:.0f} s, {bad} mismatches, last trunk {m.last_trunk_kernel()}", flush=True)
    ok = ok and bad == 0
    # the slice pipeline (black-tile filter on the device: the trunk learns its tile count from the plan): random runs of masked
    # slices against each slice reconstructed alone
    imgs = np.stack([syn.make_slice(k, 320, 320, brain_mask=True) for k in range(10)])
    d_i = m.device_array(imgs.shape).copy_from(imgs)
    d_r = m.device_array(imgs.shape)
    alone = np.stack([m.reconstruct(imgs[k]) for k in range(10)])
    t_end, calls, bad = time.time() + secs / 3, 0, 0
    while time.time() < t_end:
        _lib.check(m._lib.msiren_set_streams(m._h, int(rng.integers(1, 4))))
        n = int(rng.integers(1, 7))
        o = int(rng.integers(0, 10 - n + 1))
        _lib.check(m._lib.msiren_reconstruct_slices_dev(m._h, d_i.ptr + o * 320 * 320 * 4, n, 320, 320, d_r.ptr))
        m.sync()
        calls += 1
        if not np.array_equal(d_r.numpy()[:n], alone[o:o + n]):
            bad += 1
            print(f"  MISMATCH {cfg}: reconstruct slices {o}..{o + n - 1}", flush=True)
    print(f"{cfg}: {calls} slice-pipeline calls in {secs / 3:.0f} s, {bad} mismatches", flush=True)
    ok = ok and bad == 0
    if cfg != "sine":
        continue
    # host-pointer calls (pageable windows copied by the runtime, the mirror's page-locked pool blocks written in place, the pipelined plan
    # from 2400 tiles): two handles in two threads on random windows of ONE pageable input pool and ONE output pool
    import ctypes
    import threading

    fp = ctypes.POINTER(ctypes.c_float)
    handles = [m, model(cfg)]
    out_pool = np.zeros((2 * POOL, 24, 24), np.float32)
    stats = {"calls": 0, "bad": 0}
    lock = threading.Lock()
    host_sizes = [1, 47, 48, 63, 64, 65, 100, 400, 401, 800, 1600, 2399, 2400, 3300]

    def host_worker(i):
        try:
            host_loop(i)
        except Exception as e:  # noqa: BLE001
            with lock:
                stats["bad"] += 1
            print(f"  FAILED host call: thread {i}: {e!r}", flush=True)

    def host_loop(i):
        r = np.random.default_rng(100 + i)
        t_end = time.time() + secs / 2
        while time.time() < t_end:
            b = int(host_sizes[r.integers(len(host_sizes))]) if r.random() < 0.7 else int(r.integers(1, 1200))
            o = int(r.integers(0, POOL - b + 1))
            # outputs: a window of the shared pool (this thread's half, so that no two calls write the same bytes) or an array of its own
            if r.random() < 0.35:     # through the Python mirror: the output comes from its page-locked pool and is stored in place
                dst = handles[i](tiles[o:o + b])
            else:
                dst = out_pool[i * POOL + o:i * POOL + o + b] if r.random() < 0.7 else np.empty((b, 24, 24), np.float32)
                _lib.check(handles[i]._lib.msiren_forward_tiles(handles[i]._h, tiles[o:o + b].ctypes.data_as(fp), b, dst.ctypes.data_as(fp)))
            want = (ref_small if b < 48 else ref)[o:o + b]
            good = np.array_equal(dst, want)
            with lock:
                stats["calls"] += 1
                if not good:
                    stats["bad"] += 1
                    print(f"  MISMATCH host call: thread {i} offset {o} size {b} max|diff| {np.abs(dst - want).max():.3e}", flush=True)

    _lib.check(m._lib.msiren_set_streams(m._h, 1))
    ths = [threading.Thread(target=host_worker, args=(i,)) for i in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    print(f"{cfg}: {stats['calls']} host-pointer calls from two threads in {secs / 2:.0f} s, {stats['bad']} mismatches", flush=True)
    ok = ok and stats["bad"] == 0
print("SOAK OK" if ok else "SOAK FAILED")
sys.exit(0 if ok else 1)
