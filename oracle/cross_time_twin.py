#!/usr/bin/env python3
"""Cross-calibration of the timed CPU baseline (build container only).

bench.py's `cpu_baseline` times oracle/torch_twin.py ("port") on the GPU box, where the reference itself cannot travel.  Here, where
/root/reference is importable, the reference's own `ModulatedSiren.forward` and the twin run on the SAME 400 tiles of one 320x320 slice, the
same weights, the same cores and thread count, best of 5 after 2 warm-ups: the ratio says how good a stand-in the twin is as a TIMING
baseline, the difference of the outputs that it computes the same thing.  Writes one text record to stdout.
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle import gen_fixtures as gf  # noqa: E402  (its import recipe: stand-in modules for five unused third-party imports)
from oracle import torch_twin as tw  # noqa: E402


def best_of(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), sum(ts) / len(ts), out


def main():
    import torch

    from mri_inr_amd import synthetic as syn
    from oracle import siren_oracle as orc

    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    ModulatedSiren, FixedAutoencoder, _ = gf._import_reference()
    print(f"torch {torch.__version__}, {threads} threads on {os.cpu_count()} CPUs")
    for act in ("sine", "morlet"):
        sd = syn.make_state_dict(seed=7, trained_like=True)
        model = gf._build_reference_model(ModulatedSiren, FixedAutoencoder, sd, H=256, L=5, Z=256, S=24, activation=act)
        img = syn.make_slice(0)
        tiles, _ = orc.image_to_patches(img, 32, 16)
        tiles = np.ascontiguousarray(tiles, np.float32)
        x = torch.from_numpy(tiles)
        t = tw.to_tensors(sd)
        with torch.no_grad():
            ref_best, ref_mean, ref_out = best_of(lambda: model(x))
        twin_best, twin_mean, twin_out = best_of(lambda: tw.forward_tiles(t, x, num_layers=5, activation=act))
        ref_out, twin_out = ref_out.numpy(), np.asarray(twin_out)
        err = float(np.abs(twin_out - ref_out).max() / np.abs(ref_out).max())
        px = 320 * 320 / 1e6
        print(f"{act}: reference ModulatedSiren.forward  best {ref_best:.3f} s  mean {ref_mean:.3f} s  = {px / ref_best:.4f} Mpixel/s")
        print(f"{act}: oracle/torch_twin.forward_tiles   best {twin_best:.3f} s  mean {twin_mean:.3f} s  = {px / twin_best:.4f} Mpixel/s"
              f"   twin / reference time = {twin_best / ref_best:.3f};  max|twin - ref| / max|ref| = {err:.2e}")


if __name__ == "__main__":
    main()
