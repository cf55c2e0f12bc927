"""Encoder tail + Modulator in split-fp16 arithmetic (csrc/encoder_modulator_f16x3.hip.h) against the fp64 oracle and against the
exact-fp32 launches per layer (MSIREN_PROLOGUE_F16X3=0): latent, modulations, outputs; batch sizes on every threshold; input scales
from 1e-6 (raw fastMRI intensities) to 1e4; batch invariance bit for bit."""
import os, subprocess, sys, json
import numpy as np
sys.path.insert(0, '.')
from oracle import siren_oracle as orc


def nerr(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-300))


def make(H=256, Z=256, L=5, prec="f16x3", residual=False, trained_like=True):
    from mri_inr_amd import ModulatedSiren, synthetic as syn
    sd = syn.make_state_dict(seed=7, dim_hidden=H, num_layers=L, latent_dim=Z, trained_like=trained_like)
    m = ModulatedSiren(2, H, 1, L, Z, 1.0, 30.0, True, 0.1, True, "custom", None, 32, 16, 24, "cuda", "sine", precision=prec, residual=residual)
    m.load_state_dict(sd); m.to("cuda")
    return m, sd


def run(tag):
    m, sd = make()
    rng = np.random.default_rng(0)
    res = {}
    for scale in (1.0, 1e-5, 1e4):
        for B in (1, 7, 16, 17, 48, 400, 1030):
            t = (rng.random((B, 32, 32), dtype=np.float32) * scale).astype(np.float32)
            z = m.encoder(t)
            mods = np.stack(m.modulator(z), 0)
            out = m(t)
            z64 = orc.encoder_forward(sd, t, dtype=np.float64)
            m64 = orc.modulator_forward(sd, z, num_layers=5, dtype=np.float64)   # from the device's latent: the Modulator alone
            o64 = orc.modulated_siren_forward(sd, t, num_layers=5, dtype=np.float64)
            # batch invariance: the first min(B, 5) tiles alone
            k = min(B, 5)
            same = bool(np.array_equal(m.encoder(t[:k]), z[:k]) and np.array_equal(m(t[:k]), out[:k]))
            res[f"{scale:g}/{B}"] = (nerr(z, z64), nerr(mods, m64), nerr(out.reshape(B, -1), o64.reshape(B, -1)), same)
            print(tag, f"scale {scale:g} B {B:5d}: latent {res[f'{scale:g}/{B}'][0]:.2e} mods {res[f'{scale:g}/{B}'][1]:.2e} out {res[f'{scale:g}/{B}'][2]:.2e} batch-invariant {same}", flush=True)
    return res


if __name__ == "__main__":
    run("f16x3-prologue" if os.environ.get("MSIREN_PROLOGUE_F16X3", "1") != "0" else "fp32-prologue ")
