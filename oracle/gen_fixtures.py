#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

Imports MatteoWohlrapp/mri-inr from /root/reference (read-only, never copied), pushes weights and
inputs drawn from ``mri_inr_amd.synthetic`` (numpy RNG => reproducible from seeds) through the
reference's own ``ModulatedSiren`` / tiling / configuration code on the CPU, and stores the
outputs.  The GPU box has no /root/reference: only the committed .npz files travel.

Five third-party modules that the reference imports transitively but never touches on this path
are absent from the image (tensorboard, polars, fastmri, skimage, seaborn); empty stand-in
modules are registered for them before the import.  Nothing of the reference's
behaviour on the hot path depends on them.

Usage:  python oracle/gen_fixtures.py            (writes tests/golden/)
"""

from __future__ import annotations

import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True


def _stub_missing_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return _Dummy()

        def __call__(self, *a, **k):
            return _Dummy()

    try:
        import tensorboard  # noqa: F401
    except Exception:
        mod("tensorboard")
        tb = mod("torch.utils.tensorboard", SummaryWriter=_Dummy)
        import torch.utils

        torch.utils.tensorboard = tb

    class _Cfg:
        @staticmethod
        def set_tbl_rows(*a, **k):
            pass

    for name, attrs in (
        ("polars", dict(Config=_Cfg, LazyFrame=_Dummy, DataFrame=_Dummy)),
        ("fastmri", {}),
        ("fastmri.data", {}),
        ("fastmri.data.transforms", {}),
        ("fastmri.data.subsample", dict(RandomMaskFunc=_Dummy)),
        ("seaborn", {}),
        ("skimage", {}),
        ("skimage.metrics", dict(normalized_root_mse=None, peak_signal_noise_ratio=None,
                                 structural_similarity=None)),
        ("h5py", {}),
    ):
        try:
            importlib.import_module(name)
        except Exception:
            mod(name, **attrs)


def _import_reference():
    _stub_missing_modules()
    sys.path.insert(0, REF)
    from src.networks.modulated_siren import ModulatedSiren  # type: ignore
    from src.networks.encoding.siren_encoder import FixedAutoencoder  # type: ignore
    from src.util import tiling as ref_tiling  # type: ignore
    return ModulatedSiren, FixedAutoencoder, ref_tiling


def _build_reference_model(ModulatedSiren, FixedAutoencoder, sd, *, H, L, Z, S, activation,
                           w0=1.0, w0_initial=30.0, use_bias=True):
    import torch

    with tempfile.TemporaryDirectory() as td:
        ckpt = os.path.join(td, "enc.pth")
        torch.save({"state_dict": FixedAutoencoder().state_dict()}, ckpt)
        model = ModulatedSiren(
            dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=Z, w0=w0,
            w0_initial=w0_initial, use_bias=use_bias, dropout=0.1, modulate=True,
            encoder_type="custom", encoder_path=ckpt, outer_patch_size=32, inner_patch_size=16,
            siren_patch_size=S, device=torch.device("cpu"), activation=activation)
    tsd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(tsd, strict=True), None
    model.eval()
    return model


def _run_net(model, mods):
    """reference SirenNet over its own grid buffer, exactly as ModulatedSiren.forward drives it."""
    import torch

    with torch.no_grad():
        B = mods.shape[1]
        coords = model.grid.clone().detach().repeat(B, 1, 1)
        out = model.net(coords, tuple(torch.from_numpy(mods[l].copy()) for l in range(mods.shape[0])))
    return out.squeeze(2).numpy()


def _run_net_layers(model, mods):
    import torch

    hid = []
    with torch.no_grad():
        B = mods.shape[1]
        x = model.grid.clone().detach().repeat(B, 1, 1)
        for l, layer in enumerate(model.net.layers):
            x = layer(x)
            x *= torch.from_numpy(mods[l].copy())[:, None, :]
            hid.append(x.numpy().copy())
        out = model.net.last_layer(x).squeeze(2).numpy()
    return hid, out


VARIANTS = {  # name: model hyper-parameters off the YAML defaults
    "w0": dict(H=256, L=5, Z=256, S=24, activation="sine", w0=2.0, w0_initial=15.0, use_bias=True),
    "nobias": dict(H=256, L=5, Z=256, S=24, activation="sine", w0=1.0, w0_initial=30.0, use_bias=False),
    "small": dict(H=128, L=3, Z=256, S=10, activation="sine", w0=1.0, w0_initial=30.0, use_bias=True),
    "morlet_w0": dict(H=256, L=5, Z=256, S=24, activation="morlet", w0=1.5, w0_initial=20.0, use_bias=True),
    "deep": dict(H=256, L=8, Z=256, S=24, activation="sine", w0=1.0, w0_initial=30.0, use_bias=True),
}


def write_model_variants(ModulatedSiren, FixedAutoencoder, torch):
    """(2b) the reference on hyper-parameters other than the YAML defaults: non-unit frequencies, no biases, a
    small and a deeper network, Morlet with non-unit frequencies -- trunk on seeded modulations and full forward
    on seeded tiles (6 each)."""
    from mri_inr_amd import synthetic as syn

    store = {}
    for i, (name, v) in enumerate(VARIANTS.items()):
        sd = syn.make_state_dict(seed=21 + i, dim_hidden=v["H"], num_layers=v["L"], latent_dim=v["Z"],
                                 siren_patch_size=v["S"], use_bias=v["use_bias"], trained_like=True)
        model = _build_reference_model(ModulatedSiren, FixedAutoencoder, sd, **v)
        mods = syn.make_mods(50 + i, v["L"], 6, v["H"])
        store[f"{name}_trunk"] = _run_net(model, mods)
        tiles = np.random.default_rng(60 + i).random((6, 32, 32), dtype=np.float32)
        with torch.no_grad():
            store[f"{name}_forward"] = model(torch.from_numpy(tiles)).numpy()
    store["meta"] = np.array(json.dumps(VARIANTS))
    np.savez_compressed(os.path.join(GOLD, "model_variants.npz"), **store)
    return sorted(store)


def write_reference_grid(ModulatedSiren, FixedAutoencoder, torch):
    """(2c) the reference's OWN coordinate buffer.  Every other fixture loads this build's grid into the reference with strict=True; here
    "grid" is dropped from the state_dict and the rest loaded with strict=False, so what `ModulatedSiren.__init__` registers
    (linspace + meshgrid(indexing="ij"), modulated_siren.py:427-433) produces the output: the buffer itself for S = 24 (every YAML),
    10 and 16, and forward(tiles) / the trunk on seeded modulations with it at S = 24 (sine and Morlet)."""
    from mri_inr_amd import synthetic as syn

    store = {}
    H, L, Z = 256, 5, 256
    sd = {k: v for k, v in syn.make_state_dict(seed=7, trained_like=True).items() if k != "grid"}
    tiles = np.random.default_rng(1).random((7, 32, 32), dtype=np.float32)
    mods = syn.make_mods(34, L, 5, H)
    for act in ("sine", "morlet"):
        with tempfile.TemporaryDirectory() as td:
            ckpt = os.path.join(td, "enc.pth")
            torch.save({"state_dict": FixedAutoencoder().state_dict()}, ckpt)
            model = ModulatedSiren(dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=Z, w0=1.0, w0_initial=30.0, use_bias=True,
                                   dropout=0.1, modulate=True, encoder_type="custom", encoder_path=ckpt, outer_patch_size=32,
                                   inner_patch_size=16, siren_patch_size=24, device=torch.device("cpu"), activation=act)
        res = model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=False)
        assert list(res.missing_keys) == ["grid"] and not res.unexpected_keys, res
        model.eval()
        store["grid_24"] = model.grid.numpy().copy()
        with torch.no_grad():
            store[f"forward_{act}"] = model(torch.from_numpy(tiles)).numpy()
        store[f"trunk_{act}"] = _run_net(model, mods)
    for S in (10, 16):
        with tempfile.TemporaryDirectory() as td:
            ckpt = os.path.join(td, "enc.pth")
            torch.save({"state_dict": FixedAutoencoder().state_dict()}, ckpt)
            model = ModulatedSiren(dim_in=2, dim_hidden=32, dim_out=1, num_layers=2, latent_dim=16, w0=1.0, w0_initial=30.0, use_bias=True,
                                   dropout=0.1, modulate=True, encoder_type="custom", encoder_path=ckpt, outer_patch_size=32,
                                   inner_patch_size=16, siren_patch_size=S, device=torch.device("cpu"), activation="sine")
        store[f"grid_{S}"] = model.grid.numpy().copy()
    store["meta"] = np.array(json.dumps(dict(seed=7, trained_like=True, tiles_seed=1, B=7, mods_seed=34, mods_B=5, torch=torch.__version__)))
    np.savez_compressed(os.path.join(GOLD, "reference_grid.npz"), **store)
    return sorted(store)


GEOMETRIES = ((8, 16), (16, 32), (32, 32), (16, 20))


def write_tiling_geometries(rt, torch):
    """(4c) tiling / weighted fold for other (inner_patch_size, siren_patch_size) pairs of the YAML surface
    (outer stays 32), on a 96x80 slice: patch geometry, per-patch means and the folded image."""
    from mri_inr_amd import synthetic as syn

    store = {}
    img = syn.make_slice(2, 96, 80, brain_mask=True)
    for inner, S in GEOMETRIES:
        patches, info = rt.image_to_patches(torch.from_numpy(img)[None], 32, inner)
        rec = np.random.default_rng(6).random((patches.shape[0], S, S), dtype=np.float32)
        out = rt.patches_to_image_weighted_average(torch.from_numpy(rec), info, S, inner, torch.device("cpu")).numpy()
        store[f"info_{inner}_{S}"] = np.array(info[0])
        store[f"patch_means_{inner}_{S}"] = patches.numpy().reshape(patches.shape[0], -1).mean(1, dtype=np.float64)
        store[f"wfold_{inner}_{S}"] = np.squeeze(out)
    np.savez_compressed(os.path.join(GOLD, "tiling_geometries.npz"), **store)
    return sorted(store)


def main():
    import torch

    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    from mri_inr_amd import synthetic as syn

    ModulatedSiren, FixedAutoencoder, rt = _import_reference()
    manifest = {"torch": torch.__version__, "numpy": np.__version__, "cases": {}}

    # ---- (1) tiny full-tensor cases: per-layer activations, sine + morlet -----------------
    for act in ("sine", "morlet"):
        H, L, Z, S, B = 32, 3, 256, 8, 3
        sd = syn.make_state_dict(seed=11, dim_hidden=H, num_layers=L, latent_dim=Z, siren_patch_size=S)
        model = _build_reference_model(ModulatedSiren, FixedAutoencoder, sd, H=H, L=L, Z=Z, S=S, activation=act)
        mods = syn.make_mods(21, L, B, H)
        hid, out = _run_net_layers(model, mods)
        np.savez_compressed(os.path.join(GOLD, f"tiny_{act}.npz"), out=out,
                            **{f"hidden{l}": h for l, h in enumerate(hid)},
                            meta=json.dumps(dict(H=H, L=L, Z=Z, S=S, B=B, seed=11, mods_seed=21, activation=act)))
        manifest["cases"][f"tiny_{act}"] = dict(H=H, L=L, S=S, B=B)

    # ---- (2) default-shape trunk: B=1 and B=64, three modulation regimes, sine + morlet ---
    H, L, Z, S = 256, 5, 256, 24
    sd = syn.make_state_dict(seed=7)
    for act in ("sine", "morlet"):
        model = _build_reference_model(ModulatedSiren, FixedAutoencoder, sd, H=H, L=L, Z=Z, S=S, activation=act)
        store = {}
        # (a) U(0.5,1.5) "trained-like"
        for B, mseed in ((1, 31), (64, 32)):
            mods = syn.make_mods(mseed, L, B, H)
            store[f"uniform_B{B}"] = _run_net(model, mods)
        # (b) with exact zeros (ReLU-like sparsity)
        mods = syn.make_mods(33, L, 16, H, lo=0.0, hi=2.0, zero_fraction=0.5)
        store["sparse_B16"] = _run_net(model, mods)
        # (c) reference default-init modulator + encoder on random tiles
        tiles = np.random.default_rng(41).random((16, 32, 32), dtype=np.float32)
        with torch.no_grad():
            z = model.encoder(torch.from_numpy(tiles))
            m = model.modulator(z)
            mods_c = np.stack([t.numpy() for t in m], 0)
            store["modulator_mods"] = mods_c
            store["modulator_latent"] = z.numpy()
            store["modulator_B16"] = _run_net(model, mods_c)
        np.savez_compressed(os.path.join(GOLD, f"trunk_{act}.npz"), **store,
                            meta=json.dumps(dict(H=H, L=L, Z=Z, S=S, seed=7, activation=act,
                                                 mods=dict(uniform_B1=31, uniform_B64=32, sparse_B16=33),
                                                 tiles_seed=41)))
        manifest["cases"][f"trunk_{act}"] = sorted(store)

    # ---- (3) full forward(tiles), default-init and O(1)-modulation ("trained-like") weights ----
    for name, tl in (("default", False), ("trained", True)):
        sdg = syn.make_state_dict(seed=7, trained_like=tl)
        for act in ("sine", "morlet"):
            model = _build_reference_model(ModulatedSiren, FixedAutoencoder, sdg, H=H, L=L, Z=Z, S=S, activation=act)
            tiles = np.random.default_rng(42).random((8, 32, 32), dtype=np.float32)
            with torch.no_grad():
                out = model(torch.from_numpy(tiles)).numpy()
                z = model.encoder(torch.from_numpy(tiles)).numpy()
                mods = np.stack([t.numpy() for t in model.modulator(torch.from_numpy(z))], 0)
            np.savez_compressed(os.path.join(GOLD, f"forward_{name}_{act}.npz"), out=out, latent=z, mods=mods,
                                meta=json.dumps(dict(seed=7, trained_like=tl, tiles_seed=42, B=8, activation=act)))
            manifest["cases"][f"forward_{name}_{act}"] = dict(out=list(out.shape), absmax=float(np.abs(out).max()))

    # ---- (4) tiling ---------------------------------------------------------------------------
    store = {}
    for name, (hh, ww) in (("320x320", (320, 320)), ("70x50", (70, 50))):
        img = syn.make_slice(3, hh, ww, brain_mask=(name == "320x320"))
        patches, info = rt.image_to_patches(torch.from_numpy(img)[None], 32, 16)
        store[f"patches_{name}"] = patches.numpy()
        store[f"info_{name}"] = np.array(info[0])
        kept, black, shape = rt.filter_and_remember_black_patches(patches)
        store[f"black_{name}"] = np.array(black, dtype=np.int64)
        rec = np.random.default_rng(5).random((patches.shape[0], 24, 24), dtype=np.float32)
        rec_t = torch.from_numpy(rec)
        store[f"wfold_{name}"] = rt.patches_to_image_weighted_average(rec_t, info, 24, 16, torch.device("cpu")).numpy()
        store[f"fold_{name}"] = rt.patches_to_image(patches, info, 32, 16).numpy()
        keep = [i for i in range(patches.shape[0]) if i not in black]
        store[f"reint_{name}"] = rt.reintegrate_black_patches(rec_t[keep], black, shape).numpy()
        store[f"center_{name}"] = rt.extract_center_batch(patches, 32, 24).numpy()
    store["weight_matrix_24"] = rt.generate_weight_matrix(24).numpy()
    store["weight_matrix_32"] = rt.generate_weight_matrix(32).numpy()
    np.savez_compressed(os.path.join(GOLD, "tiling.npz"), **store)
    manifest["cases"]["tiling"] = sorted(store)

    manifest["cases"]["tiling_geometries"] = write_tiling_geometries(rt, torch)
    manifest["cases"]["model_variants"] = write_model_variants(ModulatedSiren, FixedAutoencoder, torch)
    manifest["cases"]["reference_grid"] = write_reference_grid(ModulatedSiren, FixedAutoencoder, torch)

    # ---- (4b) whole-slice reconstruction as metrics_error drives it (error.py:231-249) --------
    sdg = syn.make_state_dict(seed=7, trained_like=True)
    model = _build_reference_model(ModulatedSiren, FixedAutoencoder, sdg, H=H, L=L, Z=Z, S=S, activation="sine")
    img = syn.make_slice(0, 160, 128, brain_mask=True)
    with torch.no_grad():
        patches, info = rt.image_to_patches(torch.from_numpy(img)[None], 32, 16)
        kept, black, shape = rt.filter_and_remember_black_patches(patches)
        rec = model(kept)
        rec = rt.reintegrate_black_patches(rec, black, shape)
        image = rt.patches_to_image_weighted_average(rec, info, 24, 16, torch.device("cpu"))
    np.savez_compressed(os.path.join(GOLD, "slice_recon.npz"), image=image.numpy(), black=np.array(black),
                        info=np.array(info[0]),
                        meta=json.dumps(dict(seed=7, trained_like=True, slice=0, H=160, W=128, brain_mask=True)))
    manifest["cases"]["slice_recon"] = dict(image=list(image.shape), n_black=len(black))

    # ---- (5) configuration loader: every shipped YAML through load_configuration ---------------
    cfgs = {}
    ydir = os.path.join(REF, "configuration")
    for root, _, files in os.walk(ydir):
        for f in sorted(files):
            if not f.endswith(".yaml"):
                continue
            path = os.path.join(root, f)
            rel = os.path.relpath(path, ydir)
            testing = os.path.basename(f).startswith("test")
            # the reference's merge mutates its module-level defaults: reload for a clean slate
            import src.configuration.configuration as rc  # type: ignore
            rc = importlib.reload(rc)
            try:
                import yaml
                with open(path) as fh:
                    user = yaml.safe_load(fh)  # the loader's input, as data
                ns = rc.load_configuration(path, testing=testing)
                cfgs[rel] = dict(testing=testing, user=user, config=rc.namespace_to_dict(ns))
            except Exception as e:  # pragma: no cover
                cfgs[rel] = dict(testing=testing, error=f"{type(e).__name__}: {e}")
    with open(os.path.join(GOLD, "configs.json"), "w") as fh:
        json.dump(cfgs, fh, indent=1, sort_keys=True)
    manifest["cases"]["configs"] = sorted(cfgs)

    with open(os.path.join(GOLD, "MANIFEST.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(GOLD)))


if __name__ == "__main__":
    if sys.argv[1:] in (["geometries"], ["variants"], ["grid"]):  # add one file without regenerating the others
        sys.path.insert(0, REPO)
        _ms, _ae, _rt = _import_reference()
        import torch
        torch.manual_seed(0)
        torch.set_num_threads(8)
        if sys.argv[1] == "geometries":
            key, keys = "tiling_geometries", write_tiling_geometries(_rt, torch)
        elif sys.argv[1] == "grid":
            key, keys = "reference_grid", write_reference_grid(_ms, _ae, torch)
        else:
            key, keys = "model_variants", write_model_variants(_ms, _ae, torch)
        mpath = os.path.join(GOLD, "MANIFEST.json")
        man = json.load(open(mpath))
        man["cases"][key] = keys
        json.dump(man, open(mpath, "w"), indent=1, sort_keys=True)
        print("wrote", key, keys)
    else:
        main()
