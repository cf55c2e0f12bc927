"""the evaluation driver end to end on the GPU.  `test_mod_siren.py --config X.yaml` (same CLI
as the reference's script) must write the reference's artefacts (metrics_error.csv, metrics_summary.txt; formats:
test_mod_siren.py:38-75,236-247 of the reference) and score the reconstruction against the FOLDED fully-sampled tiles
(error.py:251-254), also on sizes that are not a multiple of the stride.  PSNR / NRMSE are checked against values
computed from the fp64 oracle's reconstruction; SSIM is a restatement of scikit-image's definition (not installed
here): PARITY UNPINNED, only checked for self-consistency."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from mri_inr_amd import metrics, synthetic as syn
from oracle import siren_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

YAML = """
data:
  dataset: {data}
  metric_samples: 0
  visual_samples: 1
  acceleration: 6
  center_fraction: 0.05
model:
  dim_in: 2
  dim_hidden: 256
  dim_out: 1
  latent_dim: 256
  num_layers: 5
  w0: 1.0
  w0_initial: 30.0
  use_bias: true
  dropout: 0.1
  encoder_type: custom
  encoder_path: null
  outer_patch_size: 32
  inner_patch_size: 16
  siren_patch_size: 24
  activation: sine
testing:
  output_dir: {out}
  output_name: run
  model_path: synthetic
"""


def test_eval_driver_artefacts_and_scores(tmp_path):
    data = tmp_path / "data"
    data.mkdir()
    sizes = {"a_square": (320, 320), "b_ragged": (200, 136), "c_small": (70, 50)}
    pairs = {}
    for k, (name, (hh, ww)) in enumerate(sizes.items()):
        full = syn.make_slice(10 + k, hh, ww, brain_mask=(name == "a_square"))
        under = ((full + np.roll(full, 1, 1) + np.roll(full, -1, 1)) / np.float32(3)).astype(np.float32)
        np.save(data / f"{name}_fully.npy", full)
        np.save(data / f"{name}_under.npy", under)
        pairs[name] = (full, under)
    cfg = tmp_path / "eval.yaml"
    cfg.write_text(YAML.format(data=data, out=tmp_path / "out"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "test_mod_siren.py"), "--config", str(cfg)], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = tmp_path / "out" / "run" / "test"

    rows = (out / "metrics_error.csv").read_text().splitlines()
    assert rows[0] == "FILENAME,PSNR,SSIM,NRMSE" and len(rows) == 1 + len(sizes)
    got = {}
    for line in rows[1:]:
        name, psnr, ssim, nrmse = line.split(",")
        got[name] = (float(psnr), float(ssim), float(nrmse))
    assert sorted(got) == sorted(sizes)

    cfg_txt = (out / "config.txt").read_text().splitlines()   # save_args_to_file (reference test_mod_siren.py:18-33): "name: value"
    assert any(l.startswith("model: ") for l in cfg_txt) and any(l.startswith("testing: ") for l in cfg_txt), cfg_txt

    summary = (out / "metrics_summary.txt").read_text()
    m = re.fullmatch(r"(?:(?:PSNR|SSIM|NRMSE):\n(?:  (?:mean|std|min|max): \S+\n){4}\n){3}", summary)
    assert m, summary
    assert summary.index("PSNR:") < summary.index("SSIM:") < summary.index("NRMSE:")
    mean_psnr = float(re.search(r"PSNR:\n  mean: (\S+)", summary).group(1))
    assert abs(mean_psnr - np.mean([v[0] for v in got.values()])) < 1e-9

    sd = syn.make_state_dict(seed=7, trained_like=True)
    for name, (full, under) in pairs.items():
        rec = orc.reconstruct_slice(sd, under, num_layers=5, dtype=np.float64)
        tiles, info = orc.image_to_patches(full, 32, 16)
        ref = orc.patches_to_image(tiles, info, 32, 16)          # what error.py:251-254 scores against
        assert ref.shape == rec.shape == (-(-full.shape[0] // 16) * 16, -(-full.shape[1] // 16) * 16)
        psnr, ssim, nrmse = got[name]
        assert abs(psnr - metrics.calculate_psnr(ref, rec)) < 2e-3, name            # dB
        assert abs(nrmse - metrics.calculate_nrmse(ref, rec)) < 1e-4 * nrmse, name
        assert abs(ssim - metrics.calculate_ssim(ref, rec)) < 1e-4 and -1.0 <= ssim <= 1.0   # self-consistency only
        if name != "a_square":
            # the scored image includes the reflect-padded rim beyond the raw slice (ragged sizes)
            hh, ww = full.shape
            assert np.allclose(ref[:hh, :ww], full, atol=1e-6)
            assert ref.shape != full.shape

    # data.visual_samples = 1: the first slice's image folder (reference test_mod_siren.py:122-173, error.py:104-183)
    vis = out / "a_square"
    full, under = pairs["a_square"]
    rec = orc.reconstruct_slice(sd, under, num_layers=5, dtype=np.float64)
    got_rec = np.load(vis / "a_square_reconstructed.npy")
    assert got_rec.shape == rec.shape and np.abs(got_rec - rec).max() <= 1e-4 * np.abs(rec).max()
    t_u, info = orc.image_to_patches(under, 32, 16)
    assert np.allclose(np.load(vis / "a_square_undersampled.npy"), orc.patches_to_image(t_u, info, 32, 16), atol=1e-6)
    t_f, _ = orc.image_to_patches(full, 32, 16)
    ref_full = orc.patches_to_image(t_f, info, 32, 16)
    assert np.allclose(np.load(vis / "a_square_fully_sampled.npy"), ref_full, atol=1e-6)
    assert np.allclose(np.load(vis / "a_square_difference.npy"), np.abs(ref_full - got_rec), atol=1e-6)
    for kind in ("reconstructed", "undersampled", "fully_sampled", "difference", "comparison"):
        assert (vis / f"a_square_{kind}.png").stat().st_size > 0
    err = dict(l.split(": ") for l in (vis / "a_square_error.txt").read_text().splitlines())       # error.py:185-197
    assert list(err) == ["PSNR", "SSIM", "NRMSE"]
    assert abs(float(err["PSNR"]) - metrics.calculate_psnr(ref_full, got_rec)) < 1e-9 and abs(float(err["NRMSE"]) - metrics.calculate_nrmse(ref_full, got_rec)) < 1e-12
    # the box and density plots of the metric samples, under the reference's file names (test_mod_siren.py:248-256, visualization.py:145, :164)
    for key in ("PSNR", "SSIM", "NRMSE"):
        for name in (f"{key}_metrics_boxplot.png", f"{key}_density_plot.png"):
            assert (out / name).stat().st_size > 1000, name


def test_harness_mirror_matches_oracle_tiling():
    from mri_inr_amd import ModulatedSiren, harness

    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=5, latent_dim=256, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda", activation="sine")
    m.load_state_dict(sd)
    m.to("cuda").eval()
    imgs = np.stack([syn.make_slice(k, 120, 88, brain_mask=(k == 0)) for k in range(2)])
    harness.bind(m)
    tiles, info = harness.image_to_patches(imgs, 32, 16)                       # the reference's positional signature (tiling.py:10)
    t0, i0 = orc.image_to_patches(imgs[0], 32, 16)
    assert info == [tuple(i0)] * 2 and np.array_equal(tiles[: t0.shape[0]], t0)
    back = harness.patches_to_image(tiles, info, 32, 16)                       # tiling.py:143
    assert back.shape == (2, 128, 96)
    assert np.abs(back[0] - orc.patches_to_image(t0, i0, 32, 16)).max() < 1e-6
    rec = harness.reconstruct_from_patches(m, tiles, info)
    assert np.array_equal(rec, m.reconstruct(imgs))              # same chain as the slice entry point, bit for bit
    with pytest.raises(ValueError):
        harness.image_to_patches(imgs, 32, 8)


def test_harness_black_filter_steps_one_by_one():
    """The reference's callers spell the chain step by step (training.py:438-445, error.py:132-152):
    filter_and_remember_black_patches -> model -> reintegrate_black_patches -> patches_to_image_weighted_average.  Each step
    against the oracle's restatement, the composition bit for bit against the one-call form; numpy in -> numpy out and
    DeviceArray in -> DeviceArray out."""
    from mri_inr_amd import ModulatedSiren, harness

    sd = syn.make_state_dict(seed=7, trained_like=True)
    m = ModulatedSiren(dim_in=2, dim_hidden=256, dim_out=1, num_layers=5, latent_dim=256, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda", activation="sine")
    m.load_state_dict(sd)
    m.to("cuda").eval()
    harness.bind(m)
    img = syn.make_slice(4, 200, 136, brain_mask=True)
    img[:40] = 0.0
    tiles, info = harness.image_to_patches(img[None], 32, 16)
    tiles[5] = 1e-12                                              # mean < 1e-10 counts as black (tiling.py:184-198)
    kept, black, shape = harness.filter_and_remember_black_patches(tiles)
    o_kept, o_black, o_shape = orc.filter_and_remember_black_patches(tiles)
    assert black == [int(i) for i in o_black] and 5 in black and 0 < len(black) < tiles.shape[0]
    assert tuple(shape) == tuple(o_shape) == tiles.shape and np.array_equal(kept, o_kept)
    out = m(kept)                                                 # (n_keep, 24, 24)
    full = harness.reintegrate_black_patches(out, black, shape)
    assert full.shape == (tiles.shape[0], 24, 24)
    assert np.array_equal(full, orc.reintegrate_black_patches(out, o_black, (tiles.shape[0], 24, 24)))
    rec = harness.patches_to_image_weighted_average(full, info, 24, 16, "cuda")
    assert np.abs(rec[0] - orc.patches_to_image_weighted_average(full, info[0], 24, 16)).max() < 1e-6
    assert np.array_equal(rec, harness.reconstruct_from_patches(m, tiles, info))   # the one-call form: same bits
    # device-resident: nothing but the index list visits the host
    d_tiles = m.device_array(tiles.shape).copy_from(tiles)
    d_kept, black2, shape2 = harness.filter_and_remember_black_patches(d_tiles)
    assert black2 == black and tuple(shape2) == tuple(shape) and np.array_equal(d_kept.numpy(), kept)
    d_full = harness.reintegrate_black_patches(m.device_array(out.shape).copy_from(out), black2, shape2)
    assert np.array_equal(d_full.numpy(), full)
    assert np.array_equal(harness.patches_to_image_weighted_average(d_full, info, 24, 16, "cuda").numpy(), rec)
    # all black / none black
    z_kept, z_black, _ = harness.filter_and_remember_black_patches(np.zeros((3, 32, 32), np.float32))
    assert z_kept.shape == (0, 32, 32) and z_black == [0, 1, 2]
    back = harness.reintegrate_black_patches(np.zeros((0, 24, 24), np.float32), z_black, (3, 32, 32))
    assert back.shape == (3, 24, 24) and not back.any()
    ones = np.ones((2, 32, 32), np.float32)
    n_kept, n_black, _ = harness.filter_and_remember_black_patches(ones)
    assert n_black == [] and np.array_equal(n_kept, ones)
    with pytest.raises(ValueError):
        harness.reintegrate_black_patches(out[:-1], black, shape)


def test_config5_masked_slice_pipeline_same_bits_as_step_by_step():
    """BASELINE config 5 (10 x 512, bf16, residual) through the slice pipeline: the number of non-black tiles is known to the
    device only, so the weight-stationary trunk lays its passes out from a count it reads itself (x1w_schedule on plan[1]); the
    one-call form must give the bits of the step-by-step chain, whose model call knows its batch on the host."""
    from mri_inr_amd import ModulatedSiren, harness

    H, L, Z = 512, 10, 128
    sd = syn.make_state_dict(seed=21, dim_hidden=H, num_layers=L, latent_dim=Z, modulator_bias_center=0.25, encoder_gain=10.0)
    m = ModulatedSiren(dim_in=2, dim_hidden=H, dim_out=1, num_layers=L, latent_dim=Z, w0=1.0, w0_initial=30.0,
                       use_bias=True, dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None,
                       outer_patch_size=32, inner_patch_size=16, siren_patch_size=24, device="cuda", activation="sine",
                       residual=True, precision="bf16")
    m.load_state_dict(sd)
    m.to("cuda").eval()
    harness.bind(m)
    imgs = np.stack([syn.make_slice(k, 320, 320, brain_mask=True) for k in (2, 5, 8)])
    rec = m.reconstruct(imgs)
    assert m.last_trunk_kernel().startswith("siren_trunk_x1w_kernel") and np.isfinite(rec).all() and rec.std() > 0
    for k in range(3):
        assert np.array_equal(m.reconstruct(imgs[k]), rec[k])
    tiles, info = harness.image_to_patches(imgs, 32, 16)
    kept, black, shape = harness.filter_and_remember_black_patches(tiles)
    assert 0 < len(black) < tiles.shape[0]
    full = harness.reintegrate_black_patches(m(kept), black, shape)
    assert np.array_equal(harness.patches_to_image_weighted_average(full, info, 24, 16, "cuda"), rec)
    # and again right away: the pass counter was left where the next launch expects it
    assert np.array_equal(m.reconstruct(imgs), rec)
    assert np.array_equal(m(kept[:57]), full[[i for i in range(tiles.shape[0]) if i not in set(black)][:57]])
    # nothing but black tiles: the trunk is launched on an upper bound and finds zero units to do
    assert not m.reconstruct(np.zeros((2, 320, 320), np.float32)).any()
    assert np.array_equal(m.reconstruct(imgs), rec)
    assert m(np.zeros((0, 32, 32), np.float32)).shape == (0, 24, 24)
