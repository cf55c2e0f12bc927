"""CPU tests of host-side pieces that need no GPU: synthetic inputs, metrics, weight files."""
import numpy as np
import pytest

from mri_inr_amd import metrics, synthetic as syn, weights


def test_grid_matches_torch_linspace_meshgrid():
    """The `grid` buffer of the reference: stack(meshgrid(linspace(-1,1,S), linspace(-1,1,S), 'ij'))
    (src/networks/modulated_siren.py:427-433).  torch's CPU linspace is vectorised (base + step*lane per SIMD
    chunk), so its last bit depends on the host's vector width; real checkpoints carry the buffer in their
    state_dict, which is what the library uses.  The synthetic grid agrees to one ulp."""
    import torch

    for S in (2, 3, 8, 24, 25, 64):
        lin = torch.linspace(-1, 1, steps=S)
        ref = torch.stack(torch.meshgrid(lin, lin, indexing="ij"), dim=-1).reshape(S * S, 2).numpy()
        assert np.abs(syn.make_grid(S) - ref).max() <= 1.2e-7, S
        assert syn.make_grid(S)[0].tolist() == [-1.0, -1.0] and syn.make_grid(S)[-1].tolist() == [1.0, 1.0]


def test_state_dict_keys_and_shapes_match_reference_layout():
    sd = syn.make_state_dict(seed=1)
    want = {
        "grid": (576, 2), "net.layers.0.weight": (256, 2), "net.layers.4.weight": (256, 256), "net.layers.4.bias": (256,),
        "net.last_layer.weight": (1, 256), "net.last_layer.bias": (1,), "modulator.layers.0.0.weight": (256, 256),
        "modulator.layers.3.0.weight": (256, 512), "encoder.encoder.encoder.0.weight": (16, 1, 3, 3),
        "encoder.encoder.encoder.4.weight": (64, 32, 8, 8), "encoder.encoder.encoder.7.weight": (256, 64),
    }
    for k, shp in want.items():
        assert sd[k].shape == shp and sd[k].dtype == np.float32, k
    assert len(sd) == 31
    # Siren.init_ ranges (modulated_siren.py:138-142)
    assert np.abs(sd["net.layers.0.weight"]).max() <= 0.5
    assert np.abs(sd["net.layers.2.weight"]).max() <= np.sqrt(6 / 256)
    # same seed -> same weights; the trained-like preset only shifts modulator biases / encoder fc
    sd2 = syn.make_state_dict(seed=1, trained_like=True)
    assert np.array_equal(sd["net.layers.3.weight"], sd2["net.layers.3.weight"])
    assert not np.array_equal(sd["modulator.layers.0.0.bias"], sd2["modulator.layers.0.0.bias"])


def test_slices_are_seeded_and_masked():
    a, b = syn.make_slice(3), syn.make_slice(3)
    assert np.array_equal(a, b) and a.shape == (320, 320) and a.dtype == np.float32
    m = syn.make_slice(3, brain_mask=True)
    assert m[0, 0] == 0 and m[160, 160] == a[160, 160]


def test_metrics_closed_forms():
    rng = np.random.default_rng(0)
    a = rng.random((64, 48))
    b = a + 0.1 * rng.standard_normal(a.shape)
    dr = max(a.max(), b.max()) - min(a.min(), b.min())
    assert metrics.calculate_data_range(a, b) == pytest.approx(dr)
    assert metrics.calculate_psnr(a, b) == pytest.approx(10 * np.log10(dr ** 2 / np.mean((a - b) ** 2)))
    assert metrics.calculate_nrmse(a, b) == pytest.approx(np.linalg.norm(a - b) / np.linalg.norm(a))
    assert metrics.calculate_ssim(a, a) == pytest.approx(1.0)
    assert 0 < metrics.calculate_ssim(a, b) < 1
    assert metrics.calculate_ssim(a, b) > metrics.calculate_ssim(a, a + 0.5 * rng.standard_normal(a.shape))


def test_ssim_against_an_independent_derivation_by_explicit_windows():
    """scikit-image is absent (SSIM stays unpinned against the library); oracle/ssim_windows.py derives the same published definition a
    second way -- explicit 7 x 7 windows that lie fully inside the image, two-pass sample covariance -- and shares no code with
    metrics.py (uniform_filter + crop, E[xy] - E[x]E[y]).  Agreement to 1e-9 on images of the evaluation's kind and on odd sizes."""
    from oracle.ssim_windows import ssim_by_windows

    rng = np.random.default_rng(5)
    for k, shape in enumerate([(320, 320), (64, 48), (7, 7), (9, 31), (33, 8)]):
        a = syn.make_slice(k, *shape).astype(np.float64) if min(shape) >= 32 else rng.random(shape)
        b = a + 0.05 * a.max() * rng.standard_normal(shape)
        if k == 0:
            a, b = a.astype(np.float32), b.astype(np.float32)      # what the harness hands over
        dr = metrics.calculate_data_range(a, b)
        got, want = metrics.calculate_ssim(a, b), ssim_by_windows(a, b, dr)
        assert abs(got - want) < 1e-9, (shape, got, want)
        assert 0 < got < 1


def test_checkpoint_roundtrip_npz_and_pth(tmp_path):
    sd = syn.make_state_dict(seed=5, dim_hidden=32, num_layers=2, latent_dim=16, siren_patch_size=8)
    for name in ("m.npz", "m.pth"):
        path = str(tmp_path / name)
        weights.save_checkpoint(path, sd)
        back = weights.load_checkpoint(path)
        assert set(back) == set(sd)
        assert all(np.array_equal(back[k], sd[k]) for k in sd)


def test_model_object_protocol_without_gpu():
    """Constructor kwargs, attribute names, state_dict protocol and error behaviour on a CPU-only host."""
    from mri_inr_amd import ModulatedSiren

    kw = dict(dim_in=2, dim_hidden=64, dim_out=1, num_layers=3, latent_dim=32, w0=1.0, w0_initial=30.0, use_bias=True,
              dropout=0.1, modulate=True, encoder_type="custom", encoder_path=None, outer_patch_size=32,
              inner_patch_size=16, siren_patch_size=12, device="cuda", activation="morlet")
    m = ModulatedSiren(**kw)
    for attr in ("dim_hidden", "num_layers", "latent_dim", "siren_patch_size", "activation", "grid", "encoder_type",
                 "outer_patch_size", "inner_patch_size", "modulate", "dim_out"):
        assert hasattr(m, attr)
    assert m.grid.shape == (144, 2)
    sd = m.state_dict()
    assert list(sd)[0] == "grid" and sd["net.layers.1.weight"].shape == (64, 64)
    sd["net.layers.1.weight"] = np.zeros((64, 63), np.float32)
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(sd)
    sd = m.state_dict()
    sd["bogus"] = np.zeros(1, np.float32)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        m.load_state_dict(sd)
    assert m.eval() is m and m.training is False
    with pytest.raises(NotImplementedError):
        m.train(True)
    with pytest.raises(ValueError):
        ModulatedSiren(**dict(kw, dim_out=3))
    with pytest.raises(NotImplementedError):
        ModulatedSiren(**dict(kw, encoder_type="vgg"))
    # any other encoder_type builds (like the reference) but has no encoder: forward fails with AttributeError
    m2 = ModulatedSiren(**dict(kw, encoder_type="default"))
    assert not any(k.startswith("encoder.") for k in m2.state_dict())
    with pytest.raises(AttributeError):
        m2(np.zeros((1, 32, 32), np.float32))


def test_committed_bench_line_follows_the_contract():
    """The bench line recorded on the MI355X (profiles/r2/10_final/bench_default.json) carries every key of the
    driver's contract, the roofline and cpu_baseline objects, the host->host / slice->slice extras, and
    self-consistent arithmetic."""
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.loads(open(os.path.join(root, "profiles", "r2", "10_final", "bench_default.json")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "extra"):
        assert k in line, k
    assert line["metric"] == "Mpixels/sec reconstructed (320x320 slice, hidden=256, 5 layers)"
    assert line["unit"] == "Mpixel/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and "configs[1]" in line["config"]["workload"]
    assert "model" not in line["config"]
    # value = pixels of all ranks per step / time per step
    px = line["config"]["slices_per_step_total"] * 320 * 320
    assert abs(line["value"] - px / (line["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * line["value"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    # algorithmic FLOPs per launch = 525 824 per coordinate x 576 coordinates x 400 tiles
    assert r["flops_per_launch"] == 525824 * 576 * 400
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == line["unit"] and c["cores"] >= 1
    e = line["extra"]
    assert 0 < e["host_to_host_mpixel_s"] < line["value"] and e["reconstruct_mpixel_s"] > 0
    # round 4's line (profiles/r4/10_final/bench_driver_like.json): the roofline block names the kernel of the TIMED region as the
    # library reports it, is consistent with `value`, says where its traffic figure comes from; every other BASELINE
    # configuration rides along
    l4 = json.loads(open(os.path.join(root, "profiles", "r4", "10_final", "bench_driver_like.json")).read().strip().splitlines()[-1])
    r4 = l4["roofline"]
    assert r4["kernel"] == "siren_trunk_f16x3n_kernel<0,3,5>" and r4["timed_region_kernels"][0]["kernel"] == r4["kernel"]
    assert abs(r4["avg_launch_ms"] - l4["ms_per_step"]) < 0.02 * l4["ms_per_step"] and r4["flops_per_launch"] == 525824 * 576 * 400
    assert abs(r4["achieved"] - r4["flops_per_launch"] / (r4["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * r4["achieved"]
    assert abs(r4["frac"] - r4["achieved"] / r4["peak"]) < 1e-9 and "profiles/r4/" in r4["traffic_source"] and r4["traffic"] > 0
    assert l4["roofline_kernel_alone"]["kernel"] == "siren_trunk_f16x3w_kernel<0,4>"
    assert l4["config"]["ranks"][0]["pci_bus_id"] and l4["cpu_baseline"]["kind"] == "port"
    for name in ("config3_64_slices_n1", "config3_64_slices_n1_one_stream", "config3_8_slices_per_rank", "config4_morlet", "fp32_trunk",
                 "config5_deep_residual_bf16"):
        c = l4["extra"]["configs"][name]
        assert c["value"] > 0 and 0.1 < c["kernel_alone_frac"] < 1.0 and c["kernel"].startswith("siren_trunk_"), (name, c)
    # round 5's line (profiles/r5/10_final/bench_driver_like.json): the same contract with this round's traffic record; the host-pointer
    # rates on page-locked arrays and at 8 slices per call ride along (never `value`)
    l5 = json.loads(open(os.path.join(root, "profiles", "r5", "10_final", "bench_driver_like.json")).read().strip().splitlines()[-1])
    r5 = l5["roofline"]
    assert r5["kernel"] == "siren_trunk_f16x3n_kernel<0,3,5>" and r5["flops_per_launch"] == 525824 * 576 * 400
    assert abs(l5["value"] - 320 * 320 / (l5["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * l5["value"]
    assert abs(r5["frac"] - r5["achieved"] / r5["peak"]) < 1e-9 and "profiles/r5/" in r5["traffic_source"] and 1e7 < r5["traffic"] < 3e7
    assert l5["roofline_kernel_alone"]["kernel"] == "siren_trunk_f16x3w_kernel<0,4>" and l5["cpu_baseline"]["kind"] == "port"
    e5 = l5["extra"]
    # outputs come from the page-locked pool (stored in place), pageable tiles are copied: between round 4's 211 and the page-locked rate
    assert 250 < e5["host_to_host_mpixel_s"] <= e5["host_to_host_pinned_mpixel_s"] * 1.02 < l5["value"] and e5["host_slice_to_slice_mpixel_s"] > 0
    assert e5["host_to_host_8_slices_mpixel_s"] > e5["host_to_host_mpixel_s"]
    for name in ("config3_64_slices_n1", "config3_64_slices_n1_one_stream", "config3_8_slices_per_rank", "config4_morlet", "fp32_trunk",
                 "config5_deep_residual_bf16"):
        assert e5["configs"][name]["value"] > 0, name
    traffic = json.load(open(os.path.join(root, "profiles", "r5", "traffic.json")))
    assert {k["kernel"] for k in traffic["kernels"]} >= {"siren_trunk_f16x3n_kernel<0,3,5>", "siren_trunk_f16x3w_kernel<0,4>", "latent_mods_f16x3_kernel<2,2,8,3>",
                                                          "siren_trunk_x1w_kernel<1,0,1>"}
    # round 6's line (profiles/r6/06_final/bench_driver_like.json, the driver's command on the final tree): SURVEY 8(d)'s primary region and the
    # strict-fp32 trunk as top-level keys AND as scalars inside config / roofline (a record that keeps only their scalars still has them); the
    # HIP runtime the library was bound to is named; this round's PMC record is the traffic source of the headline kernel
    l6 = json.loads(open(os.path.join(root, "profiles", "r6", "06_final", "bench_driver_like.json")).read().strip().splitlines()[-1])
    assert l6["steps"] == 20 and l6["warmup"] == 5 and l6["n_gpus"] == 1 and l6["scaling"] == "weak" and l6["vs_baseline"] is None
    assert abs(l6["value"] - 320 * 320 / (l6["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * l6["value"]
    r6, c6 = l6["roofline"], l6["config"]
    assert r6["kernel"] == "siren_trunk_f16x3n_kernel<0,3,5>" and abs(r6["frac"] - r6["achieved"] / r6["peak"]) < 1e-9 and 0.5 < r6["frac"] < 0.6
    assert 250 < l6["host_to_host"]["value"] == c6["host_to_host_mpixel_s"] == l6["extra"]["host_to_host_mpixel_s"] < l6["value"]
    assert l6["fp32"]["kernel"] == "siren_trunk_f32_kernel<256,0,0>" and 100 < l6["fp32"]["value"] == r6["fp32_trunk_mpixel_s"] < 115
    assert 0.7 < l6["fp32"]["kernel_alone_frac"] == r6["fp32_trunk_kernel_alone_frac"] < 0.85 and l6["fp32"]["peak_tflops"] == 157.3
    assert c6["hip_runtime"]["torch_bundled"] is False and c6["hip_runtime_version"] == c6["hip_runtime"]["built_against_hip"] and c6["torch_first"] is False
    assert c6["config3_64_slices_n1_mpixel_s"] == l6["extra"]["configs"]["config3_64_slices_n1"]["value"] > 300
    t6 = json.load(open(os.path.join(root, "profiles", "r6", "traffic.json")))
    assert {k["kernel"] for k in t6["kernels"]} == {"siren_trunk_f32_kernel<256,0,0>", "siren_trunk_f16x3n_kernel<0,3,5>"}
    for k in t6["kernels"]:
        assert abs(k["bytes_per_launch"] - (2 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024) < 1 and os.path.exists(os.path.join(root, k["source"]))
    # both runtimes side by side (profiles/r6/02_lines): the same command, torch-free and --torch-first, within 1 % of each other
    a, b = (json.loads(open(os.path.join(root, "profiles", "r6", "02_lines", n)).read().strip().splitlines()[-1]) for n in ("bench_default.json", "bench_torch_first.json"))
    assert a["config"]["hip_runtime"]["torch_bundled"] is False and b["config"]["hip_runtime"]["torch_bundled"] is True
    assert a["config"]["hip_runtime"]["hip_runtime_version"] != b["config"]["hip_runtime"]["hip_runtime_version"]
    assert abs(a["value"] - b["value"]) < 0.01 * a["value"] and abs(a["fp32"]["value"] - b["fp32"]["value"]) < 0.01 * a["fp32"]["value"]
    ref = json.load(open(os.path.join(root, "profiles", "r6", "n1_reference.json")))["config3_64_slices_n1"]
    assert ref["value"] == a["extra"]["configs"]["config3_64_slices_n1"]["value"]
    # the strong-scaling form of BASELINE configs[2] on one GPU, and its 4-rank rehearsal on one card
    for name, n in (("bench_strong64_n1.json", 1), ("bench_strong64_gloo4_one_card.json", 4)):
        s = json.loads(open(os.path.join(root, "profiles", "r2", "10_final", name)).read().strip().splitlines()[-1])
        assert s["scaling"] == "strong" and s["n_gpus"] == n and s["config"]["slices_per_step_total"] == 64
        assert abs(s["value"] - 64 * 320 * 320 / (s["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * s["value"]


def test_metric_plots_have_the_references_names_and_the_density_is_seaborns_default_kde(tmp_path):
    """test_mod_siren.py:248-256 of the reference: one box plot and one density plot per metric, named ``{key}_metrics_boxplot.png`` /
    ``{key}_density_plot.png`` (src/util/visualization.py:145, :164).  The density restates seaborn's ``kdeplot`` defaults (seaborn is
    not installed): Gaussian KDE with Scott's factor n^(-1/5), 200 points from min - 3 bw to max + 3 bw -- checked against the closed
    form, not against scipy's own evaluation."""
    from mri_inr_amd import metric_plots as vis

    rng = np.random.default_rng(3)
    m = {"PSNR": list(rng.normal(31.0, 2.0, 40)), "SSIM": list(rng.uniform(0.7, 0.95, 40)), "NRMSE": [0.12] * 40}   # (zero variance: empty axes)
    vis.metrics_boxplot(m, tmp_path / "out")
    vis.metrics_density_plot(m, tmp_path / "out", suffix="ignored")
    for key in m:
        for name in (f"{key}_metrics_boxplot.png", f"{key}_density_plot.png"):
            p = tmp_path / "out" / name
            assert p.exists() and p.stat().st_size > 1000 and p.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n", name
    v = np.asarray(m["PSNR"])
    x, d = vis.kde_curve(v)
    bw = v.std(ddof=1) * len(v) ** (-0.2)                                   # Scott, one dimension
    assert len(x) == 200 and abs(x[0] - (v.min() - 3 * bw)) < 1e-9 and abs(x[-1] - (v.max() + 3 * bw)) < 1e-9
    want = np.exp(-0.5 * ((x[:, None] - v[None, :]) / bw) ** 2).sum(1) / (len(v) * bw * np.sqrt(2 * np.pi))
    assert np.allclose(d, want, rtol=1e-9, atol=0)
    assert abs(getattr(np, "trapezoid", getattr(np, "trapz", None))(d, x) - 1.0) < 5e-3                                 # (+-3 bw beyond the data holds all but 0.3 % of the mass)
    assert vis.kde_curve(m["NRMSE"]) is None and vis.kde_curve([1.0]) is None


def test_image_writers_mirror_the_references_save_image_and_comparison(tmp_path):
    """src/util/visualization.py:44-110 as visual_error uses them (error.py:160-183): `save_image` writes `{dir}/{filename}.png` of the
    min-max normalised image with a colour bar, `save_image_comparison` four titled panels; a constant image does not produce NaNs."""
    from mri_inr_amd import metric_plots as mp

    rng = np.random.default_rng(5)
    full = rng.random((48, 40)).astype(np.float32) * 3e-5           # fastMRI-like magnitudes: nothing is visible without the normalisation
    under, rec = full * 0.8, full + 1e-6
    n = mp.normalize_scan(full)
    assert n.min() == 0.0 and n.max() == 1.0 and np.allclose(n, (full - full.min()) / (full.max() - full.min()))
    assert not np.isnan(mp.normalize_scan(np.full((4, 4), 2.0))).any()
    mp.save_image(full[None], "slice_fully_sampled", tmp_path / "v")             # (1, H, W) as the reference's tensors: squeezed
    mp.save_image(np.abs(full - rec), "slice_difference", tmp_path / "v", cmap="viridis")
    mp.save_image_comparison(full, under, rec, tmp_path / "v" / "slice_comparison")
    for name in ("slice_fully_sampled.png", "slice_difference.png", "slice_comparison.png"):
        p = tmp_path / "v" / name
        assert p.exists() and p.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n" and p.stat().st_size > 2000, name


def test_every_environment_knob_is_in_the_design_table_and_there_are_at_most_twelve():
    """DESIGN.md section 9 is THE list of environment knobs (round 6 cut 30 to 12).  Every MSIREN_* variable the library (getenv in
    mri_inr_amd/csrc), the Python package and bench.py read must be a row of that table, and nothing else may be."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    for dp, _, files in os.walk(os.path.join(root, "mri_inr_amd")):
        if os.sep + "build" in dp or "__pycache__" in dp:
            continue
        for f in files:
            if f.endswith((".hip", ".h", ".py")):
                src = open(os.path.join(dp, f), errors="replace").read()
                read |= set(re.findall(r'getenv\("(MSIREN_[A-Z0-9_]+)"\)', src))
                read |= set(re.findall(r'env(?:iron)?(?:\.get)?[\(\[]\s*"(MSIREN_[A-Z0-9_]+)"', src))
    src = open(os.path.join(root, "bench.py")).read()
    read |= set(re.findall(r'environ(?:\.get)?[\(\[]\s*"(MSIREN_[A-Z0-9_]+)"', src))
    design = open(os.path.join(root, "DESIGN.md")).read()
    sec = design[design.index("## 9. Environment knobs"):design.index("## 10.")]
    table = set(re.findall(r"^\| `(MSIREN_[A-Z0-9_]+)", sec, re.M))
    assert read == table, (sorted(read - table), sorted(table - read))
    assert len(table) <= 12, sorted(table)
