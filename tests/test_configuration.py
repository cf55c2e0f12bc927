"""YAML surface: every YAML the reference ships, fed to this build's loader, yields the tree the
reference's own loader produces.  Fixture tests/golden/configs.json (oracle/gen_fixtures.py) holds,
per reference file, the parsed YAML input ("user") and the reference loader's output ("config")."""
import json
import os

import pytest
import yaml

from conftest import GOLDEN
from mri_inr_amd import configuration as cfgmod
from mri_inr_amd import load_configuration, model_kwargs

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(GOLDEN, "configs.json")))


def _write(tmp_path, name, data):
    p = tmp_path / (name.replace("/", "__"))
    p.write_text(yaml.safe_dump(data, sort_keys=False))
    return str(p)


@pytest.mark.parametrize("rel", sorted(GOLD))
def test_loader_matches_reference(rel, tmp_path):
    want = GOLD[rel]
    ns = load_configuration(_write(tmp_path, rel, want["user"]), testing=want["testing"])
    assert cfgmod.namespace_to_dict(ns) == want["config"]


def test_all_19_reference_yaml_files_covered():
    assert len(GOLD) == 19 and all("error" not in v for v in GOLD.values())


def test_no_cross_call_contamination(tmp_path):
    a = load_configuration(_write(tmp_path, "a", GOLD["ablations/test_morlet.yaml"]["user"]), testing=True)
    b = load_configuration(_write(tmp_path, "b", GOLD["test_modulated_siren.yaml"]["user"]), testing=True)
    assert a.model.activation == "morlet" and b.model.activation == "sine"
    assert cfgmod.default_test_config["model"].get("activation") is None


def test_shipped_eval_configs_load():
    for name, act in (("eval_sine.yaml", "sine"), ("eval_morlet.yaml", "morlet")):
        c = load_configuration(os.path.join(REPO, "configuration", name), testing=True)
        assert c.model.activation == act and c.model.dim_hidden == 256 and c.testing.model_path == "synthetic"


def test_model_kwargs_and_quirks(tmp_path):
    c = load_configuration(_write(tmp_path, "t", GOLD["test_modulated_siren.yaml"]["user"]), testing=True)
    kw = model_kwargs(c, device="cuda")
    assert set(kw) == {"dim_in", "dim_hidden", "dim_out", "num_layers", "latent_dim", "w0", "w0_initial", "use_bias",
                       "dropout", "modulate", "encoder_type", "encoder_path", "outer_patch_size", "inner_patch_size",
                       "siren_patch_size", "device", "activation"}
    assert kw["dim_hidden"] == 256 and kw["siren_patch_size"] == 24 and kw["modulate"] is True
    # quirk (b): the test defaults have no "activation"; a YAML without it fails like the reference
    c2 = load_configuration(_write(tmp_path, "u", {"model": {"dim_hidden": 64}}), testing=True)
    assert c2.model.dim_hidden == 64 and c2.model.encoder_type == "default"
    with pytest.raises(AttributeError):
        model_kwargs(c2)
    # train defaults do carry it
    assert load_configuration(_write(tmp_path, "u", {"model": {"dim_hidden": 64}}), testing=False).model.activation == "sine"
