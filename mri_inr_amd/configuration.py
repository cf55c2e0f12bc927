"""YAML configuration surface of the reference (src/configuration/configuration.py).

``load_configuration(path, testing)`` merges the user's YAML over the reference's defaults
(:11-93) and returns the same nested ``SimpleNamespace`` tree (``config.model.*``,
``config.data.*``, ``config.testing.*`` / ``config.training.*``), so every shipped YAML file
(configuration/**/*.yaml) loads unchanged.  Checked key by key against the reference's own
loader in tests/test_configuration.py (fixture tests/golden/configs.json).

One deliberate difference: the reference merges *into its module-level default dicts*
(:107-112, :178-181), so a second call in the same process sees the first file's values.
Here the defaults are deep-copied per call.  A single call per process -- the only way the
reference's scripts use it -- behaves identically.
"""

from __future__ import annotations

import argparse
import copy
import types

import yaml

_MODEL_DEFAULTS = {
    "dim_in": 2,
    "dim_hidden": 256,
    "dim_out": 1,
    "latent_dim": 256,
    "num_layers": 5,
    "w0": 1.0,
    "w0_initial": 30.0,
    "use_bias": True,
    "dropout": 0.1,
    "encoder_type": "default",
    "encoder_path": "./model/custom_encoder.pth",
    "outer_patch_size": 32,
    "inner_patch_size": 16,
    "siren_patch_size": 24,
}

# reference :11-61 -- note "activation" exists only in the training defaults (:43 vs :72-87)
default_train_config = {
    "data": {
        "train": {"dataset": "", "num_samples": None, "mri_type": "FLAIR", "num_workers": 4},
        "val": {"dataset": None, "num_samples": 10, "mri_type": "FLAIR", "num_workers": 4},
        "acceleration": 6,
        "center_fraction": 0.05,
    },
    "model": dict(_MODEL_DEFAULTS, activation="sine"),
    "training": {
        "lr": 0.0001,
        "batch_size": 10,
        "epochs": 100,
        "output_dir": "./output",
        "output_name": "modulated_siren",
        "optimizer": "Adam",
        "logging": False,
        "criterion": "MSE",
        "model": {"continue_training": False, "model_path": None, "optimizer_path": None},
    },
}

# reference :64-93
default_test_config = {
    "data": {
        "dataset": "",
        "test_files": None,
        "metric_samples": None,
        "visual_samples": 0,
        "acceleration": 6,
        "center_fraction": 0.05,
    },
    "model": dict(_MODEL_DEFAULTS),
    "testing": {"output_dir": "./output", "output_name": "modulated_siren", "model_path": ""},
}


def merge_configs(defaults: dict, user_configs: dict) -> dict:
    """Recursive overlay: dict values merge into existing keys, everything else replaces (:96-112)."""
    for key, value in (user_configs or {}).items():
        if isinstance(value, dict) and key in defaults:
            merge_configs(defaults[key], value)
        else:
            defaults[key] = value
    return defaults


def convert_to_namespace(data):
    """dict tree -> SimpleNamespace tree; lists and scalars are kept (:115-129)."""
    if isinstance(data, dict):
        return types.SimpleNamespace(**{k: convert_to_namespace(v) for k, v in data.items()})
    return data


def namespace_to_dict(obj):
    """Inverse of :func:`convert_to_namespace` (:132-149)."""
    if isinstance(obj, types.SimpleNamespace):
        obj = vars(obj)
    if isinstance(obj, dict):
        return {k: namespace_to_dict(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [namespace_to_dict(v) for v in obj]
    return obj


def save_config_to_yaml(config, filename):
    with open(filename, "w") as fh:
        yaml.dump(namespace_to_dict(config), fh, default_flow_style=False, sort_keys=False)


def load_configuration(file_path, testing: bool = False):
    """YAML file -> namespace, merged over the test or train defaults (:164-185)."""
    with open(file_path, "r") as fh:
        user = yaml.safe_load(fh)
    base = copy.deepcopy(default_test_config if testing else default_train_config)
    return convert_to_namespace(merge_configs(base, user))


def load_configuration_no_defaults(file_path):
    with open(file_path, "r") as fh:
        return convert_to_namespace(yaml.safe_load(fh))


def parse_args(argv=None):
    """The single CLI flag of the reference's scripts: ``--config`` (:206-212)."""
    parser = argparse.ArgumentParser(description="Evaluate a modulated SIREN on MRI data (MI355X path).")
    parser.add_argument("--config", type=str, required=True, help="Path to the configuration file")
    return parser.parse_args(argv)


def model_kwargs(config, device="cuda", modulate=True) -> dict:
    """The 17 keyword arguments test_mod_siren.py:96-114 passes to ModulatedSiren."""
    m = config.model
    return dict(
        dim_in=m.dim_in, dim_hidden=m.dim_hidden, dim_out=m.dim_out, num_layers=m.num_layers,
        latent_dim=m.latent_dim, w0=m.w0, w0_initial=m.w0_initial, use_bias=m.use_bias, dropout=m.dropout,
        modulate=modulate, encoder_type=m.encoder_type, encoder_path=m.encoder_path,
        outer_patch_size=m.outer_patch_size, inner_patch_size=m.inner_patch_size,
        siren_patch_size=m.siren_patch_size, device=device,
        activation=m.activation,  # AttributeError if a test YAML omits it, as in the reference (quirk b)
    )
