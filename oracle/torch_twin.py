"""CPU ORACLE (timing twin) -- test/bench infrastructure only, never the product path.

The same maths as oracle/siren_oracle.py written with torch CPU ops (ATen/MKL, all host cores),
i.e. the kind of arithmetic the reference's own CPU path performs (stock ``F.linear``, ``torch.sin``,
in-place modulation; src/networks/modulated_siren.py:144-157, 215-233, 325-343;
src/networks/encoding/siren_encoder.py:503-512).  It exists so that ``bench.py`` can report a CPU
baseline on the GPU box's host cores -- the reference itself cannot travel there.  It is written
from the maths of SURVEY.md Appendix A, is checked against the numpy oracle and the golden
fixtures in tests/test_oracle_golden.py, and is ``"kind": "port"`` in the bench JSON.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def to_tensors(sd: dict) -> dict:
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd.items()}


@torch.no_grad()
def trunk(t: dict, mods: torch.Tensor, *, num_layers: int, w0=1.0, w0_initial=30.0, activation="sine"):
    """mods (L,B,H) -> (B,P)."""
    B = mods.shape[1]
    x = t["grid"].unsqueeze(0).expand(B, -1, -1)
    for l in range(num_layers):
        w = w0_initial if l == 0 else w0
        p = F.linear(x, t[f"net.layers.{l}.weight"], t.get(f"net.layers.{l}.bias"))
        a = torch.sin(w * p)
        if activation == "morlet":
            a = a * torch.exp(-0.5 * p * p)
        a *= mods[l].unsqueeze(1)
        x = a
    p = F.linear(x, t["net.last_layer.weight"], t.get("net.last_layer.bias"))
    return torch.sin(w0 * p).squeeze(2)


@torch.no_grad()
def modulator(t: dict, z: torch.Tensor, *, num_layers: int):
    x, outs = z, []
    for l in range(num_layers):
        h = torch.relu(F.linear(x, t[f"modulator.layers.{l}.0.weight"], t[f"modulator.layers.{l}.0.bias"]))
        outs.append(h)
        x = torch.cat((h, z), dim=1)
    return torch.stack(outs, 0)


@torch.no_grad()
def encoder(t: dict, tiles: torch.Tensor):
    p = "encoder.encoder.encoder."
    x = tiles.unsqueeze(1)
    x = F.leaky_relu(F.conv2d(x, t[p + "0.weight"], t[p + "0.bias"], stride=2, padding=1), 0.2)
    x = F.leaky_relu(F.conv2d(x, t[p + "2.weight"], t[p + "2.bias"], stride=2, padding=1), 0.2)
    x = F.leaky_relu(F.conv2d(x, t[p + "4.weight"], t[p + "4.bias"]), 0.2)
    return F.linear(x.flatten(1), t[p + "7.weight"], t[p + "7.bias"])


@torch.no_grad()
def forward_tiles(t: dict, tiles: torch.Tensor, *, num_layers: int, w0=1.0, w0_initial=30.0,
                  activation="sine", siren_patch_size=24):
    """ModulatedSiren.forward: tiles (B,O,O) -> (B,S,S)."""
    mods = modulator(t, encoder(t, tiles), num_layers=num_layers)
    out = trunk(t, mods, num_layers=num_layers, w0=w0, w0_initial=w0_initial, activation=activation)
    return out.reshape(out.shape[0], siren_patch_size, siren_patch_size)
