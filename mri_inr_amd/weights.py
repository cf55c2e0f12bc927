"""Checkpoint readers.  The reference saves raw ``state_dict()`` objects with ``torch.save``
(src/train/training.py:398-416) and the encoder as ``{"state_dict": ...}``
(src/networks/encoding/siren_encoder.py:547); ``.npz`` is accepted as a torch-free alternative."""

from __future__ import annotations

import numpy as np


def load_checkpoint(path: str) -> dict:
    """Return ``{key: np.ndarray}`` (one level of ``{"state_dict": {...}}`` nesting is kept)."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {k: np.asarray(z[k]) for k in z.files}
    import torch  # deserialisation only

    obj = torch.load(path, map_location="cpu", weights_only=True)

    def conv(o):
        if isinstance(o, dict):
            return {k: conv(v) for k, v in o.items()}
        if hasattr(o, "detach"):
            return o.detach().cpu().numpy()
        return o

    return conv(obj)


def save_checkpoint(path: str, state_dict: dict) -> None:
    """Write a state_dict as ``.npz`` or as a torch ``.pth`` the reference can load."""
    if path.endswith(".npz"):
        np.savez(path, **{k: np.asarray(v) for k, v in state_dict.items()})
        return
    import torch

    torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state_dict.items()}, path)
