"""Helpers to read tests/golden/*.npz (see tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import torch

from deer_vla_amd.config import DeerConfig
from deer_vla_amd import synthetic as syn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name))
    cfg = DeerConfig(**json.loads(bytes(z["cfg_json"]).decode()))
    seed = int(z["seed"])
    arrs = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files if k not in ("cfg_json", "seed")}
    return cfg, seed, arrs


def state(cfg, seed):
    return syn.make_synthetic_state(cfg, seed)


_FULL_STATES = {}


def full_size_state(cfg, seed, **kw):
    """make_synthetic_state for the full-size configurations, kept for the session (20 s and 6 GB of host memory per call: a dozen GPU tests
    build the same MPT-1B / ViT-L state).  READ-ONLY by contract: tests that edit a state dict build their own.  At most two states are held."""
    key = (json.dumps(cfg.__dict__, sort_keys=True, default=str), int(seed), tuple(sorted(kw.items())))
    if key not in _FULL_STATES:
        if len(_FULL_STATES) >= 2:
            _FULL_STATES.pop(next(iter(_FULL_STATES)))
        _FULL_STATES[key] = syn.make_synthetic_state(cfg, seed, **kw)
    return _FULL_STATES[key]


def s2str(t):
    return bytes(t.numpy().astype("uint8")).decode()
