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


def s2str(t):
    return bytes(t.numpy().astype("uint8")).decode()
