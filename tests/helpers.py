"""Shared helpers for the tests (CPU side): golden loading, oracle nets with synthetic weights."""
import json
import os

import numpy as np
import torch

from oracle import hat_oracle as O
from super_resolution_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "meta.json")) as f:
    META = json.load(f)
W_SEED, X_SEED = META["w_seed"], META["x_seed"]


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def cfg_of(name):
    return O.make_cfg(**META["cfgs"][name])


def oracle_sd(name):
    cfg = cfg_of(name)
    return cfg, synth.synth_state_dict(O.blank_state_dict(cfg), W_SEED)


def max_abs(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())
