"""Exported reference checkpoints (raw little-endian fp32 + manifest.json, made by
tools/make_goldens.py from weights/<name>/best.pth) -> state_dict of torch tensors with the
reference's key names."""
import json
import os

import numpy as np
import torch

_WDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights")

W_GRU = "GRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST"
W_GRU_ESR = "GRU-HS[64]-L[ESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE]_BEST"
W_DIFFDEL = "DiffDelGRU-HS[64]-L[DCPreESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"
W_DIFFDEL_ESR = "DiffDelGRU-HS[64]-L[ESR]-DS[ReelToReel_Dataset_MiniPulse100_CHOWTAPE_WOWFLUTTER]_BEST"


def available():
    with open(os.path.join(_WDIR, "manifest.json")) as f:
        return sorted(json.load(f))


def load_state_dict(name, map_location="cpu"):
    """Stand-in for torch.load('<weights>/<name>/best.pth') (code/test-model.py:233)."""
    with open(os.path.join(_WDIR, "manifest.json")) as f:
        man = json.load(f)[name]
    blob = np.fromfile(os.path.join(_WDIR, man["file"]), dtype="<f4")
    sd = {}
    for t in man["tensors"]:
        a = blob[t["offset"]:t["offset"] + t["count"]].reshape(t["shape"]).copy()
        sd[t["key"]] = torch.from_numpy(a).to(map_location)
    return sd
