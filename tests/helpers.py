"""Shared helpers for the test-suite: exported-weight loading for the ORACLE side."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WDIR = os.path.join(ROOT, "neural-tape-modeling_amd", "weights")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def state_dict_np(name):
    """Exported checkpoint -> {key: np.ndarray} in the reference's state_dict layout."""
    with open(os.path.join(WDIR, "manifest.json")) as f:
        man = json.load(f)[str(name)]
    blob = np.fromfile(os.path.join(WDIR, man["file"]), dtype="<f4")
    return {t["key"]: blob[t["offset"]:t["offset"] + t["count"]].reshape(t["shape"]).copy()
            for t in man["tensors"]}


def oracle_weights(name):
    import oracle
    return oracle.Weights.from_state_dict(state_dict_np(name))


def load(npz):
    return np.load(os.path.join(GOLDEN, npz), allow_pickle=False)
