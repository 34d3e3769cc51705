"""Tiny container format for golden cases: a list of dicts (arrays + scalars) in one .npz."""
from __future__ import annotations

import json
import os

import numpy as np

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))


def save_cases(name: str, cases: list[dict]) -> str:
    arrays = {}
    meta = []
    for i, case in enumerate(cases):
        m = {}
        for k, v in case.items():
            if isinstance(v, np.ndarray):
                arrays[f"{i}/{k}"] = v
            elif v is None or isinstance(v, (bool, int, float, str, list, dict)):
                m[k] = v
            elif isinstance(v, (np.integer, np.floating, np.bool_)):
                m[k] = v.item()
            else:
                raise TypeError(f"case {i} field {k}: {type(v)}")
        meta.append(m)
    arrays["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **arrays)
    return path


def load_cases(name: str) -> list[dict]:
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    with np.load(path) as z:
        meta = json.loads(bytes(z["__meta__"]).decode())
        cases = [dict(m) for m in meta]
        for key in z.files:
            if key == "__meta__":
                continue
            i, k = key.split("/", 1)
            cases[int(i)][k] = z[key]
    return cases
