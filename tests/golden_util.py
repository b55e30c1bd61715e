"""Loader for tests/golden/*.json (see tools/gen_golden.py)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load(curve_name):
    with open(os.path.join(HERE, "golden", f"{curve_name}.json")) as f:
        return json.load(f)


def scalars(hexes):
    return (np.frombuffer(b"".join(bytes.fromhex(h) for h in hexes), np.uint64).reshape(-1, 4).copy()
            if hexes else np.zeros((0, 4), np.uint64))


def points(hexes):
    return (np.frombuffer(b"".join(bytes.fromhex(h) for h in hexes), np.uint64).reshape(-1, 12).copy()
            if hexes else np.zeros((0, 12), np.uint64))


def xy_inf_matches(xy_inf, point_hex):
    want = points([point_hex])[0]
    if int(xy_inf[8]):
        return not want[8:].any()          # identity is (0, 1, 0): Z limbs all zero
    return bool((xy_inf[:8] == want[:8]).all()) and bool(want[8:].any())
