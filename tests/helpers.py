"""shared helpers for the parity tests"""
import json
import os
import numpy as np
from ethsnarks_amd import r1cs as R, fields as F

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    with open(os.path.join(GOLDEN, "proofs_pyref.json")) as f:
        return json.load(f)


def build_case(case):
    if case["kind"] == "chain":
        r, w = R.synthetic_chain(case["nC"], case["nIn"], case["seed"])
    else:
        r, w = R.random_r1cs(case["nC"], case["nIn"], seed=case["seed"], small_values=case["small_values"])
    toxic = [int(t, 16) for t in case["toxic"]]
    return r, w, toxic


def rand_scalars(n, seed, ones_every=0, zeros_every=0):
    rng = R.SplitMix64(seed)
    sc = [rng.fr() for _ in range(n)]
    if ones_every:
        for i in range(0, n, ones_every):
            sc[i] = 1
    if zeros_every:
        for i in range(1, n, zeros_every):
            sc[i] = 0
    return sc


def tiled_bases(oracle, n, g2=False, distinct=512, seed=11):
    """n valid curve points: `distinct` oracle-generated multiples of the generator, tiled."""
    rng = R.SplitMix64(seed)
    d = min(n, distinct)
    if d == 0:
        return np.zeros((0, 16 if g2 else 8), dtype=np.uint64)
    pts = oracle.batch_mul(F.fr_to_mont([rng.fr() for _ in range(d)]), g2=g2)
    reps = (n + d - 1) // d
    return np.tile(pts, (reps, 1))[:n].copy()
