"""shared test helpers: dataset fixtures for the product (dcora_amd) and for the oracle (oracle.orc)"""
import os

import numpy as np

from dcora_amd.datasets import DATA, data_path, plain_path, product_dataset  # noqa: F401
from oracle.flows import oracle_dataset  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))


def random_point(r, d, n, seed, project):
    rng = np.random.default_rng(seed)
    return project(r, d, n, rng.uniform(-1, 1, (r, (d + 1) * n)))


def random_tangent(r, d, n, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((r, (d + 1) * n))


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(1e-300, np.linalg.norm(np.asarray(b)))
