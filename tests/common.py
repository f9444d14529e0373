"""shared test helpers: dataset fixtures for the product (dcora_amd) and for the oracle (oracle.orc)"""
import gzip
import os
import shutil
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")


def data_path(name):
    return os.path.join(DATA, name + ".g2o.gz")


_tmp = {}


def plain_path(name):
    """decompressed copy (the oracle's reader takes plain files)"""
    if name not in _tmp:
        fd, p = tempfile.mkstemp(suffix="_%s.g2o" % name)
        with os.fdopen(fd, "wb") as out, gzip.open(data_path(name), "rb") as src:
            shutil.copyfileobj(src, out)
        _tmp[name] = p
    return _tmp[name]


def oracle_dataset(name):
    from oracle import orc
    return orc.read_g2o(plain_path(name))


def product_dataset(name):
    import dcora_amd as da
    return da.Dataset.load_g2o(data_path(name))


def random_point(r, d, n, seed, project):
    rng = np.random.default_rng(seed)
    return project(r, d, n, rng.uniform(-1, 1, (r, (d + 1) * n)))


def random_tangent(r, d, n, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((r, (d + 1) * n))


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(1e-300, np.linalg.norm(np.asarray(b)))
