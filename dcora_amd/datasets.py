"""Where the public datasets shipped with the repository live (tests/golden/data: g2o / pyfg files of the reference's
data directory, gzip-compressed) and how to load them into the product's handles.  Used by bench.py and the tests."""
import gzip
import os
import shutil
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")


def data_path(name):
    return os.path.join(DATA, name + ".g2o.gz")


_tmp = {}


def plain_path(name, ext="g2o"):
    """decompressed copy (readers that take plain files: the C ABI's g2o / pyfg loaders, the oracle)"""
    key = (name, ext)
    if key not in _tmp:
        fd, p = tempfile.mkstemp(suffix="_%s.%s" % (name, ext))
        with os.fdopen(fd, "wb") as out, gzip.open(os.path.join(DATA, "%s.%s.gz" % (name, ext)), "rb") as src:
            shutil.copyfileobj(src, out)
        _tmp[key] = p
    return _tmp[key]


def product_dataset(name):
    from . import Dataset
    return Dataset.load_g2o(data_path(name))
