"""The host half of the set-up path under ASan + UBSan (CPU only; the GPU pool offers no sanitizer): ordering with the
merged dense top, symbolic analysis + host executor of the device Cholesky's schedule, partitioned-inverse builder
(threads, deferred weight fill, hub Schur complement) and its host replay, on a lattice block, a sphere2500 block and a
hub matrix.  Any sanitizer report fails the run; the residuals of the replay are checked as well."""
import os
import re
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

import common

ROOT = os.path.dirname(common.HERE)
SRC = [os.path.join(ROOT, "dcora_amd", "csrc", f) for f in ("host_cholsym.cpp", "host_sparse.cpp", "host_partinv.cpp", "host_partinv3.cpp", "env.cpp")]


def _dump(A, path):
    A = sp.csr_matrix(A)
    A.sort_indices()
    with open(path, "wb") as f:
        np.array([A.shape[0], A.nnz], np.int32).tofile(f)
        A.indptr.astype(np.int32).tofile(f)
        A.indices.astype(np.int32).tofile(f)
        A.data.astype(np.float64).tofile(f)


def test_host_setup_code_under_asan_ubsan(built, tmp_path):
    import dcora_amd as da
    from dcora_amd import synth
    exe = str(tmp_path / "san_host_setup")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "dcora_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
           os.path.join(common.HERE, "cpp", "san_host_setup.cpp")] + SRC + ["-lpthread", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    lat = synth.lattice_se3(8, 8, 6)
    Ql = da.build_Q_pgo(lat).to_scipy()
    _dump(Ql + 0.1 * sp.identity(Ql.shape[0]), tmp_path / "lat.bin")
    ds = common.product_dataset("sphere2500")
    Qs = da.build_Q_pgo(ds).to_scipy()[:3200, :3200]
    _dump(Qs + 0.1 * sp.identity(3200), tmp_path / "sphere.bin")
    R = sp.random(500, 500, 0.01, random_state=1)
    H = (R + R.T + 30 * sp.identity(500)).tolil()
    H[499, :] = 0.1   # a hub
    H[:, 499] = 0.1
    H[499, 499] = 100
    _dump(H, tmp_path / "hub.bin")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    out = subprocess.run([exe, str(tmp_path / "lat.bin"), "4", str(tmp_path / "sphere.bin"), "4",
                          str(tmp_path / "hub.bin"), "1"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-2000:]
    resid = [float(x) for x in re.findall(r"resid ([0-9.eE+-]+)", out.stdout)]
    assert len(resid) == 3 and max(resid) < 1e-8, out.stdout
    assert out.stdout.count("ok 1") == 12 and out.stdout.count("differ 0") == 3, out.stdout
