"""Forms of two hot kernels live side by side (the choice is read once per process): the block Q-apply without LDS
(k_spmm_bsr2, the default) against the LDS-staged one (DCORA_BSR_KERNEL=v1), which must agree BITWISE (same
summation order), the form with 16-byte gathers (k_spmm_bsr3, DCORA_BSR_KERNEL=v3) against both to rounding, and the
entry-per-lane level kernel of the sparse preconditioner (k_sp_level2) against the (entry, value)-per-lane one
(DCORA_SP_KERNEL=v1), which sum in a different order and must agree to rounding.  One child process per form."""
import os
import subprocess
import sys

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import common
import dcora_amd as da
from dcora_amd import synth
out = {}
cases = [("lattice", synth.lattice_se3(9, 8, 7, seed=5)), ("sphere", common.product_dataset("sphere2500")),
         ("grid2d", common.product_dataset("pose_graph_optimization_test_2d"))]
for name, ds in cases:
    Q = da.build_Q_pgo(ds)
    k = (ds.d + 1) * ds.n
    for r in sorted({ds.d, 5, 8}):
        rng = np.random.default_rng(3)
        X = rng.standard_normal((r, k))
        Gm = rng.standard_normal((r, k))
        P = da.QuadraticProblem(r, ds.d, ds.n, Q, G=Gm, reg=0.1)
        out["%s_%d_info" % (name, r)] = np.array([P.qapply_info()["kernel"] == "k_spmm", P.precond_info()["kind"] == "sparse"])
        out["%s_%d_f" % (name, r)] = np.array([P.f(X)])
        out["%s_%d_g" % (name, r)] = P.EucGrad(X)
        Xm = da.manifold_project(r, ds.d, ds.n, X)
        V = P.RieGrad(Xm)
        out["%s_%d_z" % (name, r)] = P.PreCondition(Xm, V)
        P.close()
np.savez(sys.argv[2], **out)
'''


def _run(tmp_path, tag, env):
    e = dict(os.environ)
    e.update({"DCORA_QAPPLY": "bsr", "DCORA_PRECOND": "sparse"})
    e.update(env)
    out = os.path.join(str(tmp_path), tag + ".npz")
    res = subprocess.run([sys.executable, "-c", CHILD, os.path.dirname(common.HERE), out], env=e, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    return np.load(out)


def test_both_forms_of_the_block_qapply_and_of_the_level_kernel_agree(built, tmp_path):
    if os.environ.get("DCORA_SOLVER_V1"):
        pytest.skip("DCORA_SOLVER_V1 switches the block Q-apply off")
    new = _run(tmp_path, "new", {})
    old = _run(tmp_path, "old", {"DCORA_BSR_KERNEL": "v1", "DCORA_SP_KERNEL": "v1"})
    v3 = _run(tmp_path, "v3", {"DCORA_BSR_KERNEL": "v3"})  # 16-byte gathers, column pairs summed per pose (another order)
    assert set(new.files) == set(old.files) == set(v3.files)
    for key in new.files:
        if key.endswith("_info"):
            assert not new[key][0] and new[key][1], key   # the block Q-apply and the sparse preconditioner ran
        elif key.endswith("_z"):
            assert common.rel(new[key], old[key]) < 1e-12, key
            assert common.rel(v3[key], old[key]) < 1e-12, key
        else:
            assert np.array_equal(new[key], old[key]), key
            assert common.rel(v3[key], old[key]) < 1e-13, key
