"""Multi-robot range-aided SLAM: ownership of the merged problem's variables and each robot's share of the problem
(ref getRobotMeasurements, src/DCORA_utils.cpp:1370-1512; landmark symbols src/Graph.cpp:584-616; unit spheres owned
by the source robot of their range, src/Graph.cpp:1092-1097; Q / G of one agent, src/Graph.cpp:824-1772).

CPU: the product's ownership tables and block extraction against an independent parse of the pyfg symbols and scipy
slicing of the oracle's global Q; the local problems add up to the global cost.
GPU: tests/testAgent.cpp:290-456 (testAgentMultiAgentRA) -- with every agent initialised at the ground truth, one
accelerated RBCD round (non-selected Nesterov updates, public-state pull, the selected agent's solve) leaves every
agent at the ground truth; and from a perturbed start RBCD rounds decrease the global cost."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import common
from test_raslam import RA, parse_pyfg_np, ra_path, ra_plain


def _expected_ownership(g):
    """independent restatement of the rules on the symbols of the file"""
    d, poses, lms, pp, rg = g
    pnames = sorted(poses, key=lambda s: (s[0], int(s[1:])))
    lnames = sorted(lms, key=lambda s: (s[1], int(s[2:])) if s[1].isupper() else ("M", int(s[1:])))
    pose_robot = [ord(s[0]) - 65 for s in pnames]
    landmark_robot = [ord(s[1]) - 65 if s[1].isupper() else 12 for s in lnames]
    count = {}
    for (a, b, rho, w) in rg:
        src = ord(a[1]) - 65 if a[0] == "L" and a[1].isupper() else (12 if a[0] == "L" else ord(a[0]) - 65)
        count[src] = count.get(src, 0) + 1
    sphere_robot = []
    for rb in sorted(count):
        sphere_robot += [rb] * count[rb]
    return pose_robot, sphere_robot, landmark_robot


@pytest.mark.parametrize("name", RA)
def test_ownership_and_agent_blocks(built, name):
    import dcora_amd as da
    from oracle import orc
    g = parse_pyfg_np(ra_path(name))
    ra = da.RADataset(ra_path(name))
    ro = orc.RADataset(ra_plain(name))
    d, n, l, b, k = ra.d, ra.n, ra.l, ra.b, ra.k
    pr, sr, lr = _expected_ownership(g)
    assert ra.pose_robot.tolist() == pr and ra.sphere_robot.tolist() == sr and ra.landmark_robot.tolist() == lr
    assert ra.robots == [0, 1]  # robots A and B, three poses each, one landmark each (LA0, LB0)
    Qg = ro.Q.to_scipy().tocsr()
    seen = np.zeros(k, int)
    rng = np.random.default_rng(0)
    r = d + 1
    X = orc.project_to_manifold(r, d, n, rng.standard_normal((r, k)), l=l, b=b)
    f_global = orc.Problem(r, d, n, ro.Q, reg=-1, l=l, b=b).f(X)
    f_sum, cross = 0.0, 0.0
    for rb in ra.robots:
        (na, la, ba), own, Qaa, Cc = ra.agent_blocks(rb)
        # the agent's own RA ordering: rotations | unit spheres | translations | landmarks
        want = [d * i + c for i in range(n) if pr[i] == rb for c in range(d)] + \
               [d * n + s for s in range(l) if sr[s] == rb] + \
               [d * n + l + i for i in range(n) if pr[i] == rb] + \
               [d * n + l + n + j for j in range(b) if lr[j] == rb]
        assert own.tolist() == want and (na, la, ba) == (pr.count(rb), sr.count(rb), lr.count(rb))
        seen[own] += 1
        rest = np.setdiff1d(np.arange(k), own)
        assert abs(Qaa.to_scipy() - Qg[own][:, own]).max() < 1e-12
        Cd = Cc.toarray()
        assert np.abs(Cd[:, own]).max() == 0 and np.abs(Cd[:, rest] - Qg[own][:, rest].toarray()).max() < 1e-12
        # local problem of the agent at the global X: 1/2 <Q_aa, Xa^T Xa> + <Xa, X C^T>
        Xa, G = X[:, own], X @ Cd.T
        Pa = orc.Problem(r, d, na, orc.CSR.from_scipy(sp.csr_matrix(Qaa.to_scipy())), G=G, reg=-1, l=la, b=ba)
        quad = 0.5 * np.sum((Xa @ Qaa.to_scipy().toarray()) * Xa)
        assert np.isclose(Pa.f(Xa), quad + np.sum(Xa * G), rtol=1e-12)
        f_sum += quad
        cross += np.sum(Xa * G)
    assert np.all(seen == 1)  # every variable has exactly one owner
    # sum of the diagonal parts + half of the cross terms = the global cost
    assert np.isclose(f_sum + 0.5 * cross, f_global, rtol=1e-11)


class _Agents:
    """RBCD++ round of the reference driver (examples/MultiRobotExample_RASLAM.cpp, Agent::iterate,
    src/Agent.cpp:535-596, 1158-1278) over per-agent device problems"""

    def __init__(self, ra, r):
        import dcora_amd as da
        self.da, self.ra, self.r = da, ra, r
        self.reg = da.precond_regularization(ra.Q)
        self.blocks = {rb: ra.agent_blocks(rb) for rb in ra.robots}
        self.P = {}
        for rb, ((na, la, ba), own, Qaa, Cc) in self.blocks.items():
            self.P[rb] = da.QuadraticProblem(r, ra.d, na, Qaa, G=np.zeros((r, own.size)), reg=self.reg, l=la, b=ba)
        self.R = len(ra.robots)
        self.gamma = self.alpha = 0.0

    def project(self, rb, M):
        (na, la, ba), own, _, _ = self.blocks[rb]
        return self.da.manifold_project(self.r, self.ra.d, na, M, l=la, b=ba)

    def round(self, X, V, selected):
        R = self.R
        self.gamma = (1 + np.sqrt(1 + 4.0 * R * R * self.gamma ** 2)) / (2.0 * R)
        self.alpha = 1.0 / (self.gamma * R)
        X, V = X.copy(), V.copy()
        for rb in self.ra.robots:           # non-selected agents: X <- Y = proj((1 - alpha) X + alpha V), V <- proj(V)
            if rb == selected:
                continue
            own = self.blocks[rb][1]
            X[:, own] = self.project(rb, (1 - self.alpha) * X[:, own] + self.alpha * V[:, own])
            V[:, own] = self.project(rb, V[:, own])
        (na, la, ba), own, Qaa, Cc = self.blocks[selected]
        Y = self.project(selected, (1 - self.alpha) * X[:, own] + self.alpha * V[:, own])
        P = self.P[selected]
        P.set_linear_term(X @ Cc.toarray().T)  # public states of the neighbours enter through G
        Xn = self.da.QuadraticOptimizer(P, self.da.ROptParameters()).optimize(Y)
        V[:, own] = self.project(selected, V[:, own] + self.gamma * (Xn - Y))
        X[:, own] = Xn
        return X, V


@pytest.mark.gpu
@pytest.mark.parametrize("name", RA)
def test_multi_agent_ra_round_keeps_ground_truth(built, name):
    import dcora_amd as da
    ra = da.RADataset(ra_path(name))
    d = ra.d
    ag = _Agents(ra, d)
    X, V = ra.gt.copy(), ra.gt.copy()
    for sel in ra.robots:  # each robot is the selected one once (tests/testAgent.cpp:402-452)
        X, V = ag.round(X, V, sel)
        own = ag.blocks[sel][1]
        assert np.abs(X[:, own] - ra.gt[:, own]).max() < 1e-9  # OPTIMIZATION_TOL, ref tests/testAgent.cpp:20
    assert np.abs(X - ra.gt).max() < 1e-9


@pytest.mark.gpu
def test_multi_agent_ra_rounds_decrease_the_global_cost(built):
    import dcora_amd as da
    from oracle import orc
    name = "range_aided_slam_test_3d"
    ra = da.RADataset(ra_path(name))
    d, n, l, b, r = ra.d, ra.n, ra.l, ra.b, ra.d + 1
    rng = np.random.default_rng(4)
    lift = np.linalg.qr(rng.standard_normal((r, d)))[0]
    X = da.manifold_project(r, d, n, lift @ ra.gt + 0.05 * rng.standard_normal((r, ra.k)), l=l, b=b)
    Pg = da.QuadraticProblem(r, d, n, ra.Q, reg=-1.0, l=l, b=b)
    ag = _Agents(ra, r)
    V = X.copy()
    costs = [Pg.f(X)]
    for it in range(12):
        X, V = ag.round(X, V, ra.robots[it % 2])
        costs.append(Pg.f(X))
    # accelerated block-coordinate descent is not monotone step by step; it must get (far) down and stay down
    assert costs[-1] < 1e-3 * costs[0] and min(costs) == min(costs[6:])
