"""Control flows over the CPU oracle (TEST INFRASTRUCTURE, like everything under oracle/): the backend that lets the
centralised CORA driver of dcora_amd/cora_flow.py run on the oracle, the multi-robot RA-SLAM loop over the oracle's
local solver, and the oracle's dataset loader.  Imported by tests/ and by bench.py's cpu legs only."""
import numpy as np
import scipy.sparse as sp

from dcora_amd.cora_flow import MIN_EIG_TOL, PARAMS
from dcora_amd.datasets import plain_path


def oracle_dataset(name):
    from oracle import orc
    return orc.read_g2o(plain_path(name))


class OracleBackend:
    name = "cpu"

    def __init__(self, ra_oracle, reg):
        from oracle import orc
        self.orc, self.ra, self.Q, self.reg = orc, ra_oracle, ra_oracle.Q, reg

    def problem(self, r):
        ra = self.ra
        return self.orc.Problem(r, ra.d, ra.n, self.Q, reg=self.reg, l=ra.l, b=ra.b)

    def optimize(self, P, X):
        Xo, res = P.optimize(X, **PARAMS)
        return Xo, res["fOpt"], res["gradNormOpt"], int(res["outer_iters"]), int(res["inner_iters"])

    def certificate(self, r, X):
        ra = self.ra
        S = self.orc.dual_certificate(r, ra.d, ra.n, X, self.Q, l=ra.l, b=ra.b)
        psd, theta, v, lmin = self.orc.fast_verification(S, MIN_EIG_TOL, block=1)
        return psd, theta, v

    def escape(self, Pnext, X, theta, v):
        return Pnext.escape_saddle(X, theta, v, 1e-4, 1e-4, second_order=True)

    def project(self, X, r):
        ra = self.ra
        return self.orc.project_solution_raslam(X, r, ra.d, ra.n, ra.l, ra.b)

    def close(self, P):
        pass


def oracle_ra_rbcd_loop(da, orc, ra, X0, r, iters, accel, restart_interval, opt):
    """Agent::iterate for every agent + central evaluation + greedy selection (ref src/Agent.cpp:535-596, 1158-1278,
    examples/MultiRobotExample_RASLAM.cpp), every numerical step on the oracle"""
    d, n, l, b, k = ra.d, ra.n, ra.l, ra.b, ra.k
    robots = ra.robots
    R = len(robots)
    Qo = orc.CSR.from_scipy(ra.Q.to_scipy())
    central = orc.Problem(r, d, n, Qo, reg=-1, l=l, b=b)
    blk, P, C = {}, {}, {}
    for rb in robots:
        dims3, own, Qaa, Cc = ra.agent_blocks(rb)
        reg = da.precond_regularization(Qaa)  # the session computes the same per-agent regularisation
        blk[rb] = (dims3, own)
        C[rb] = Cc.tocsr()
        P[rb] = lambda G, rb=rb, Qaa=Qaa, dims3=dims3, reg=reg: orc.Problem(
            r, d, dims3[0], orc.CSR.from_scipy(sp.csr_matrix(Qaa.to_scipy())), G=G, reg=reg, l=dims3[1], b=dims3[2])
    proj = lambda rb, M: orc.project_to_manifold(r, d, blk[rb][0][0], M, l=blk[rb][0][1], b=blk[rb][0][2])
    X = X0.copy()
    Xa = {rb: X[:, blk[rb][1]].copy() for rb in robots}
    V = {rb: Xa[rb].copy() for rb in robots}
    Y = {rb: Xa[rb].copy() for rb in robots}
    gamma = alpha = 0.0
    sel, trace = 0, []
    for it in range(1, iters + 1):
        if accel:
            gamma = (1 + np.sqrt(1 + 4.0 * R * R * gamma * gamma)) / (2.0 * R)
            alpha = 1.0 / (gamma * R)
        restart = accel and ((it + 1) % restart_interval == 0)
        XPrev = {rb: Xa[rb].copy() for rb in robots}
        for i, rb in enumerate(robots):
            if i == sel or not accel:
                continue
            if restart:
                V[rb], Y[rb] = Xa[rb].copy(), Xa[rb].copy()
            else:
                Y[rb] = proj(rb, (1 - alpha) * Xa[rb] + alpha * V[rb])
                Xa[rb] = Y[rb].copy()
                V[rb] = proj(rb, V[rb])
                X[:, blk[rb][1]] = Xa[rb]
        rb = robots[sel]

        def solve(start):
            G = (C[rb] @ X.T).T
            return P[rb](G).optimize(start, **opt)[0]

        if accel:
            Y[rb] = proj(rb, (1 - alpha) * Xa[rb] + alpha * V[rb])
            Xn = solve(Y[rb])
            V[rb] = proj(rb, V[rb] + gamma * Xn - gamma * Y[rb])
            Xa[rb] = Xn
            if restart:
                Xa[rb] = solve(XPrev[rb])
                V[rb], Y[rb] = Xa[rb].copy(), Xa[rb].copy()
        else:
            Xa[rb] = solve(Xa[rb])
        X[:, blk[rb][1]] = Xa[rb]
        if restart:
            gamma = alpha = 0.0
        RG = central.rgrad(X)
        bn = np.array([np.linalg.norm(RG[:, blk[q][1]]) for q in robots])
        trace.append((sel, 2 * central.f(X), np.linalg.norm(RG)))
        nxt = int(np.argmax(bn))
        sel = nxt if C[rb].nnz > 0 else sel
    return X, np.array(trace)
