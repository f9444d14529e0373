"""The multi-rank product path on ONE GPU: several processes (one per rank, all on device 0) run the RBCD loop through
the library's neighbour exchange (dcora_exchange_*: IPC peer stores or the shared host segment, flag words, the
evaluation all-gather) and must reproduce the single-session run -- the iterates bit for bit, the costs to rounding
(the evaluation sums per agent instead of centrally).  ref examples/MultiRobotExample.cpp:223-307,
src/Agent.cpp:113-152, 844-906."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu

WORKER = os.path.join(common.HERE, "exchange_worker.py")


def run_ranks(tmp_path, world, name, R, r, iters, mode, X0, transport=None, wait=None, certify=None, extra_env=None):
    np.save(os.path.join(tmp_path, "X0.npy"), X0)
    job = "t%s" % uuid.uuid4().hex[:12]
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if transport:
        env["DCORA_EXCHANGE"] = transport
    else:
        env.pop("DCORA_EXCHANGE", None)
    if wait:
        env["DCORA_EXCHANGE_WAIT"] = wait
    else:
        env.pop("DCORA_EXCHANGE_WAIT", None)
    if certify is not None:
        env["DCORA_TEST_CERTIFY"] = repr(certify)
    else:
        env.pop("DCORA_TEST_CERTIFY", None)
    for k_, v_ in (extra_env or {}).items():
        env[k_] = v_
    procs = [subprocess.Popen([sys.executable, WORKER, str(k), str(world), job, name, str(R), str(r), str(iters),
                               str(tmp_path), mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for k in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for k, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (k, outs[k][-3000:])
    return [np.load(os.path.join(tmp_path, "rank%d.npz" % k)) for k in range(world)]


def single(da, ds, R, r, iters, mode, X0):
    s = da.RbcdSession(ds, num_robots=R, r=r, acceleration=(mode == "greedy"))
    s.set_X(X0)
    if mode == "greedy":
        out = s.run(max_iters=iters, rgrad_tol=0.0)
        cost, gn, sel = out["cost"], out["gradnorm"], out["selected"]
    else:
        col, nc = s.colours()
        cost, gn, sel = [], [], []
        for _ in range(iters):
            for c in range(nc):
                s.iterate_set(np.flatnonzero(col == c).astype(np.int32))
            c2, g, bn, nxt = s.evaluate()
            cost.append(c2)
            gn.append(g)
            sel.append(nxt)
    X = s.get_X()
    s.close()
    return np.asarray(cost), np.asarray(gn), np.asarray(sel), X


CASES = [
    # dataset, agents, ranks, iterations, mode, transport, who waits for the producer's flag: "device" = the scatter
    # kernel polls it (the default where every rank has a GPU of its own), "host" = the host spins (the default where
    # ranks share a GPU), None = that default
    ("sphere2500", 5, 2, 40, "greedy", None, "device"),        # restart round (30) inside
    ("sphere2500", 5, 4, 12, "greedy", "staged", "device"),    # rank 3 hosts no agent; shared-host-segment transport
    ("torus3D", 8, 4, 10, "greedy", None, "device"),           # BASELINE config 3's split, two agents per rank
    ("torus3D", 8, 2, 4, "coloured", None, "device"),          # simultaneous updates of one colour, then post + wait
    ("sphere2500", 5, 2, 12, "greedy", None, "host"),          # the round-2 form of the wait
    ("torus3D", 8, 4, 6, "greedy", "staged", "host"),
    ("torus3D", 8, 2, 6, "greedy", None, None),
]


@pytest.mark.parametrize("name,R,world,iters,mode,transport,wait", CASES)
def test_ranks_reproduce_single_session(tmp_path, name, R, world, iters, mode, transport, wait):
    import dcora_amd as da
    ds = common.product_dataset(name)
    r = 5
    X0 = common.random_point(r, ds.d, ds.n, 11, lambda r_, d_, n_, M: da.manifold_project(r_, d_, n_, M))
    cost, gn, sel, X = single(da, ds, R, r, iters, mode, X0)
    res = run_ranks(str(tmp_path), world, name, R, r, iters, mode, X0, transport, wait)
    want_mode = 2 if transport == "staged" else 1
    for k, o in enumerate(res):
        assert int(o["mode"]) == want_mode, "rank %d used transport %d" % (k, int(o["mode"]))
        own_gpu = da.device_count() >= world  # tests/exchange_worker.py: device = rank % device_count()
        expect = wait or ("device" if own_gpu else "host")
        assert str(o["wait"]).startswith(expect), (str(o["wait"]), expect)
        assert int(o["device"]) == k % max(da.device_count(), 1)
        if want_mode == 1:  # the halo buffers other ranks store into are fine-grained device memory
            assert bool(o["finegrained"]), "rank %d fell back to a coarse-grained halo buffer" % k
        assert np.array_equal(o["selected"], sel), (k, o["selected"], sel)
        assert np.allclose(o["cost"], cost, rtol=1e-11, atol=0), (k, np.max(np.abs(o["cost"] - cost) / np.abs(cost)))
        assert np.allclose(o["gradnorm"], gn, rtol=1e-9, atol=0)
        assert np.array_equal(o["X"], X), "rank %d: iterates differ from the single session (max %g)" % (
            k, np.max(np.abs(o["X"] - X)))
        # every rank reads the same evaluation scalars: identical traces on all of them
        assert np.array_equal(o["cost"], res[0]["cost"]) and np.array_equal(o["gradnorm"], res[0]["gradnorm"])
    # neighbour-only traffic: ranks that host agents post, and only to ranks hosting their neighbours
    assert sum(int(o["posts"]) for o in res) > 0
    assert all(int(o["peers"]) <= world - 1 for o in res)


@pytest.mark.parametrize("name,R,world,iters", [("sphere2500", 5, 2, 25), ("torus3D", 8, 4, 12), ("sphere2500", 5, 4, 8)])
def test_certification_across_ranks_matches_one_gpu(tmp_path, name, R, world, iters):
    """dcora_exchange_certify (SURVEY 8(e) "Collective": row-block S v with the halo exchange, inner products summed
    over the ranks) against fastVerification of the same iterate on one GPU: same verdict, lambda_min and theta, and
    the eigenvector up to its sign.  A few RBCD iterations from a random start leave a saddle direction, so the
    eigenpair -- the distributed part -- is what gets compared."""
    import dcora_amd as da
    ds = common.product_dataset(name)
    r, eta = 5, 1e-3
    X0 = common.random_point(r, ds.d, ds.n, 11, lambda r_, d_, n_, M: da.manifold_project(r_, d_, n_, M))
    res = run_ranks(str(tmp_path), world, name, R, r, iters, "greedy", X0, certify=eta)
    X = res[0]["X"]
    Q = da.build_Q_pgo(ds)
    S = da.dual_certificate(r, ds.d, ds.n, X, Q)
    psd, theta, v, lmin = da.fast_verification(S, eta, block=ds.d + 1)
    A = S.to_scipy()
    for k, o in enumerate(res):
        assert bool(o["cert_ok"]) == bool(psd) == False
        assert bool(o["cert_distributed"]), "rank %d: the row-block Lanczos runs did not converge" % k
        assert int(o["cert_matvecs"]) > 0
        lam, th, vv = float(o["cert_lambda"]), float(o["cert_theta"]), o["cert_v"]
        assert abs(lam - lmin) <= 1e-6 * max(1.0, abs(lmin)), (lam, lmin)
        assert abs(np.linalg.norm(vv) - 1) < 1e-12
        assert abs(th - vv @ (A @ vv)) <= 1e-9 * max(1.0, abs(th))          # theta is the curvature along v
        assert abs(th - theta) <= 1e-5 * max(1.0, abs(theta)), (th, theta)
        assert min(np.linalg.norm(vv - v), np.linalg.norm(vv + v)) < 1e-3
        # (S + eta I) v = lambda v to the tolerance of the run
        assert np.linalg.norm(A @ vv + eta * vv - lam * vv) < 1e-3 * max(1.0, abs(lam))
        # every rank holds the same bits
        assert np.array_equal(vv, res[0]["cert_v"]) and lam == float(res[0]["cert_lambda"])


class Hip:
    """the few HIP runtime calls the in-process transport below needs, from the runtime the library already loaded"""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]

    def malloc(self, nbytes):
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), nbytes) == 0
        return p.value

    def d2d(self, dst, src, nbytes):
        assert self.rt.hipMemcpy(dst, src, nbytes, 3) == 0

    def d2h(self, dst_np, src, nbytes):
        assert self.rt.hipMemcpy(dst_np.ctypes.data, src, nbytes, 2) == 0

    def free(self, p):
        self.rt.hipFree(p)


def test_two_sessions_pack_unpack_one_process():
    """rank 0 / rank 1 sessions of world_size 2 in ONE process: pack_public_dev -> device-to-device copy ->
    unpack_public_dev between the phases reproduces the single session bit for bit (the pieces a host with its own
    transport would use)"""
    import dcora_amd as da
    name, R, r, iters = "sphere2500", 5, 5, 35
    ds = common.product_dataset(name)
    X0 = common.random_point(r, ds.d, ds.n, 5, lambda r_, d_, n_, M: da.manifold_project(r_, d_, n_, M))
    cost, gn, sel, X = single(da, ds, R, r, iters, "greedy", X0)
    world = 2
    ss = [da.RbcdSession(ds, num_robots=R, r=r, rank=k, world_size=world) for k in range(world)]
    per = (R + world - 1) // world
    owner = [a // per for a in range(R)]
    dh = ds.d + 1
    slot = 8 * r * dh * max(ss[0].public_count(a) for a in range(R))
    hip = Hip()
    send, recv = hip.malloc(slot), hip.malloc(slot)
    evs = [hip.malloc(16 * R) for _ in range(world)]
    for s in ss:
        s.set_X(X0)

    def move(a):
        src = ss[owner[a]]
        src.pack_public_dev(a, send)
        src.synchronize()
        hip.d2d(recv, send, slot)  # the device-to-device copy a transport would make
        for k, s in enumerate(ss):
            if k != owner[a]:
                s.unpack_public_dev(a, recv)
                s.synchronize()

    selected = 0
    tr_cost, tr_sel = [], []
    for _ in range(iters):
        for s in ss:
            s.phase_nonselected(selected)
        for a in range(R):
            if a != selected:
                move(a)
        for s in ss:
            s.phase_selected(selected)
        move(selected)
        tot = np.zeros(2 * R)
        for k, s in enumerate(ss):
            s.phase_evaluate_dev(evs[k])
            s.synchronize()
            h = np.zeros(2 * R)
            hip.d2h(h, evs[k], 16 * R)
            tot += h
        tr_cost.append(float(tot[1::2].sum()))
        tr_sel.append(selected)
        selected = int(np.argmax(np.sqrt(tot[0::2])))
    assert np.array_equal(tr_sel, sel)
    assert np.allclose(tr_cost, cost, rtol=1e-11, atol=0)
    Xs = [s.get_X() for s in ss]
    for a in range(R):
        n_a = ds.n // R if a < R - 1 else ds.n - (R - 1) * (ds.n // R)
        c0 = a * (ds.n // R) * dh
        blk = slice(c0, c0 + n_a * dh)
        assert np.array_equal(Xs[owner[a]][:, blk], X[:, blk]), "agent %d differs" % a
    for s in ss:
        s.close()
    for p in [send, recv] + evs:
        hip.free(p)


@pytest.mark.parametrize("faults,wait", [(0, None), (1, "device"), (2, "device"), (1, None)])
def test_link_check_steps_down_together(tmp_path, faults, wait):
    """dcora_exchange_create ends with a link check: a 4 KB pattern + flag per neighbour through the transport and the
    wait about to be used.  With injected failures of the last rank's first rounds all ranks step down the same ladder
    (device-side wait -> host wait -> shared host segment) and the run still reproduces the single session bit for bit."""
    import dcora_amd as da
    name, R, world, iters, r = "sphere2500", 5, 2, 5, 5
    ds = common.product_dataset(name)
    X0 = common.random_point(r, ds.d, ds.n, 11, lambda r_, d_, n_, M: da.manifold_project(r_, d_, n_, M))
    cost, gn, sel, X = single(da, ds, R, r, iters, "greedy", X0)
    res = run_ranks(str(tmp_path), world, name, R, r, iters, "greedy", X0, None, wait,
                    extra_env={"DCORA_TEST_PROBE_FAULT": str(faults)})
    for k, o in enumerate(res):
        assert int(o["link_rounds"]) == faults + 1, (k, int(o["link_rounds"]))
        started_on_device = str(wait) == "device" or (wait is None and da.device_count() >= world)
        ladder = (["device"] if started_on_device else []) + ["host", "staged"]
        now = ladder[min(faults, len(ladder) - 1)]
        assert int(o["mode"]) == (2 if now == "staged" else 1), (k, now, int(o["mode"]))
        assert bool(o["link_no_ipc"]) == (now == "staged")
        if faults == 0:
            assert not bool(o["link_no_device_wait"]) and not bool(o["link_no_ipc"]) and float(o["link_us"]) < 2e6
        assert np.array_equal(o["X"], X) and np.array_equal(o["selected"], sel)


@pytest.mark.parametrize("faults", [0, 1, 2])
def test_link_check_ladder_with_ranks_that_wait_differently(tmp_path, faults):
    """the form of the wait is rank-local (ranks sharing a GPU spin on the host, the others poll on the
    device).  The step down must come from the ranks' shared votes: rank 0 polling on the device and rank 1 waiting on
    the host used to leave the ladder on different rungs (mixed transports, then DCORA_ERR_EXCHANGE_LINK).  Now: device
    wait anywhere -> host wait everywhere -> staged, on every rank alike, and the run equals the single session."""
    import dcora_amd as da
    name, R, world, iters, r = "sphere2500", 5, 2, 5, 5
    ds = common.product_dataset(name)
    X0 = common.random_point(r, ds.d, ds.n, 11, lambda r_, d_, n_, M: da.manifold_project(r_, d_, n_, M))
    cost, gn, sel, X = single(da, ds, R, r, iters, "greedy", X0)
    res = run_ranks(str(tmp_path), world, name, R, r, iters, "greedy", X0, None, None,
                    extra_env={"DCORA_TEST_PROBE_FAULT": str(faults), "DCORA_TEST_WAIT_BY_RANK": "device,host"})
    for k, o in enumerate(res):
        assert int(o["link_rounds"]) == faults + 1, (k, int(o["link_rounds"]))
        assert int(o["mode"]) == (2 if faults == 2 else 1), (k, int(o["mode"]))
        assert bool(o["link_no_ipc"]) == (faults == 2)
        if faults == 0:
            assert str(o["wait"]).startswith("device" if k == 0 else "host")
        else:
            assert str(o["wait"]).startswith("host")
        assert np.array_equal(o["X"], X) and np.array_equal(o["selected"], sel)


def test_link_check_that_cannot_pass_fails_fast_on_every_rank(tmp_path):
    """no transport passes the check: every rank returns DCORA_ERR_EXCHANGE_LINK within seconds -- never a hang in the
    first post"""
    import dcora_amd as da
    name, R, world, r = "sphere2500", 5, 2, 5
    ds = common.product_dataset(name)
    X0 = common.random_point(r, ds.d, ds.n, 11, lambda r_, d_, n_, M: da.manifold_project(r_, d_, n_, M))
    res = run_ranks(str(tmp_path), world, name, R, r, 1, "greedy", X0, None, None,
                    extra_env={"DCORA_TEST_PROBE_FAULT": "9", "DCORA_TEST_EXPECT_LINK_ERROR": "1"})
    for o in res:
        assert "link check" in str(o["error"]) and float(o["seconds"]) < 60


def test_a_usage_error_of_certify_does_not_poison_the_job(tmp_path):
    """certify without the global Q on rank 0: every rank gets the usage error, the exchange goes on working and a
    proper certify afterwards gives the one-GPU verdict (such errors used to mark the job as failed)"""
    import dcora_amd as da
    name, R, world, iters, r = "sphere2500", 5, 2, 3, 5
    ds = common.product_dataset(name)
    X0 = common.random_point(r, ds.d, ds.n, 11, lambda r_, d_, n_, M: da.manifold_project(r_, d_, n_, M))
    res = run_ranks(str(tmp_path), world, name, R, r, iters, "greedy", X0, certify=1e-3,
                    extra_env={"DCORA_TEST_CERTIFY_USAGE": "1"})
    for o in res:
        assert not bool(o["cert_ok"]) and int(o["cert_matvecs"]) > 0
