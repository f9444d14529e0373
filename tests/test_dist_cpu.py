"""world_size-2 gloo test (CPU) of the one-process-per-GPU RBCD protocol used by bench.py: ownership a // ceil(R / world),
public-pose pack -> all_gather (pull) / broadcast (push) -> unpack, block-wise evaluation + all-reduce, greedy selection.  The GPU session is
replaced by a numpy model with the same interface whose local solve is the oracle, so the distributed result must
equal the single-process oracle run bit-for-bit in its selection sequence."""
import os
import time
import sys

import numpy as np
import pytest

import common

WORLD = 2


def _worker(rank, world, port, tmpdir, mode="greedy"):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(common.HERE))
    sys.path.insert(0, common.HERE)
    import g2o_np
    from oracle import orc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    name, R, r, iters = "smallGrid3D", 5, 5, 12
    ds = common.oracle_dataset(name)
    g = g2o_np.read_g2o(common.data_path(name))
    d, n, dh = ds.d, ds.n, ds.d + 1
    per = n // R
    start = [b * per for b in range(R)]
    end = [n if b == R - 1 else (b + 1) * per for b in range(R)]
    robot = lambda i: min(i // per, R - 1)
    Qg = g2o_np.dense_Q(g)
    import scipy.sparse as sp
    pub = [set() for _ in range(R)]
    for (i, j, *_r) in g["edges"]:
        if robot(i) != robot(j):
            pub[robot(i)].add(i)
            pub[robot(j)].add(j)
    pub = [sorted(p) for p in pub]
    cols = lambda b: slice(start[b] * dh, end[b] * dh)
    X0 = np.load(os.path.join(tmpdir, "X0.npy"))
    X = X0.copy()                      # every rank keeps a mirror; only owned blocks + public poses are authoritative
    per_rank = (R + world - 1) // world      # agent a: rank a // per_rank, slot a % per_rank (as bench.py RankDriver)
    owner = lambda a: a // per_rank
    hosted = [b for b in range(R) if owner(b) == rank]
    probs = {}
    for b in hosted:
        Qbb = orc.CSR.from_scipy(sp.csr_matrix(Qg[cols(b), cols(b)]))
        probs[b] = Qbb
    V = {b: X[:, cols(b)].copy() for b in hosted}
    gamma = alpha = 0.0
    trace = []
    selected = 0

    def G_of(b):
        Qcb = Qg[:, cols(b)].copy()
        Qcb[cols(b), :] = 0
        return X @ Qcb

    pidx = [np.array([p * dh + c for p in pub[a] for c in range(dh)], dtype=np.int64) for a in range(R)]
    slot = r * max(len(i) for i in pidx)

    def pull_all_but(sel):
        mine = torch.zeros(per_rank * slot, dtype=torch.float64)
        for a in hosted:
            if a != sel:
                v = np.ascontiguousarray(X[:, pidx[a]].T).reshape(-1)
                mine[(a % per_rank) * slot:(a % per_rank) * slot + v.size] = torch.from_numpy(v.copy())
        parts = [torch.zeros(per_rank * slot, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, mine)
        for a in range(R):
            if a != sel and owner(a) != rank:
                v = parts[owner(a)][(a % per_rank) * slot:(a % per_rank) * slot + r * len(pidx[a])].numpy()
                X[:, pidx[a]] = v.reshape(len(pidx[a]), r).T

    def exchange_set(agents):
        mine = torch.zeros(per_rank * slot, dtype=torch.float64)
        for a in agents:
            if a in hosted:
                v = np.ascontiguousarray(X[:, pidx[a]].T).reshape(-1)
                mine[(a % per_rank) * slot:(a % per_rank) * slot + v.size] = torch.from_numpy(v.copy())
        parts = [torch.zeros(per_rank * slot, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, mine)
        for a in agents:
            if owner(a) != rank:
                v = parts[owner(a)][(a % per_rank) * slot:(a % per_rank) * slot + r * len(pidx[a])].numpy()
                X[:, pidx[a]] = v.reshape(len(pidx[a]), r).T

    def push(a):
        buf = torch.zeros(r * len(pidx[a]), dtype=torch.float64)
        if owner(a) == rank:
            buf = torch.from_numpy(np.ascontiguousarray(X[:, pidx[a]].T).reshape(-1).copy())
        dist.broadcast(buf, src=owner(a))
        if owner(a) != rank:
            X[:, pidx[a]] = buf.numpy().reshape(len(pidx[a]), r).T

    def evaluate():
        ev = torch.zeros(2 * R, dtype=torch.float64)
        for b in hosted:
            nb = end[b] - start[b]
            Xb = X[:, cols(b)]
            EG = Xb @ Qg[cols(b), cols(b)] + G_of(b)
            RG = orc.tangent_project(r, d, nb, Xb, EG)
            ev[2 * b] = float(np.sum(RG * RG))
            ev[2 * b + 1] = float(np.sum(Xb * EG))
        dist.all_reduce(ev)
        return ev.numpy()

    if mode == "coloured":
        # bench.py RankDriver.tick: the agents of one colour solve at the same time on their ranks from one snapshot
        # of the neighbour states (non-accelerated), then ONE all_gather moves their public poses
        sets = np.load(os.path.join(tmpdir, "sets.npy"), allow_pickle=True)
        for sweep in range(2):
            for S in sets:
                new = {}
                for b in S:
                    if b in hosted:
                        P = orc.Problem(r, d, end[b] - start[b], probs[b], G=G_of(b))
                        new[b] = P.optimize(X[:, cols(b)])[0]
                for b, Xn in new.items():
                    X[:, cols(b)] = Xn
                exchange_set(list(S))
            h = evaluate()
            trace.append((sweep, float(h[1::2].sum()), float(np.sqrt(h[0::2].sum()))))
        iters = 0
    for it in range(iters):
        gamma = (1 + np.sqrt(1 + 4.0 * R * R * gamma * gamma)) / (2.0 * R)
        alpha = 1.0 / (gamma * R)
        for b in hosted:
            if b == selected:
                continue
            nb = end[b] - start[b]
            Y = orc.project_to_manifold(r, d, nb, (1 - alpha) * X[:, cols(b)] + alpha * V[b])
            X[:, cols(b)] = Y
            V[b] = orc.project_to_manifold(r, d, nb, V[b])
        pull_all_but(selected)
        if selected in hosted:
            b = selected
            nb = end[b] - start[b]
            Y = orc.project_to_manifold(r, d, nb, (1 - alpha) * X[:, cols(b)] + alpha * V[b])
            P = orc.Problem(r, d, nb, probs[b], G=G_of(b))
            Xn, _ = P.optimize(Y)
            V[b] = orc.project_to_manifold(r, d, nb, V[b] + gamma * (Xn - Y))
            X[:, cols(b)] = Xn
        push(selected)
        ev = torch.zeros(2 * R, dtype=torch.float64)
        for b in hosted:
            nb = end[b] - start[b]
            Xb = X[:, cols(b)]
            EG = Xb @ Qg[cols(b), cols(b)] + G_of(b)
            RG = orc.tangent_project(r, d, nb, Xb, EG)
            ev[2 * b] = float(np.sum(RG * RG))
            ev[2 * b + 1] = float(np.sum(Xb * EG))
        dist.all_reduce(ev)
        h = ev.numpy()
        trace.append((selected, float(h[1::2].sum()), float(np.sqrt(h[0::2].sum()))))
        selected = int(np.argmax(np.sqrt(h[0::2])))
    if rank == 0:
        np.save(os.path.join(tmpdir, "trace.npy"), np.array(trace))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_protocol_matches_single_process(built, tmp_path):
    import torch.multiprocessing as mp
    from oracle import orc
    ds = common.oracle_dataset("smallGrid3D")
    X0 = common.random_point(5, ds.d, ds.n, 3, orc.project_to_manifold)
    np.save(tmp_path / "X0.npy", X0)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path)), nprocs=WORLD, join=True)
    tr = np.load(tmp_path / "trace.npy")
    ref = orc.run_rbcd(ds, X0, num_robots=5, r_min=5, max_iters=12, staircase=0, rgrad_tol=0.0)
    # restart interval 30 is not crossed in 12 iterations
    assert np.array_equal(tr[:, 0].astype(int), ref["selected"])
    assert np.allclose(tr[:, 1], ref["cost"], rtol=1e-9)
    assert np.allclose(tr[:, 2], ref["gradnorm"], rtol=1e-7)


def test_two_rank_coloured_ticks_match_one_after_the_other(built, tmp_path):
    """the N > 1 path of the simultaneous-update mode (dcora_rbcd_iterate_set + one all_gather per tick)"""
    import scipy.sparse as sp
    import torch.multiprocessing as mp
    import g2o_np
    from oracle import orc
    name, R, r = "smallGrid3D", 5, 5
    ds = common.oracle_dataset(name)
    d, n, dh = ds.d, ds.n, ds.d + 1
    X0 = common.random_point(r, d, n, 3, orc.project_to_manifold)
    np.save(tmp_path / "X0.npy", X0)
    g = g2o_np.read_g2o(common.data_path(name))
    per = n // R
    robot = lambda i: min(i // per, R - 1)
    adj = [set() for _ in range(R)]
    for (i, j, *_r) in g["edges"]:
        if robot(i) != robot(j):
            adj[robot(i)].add(robot(j))
            adj[robot(j)].add(robot(i))
    col = []
    for a in range(R):  # the product's rule (dcora_rbcd_agent_colours): smallest colour free among lower neighbours
        used = {col[b] for b in adj[a] if b < a}
        col.append(min(c for c in range(R) if c not in used))
    sets = np.empty(max(col) + 1, dtype=object)
    for c in range(max(col) + 1):
        sets[c] = [a for a in range(R) if col[a] == c]
    np.save(tmp_path / "sets.npy", sets, allow_pickle=True)
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path), "coloured"), nprocs=WORLD, join=True)
    tr = np.load(tmp_path / "trace.npy")
    # single process, the same agents one after the other
    Qg = sp.csc_matrix(g2o_np.dense_Q(g))
    cols = [np.arange(b * per * dh, (n if b == R - 1 else (b + 1) * per) * dh) for b in range(R)]
    X = X0.copy()
    costs = []
    for sweep in range(2):
        for S in sets:
            for b in S:
                own = cols[b]
                C = Qg[:, own].toarray()
                C[own, :] = 0
                P = orc.Problem(r, d, own.size // dh, orc.CSR.from_scipy(sp.csr_matrix(Qg[own][:, own])), G=X @ C)
                X[:, own] = P.optimize(X[:, own])[0]
        costs.append(float(np.sum((X @ Qg) * X)))
    assert np.allclose(tr[:, 1], costs, rtol=1e-10) and costs[1] < costs[0]


# ---- the library's own exchange: host half of the protocol, real processes, no GPU -----------------------------------
def _selftest_rank(rank, world, job, R, rounds, tmpdir):
    import ctypes as C
    sys.path.insert(0, os.path.dirname(common.HERE))
    from dcora_amd import capi
    cs = C.c_double()
    rc = capi.lib().dcora_exchange_host_selftest(job.encode(), rank, world, R, rounds, C.byref(cs))
    msg = capi.lib().dcora_last_error().decode()
    np.save(os.path.join(tmpdir, "cs%d.npy" % rank), np.array([rc, cs.value]))
    if rc:
        raise RuntimeError("rank %d: status %d: %s" % (rank, rc, msg))


@pytest.mark.parametrize("world,R", [(2, 5), (3, 8), (4, 5)])
def test_exchange_host_protocol_between_processes(built, tmp_path, world, R):
    """dcora_exchange_host_selftest in `world` processes: the bootstrap through the POSIX shared segment, the barriers,
    the per-agent flag words with their parity double-buffering and the evaluation all-gather are the code the GPU
    ranks run (dcora_amd/csrc/exchange.hip); host stores stand in for the device's.  (4 ranks / 5 agents: a rank
    without agents takes part in every barrier and wait.)"""
    import uuid
    import torch.multiprocessing as mp
    rounds = 50
    job = "cpu%s" % uuid.uuid4().hex[:10]
    mp.spawn(_selftest_rank, args=(world, job, R, rounds, str(tmp_path)), nprocs=world, join=True)
    want = sum((q + 0.5 * a) * (a + 1) + (0.25 * q - a) for q in range(1, rounds + 1) for a in range(R))
    for k in range(world):
        rc, cs = np.load(tmp_path / ("cs%d.npy" % k))
        assert rc == 0 and abs(cs - want) <= 1e-9 * abs(want), (k, rc, cs, want)
    assert not os.path.exists("/dev/shm/dcora_" + job)  # rank 0 unlinks the name once everybody is attached


def test_exchange_host_protocol_refuses_a_mismatched_job(built, tmp_path, monkeypatch):
    """a rank that attaches with another shape (number of agents) is told so instead of reading foreign slots -- and it
    does NOT write into the segment it refused (under a shared name that segment may be an unrelated
    live job's): the job it did not join gives up on its own timeout, DCORA_EXCHANGE_TIMEOUT_S"""
    import ctypes as C
    import multiprocessing as mp
    import uuid
    from dcora_amd import capi
    monkeypatch.setenv("DCORA_EXCHANGE_TIMEOUT_S", "6")
    job = "cpu%s" % uuid.uuid4().hex[:10]
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_selftest_rank, args=(0, 2, job, 5, 3, str(tmp_path)))
    p.start()
    cs = C.c_double()
    rc = capi.lib().dcora_exchange_host_selftest(job.encode(), 1, 2, 6, 3, C.byref(cs))
    assert rc != 0 and b"another shape" in capi.lib().dcora_last_error()
    p.join(90)
    assert p.exitcode is not None and p.exitcode != 0  # rank 0 times out waiting for its rank 1 (6 s here)


def _late_rank0(world, job, R, rounds, tmpdir, delay):
    time.sleep(delay)
    _selftest_rank(0, world, job, R, rounds, tmpdir)


def test_exchange_bootstrap_does_not_attach_to_a_stale_segment(built, tmp_path):
    """a crashed job left an initialised segment of the same shape under the name; the other ranks of a new job start
    BEFORE rank 0 replaces it: they must recognise it as stale (its creator is gone, the name moves on) and attach to
    the new one -- the round-2 bootstrap mapped the stale inode and every rank sat in the barrier until its time-out"""
    import multiprocessing as mp
    import uuid
    from dcora_amd import capi
    world, R, rounds = 3, 5, 20
    job = "cpu%s" % uuid.uuid4().hex[:10]
    assert capi.lib().dcora_debug_exchange_leave_stale(job.encode(), world, R) == 0
    assert os.path.exists("/dev/shm/dcora_" + job)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_selftest_rank, args=(k, world, job, R, rounds, str(tmp_path))) for k in (1, 2)]
    for p in procs:
        p.start()
    late = ctx.Process(target=_late_rank0, args=(world, job, R, rounds, str(tmp_path), 1.0))
    late.start()
    for p in procs + [late]:
        p.join(60)
        assert p.exitcode == 0
    want = sum((q + 0.5 * a) * (a + 1) + (0.25 * q - a) for q in range(1, rounds + 1) for a in range(R))
    for k in range(world):
        rc, cs = np.load(tmp_path / ("cs%d.npy" % k))
        assert rc == 0 and abs(cs - want) <= 1e-9 * abs(want)
    assert not os.path.exists("/dev/shm/dcora_" + job)


def _selftest_rank_slow(rank, world, job, R, rounds, tmpdir, timeout_s):
    os.environ["DCORA_EXCHANGE_TIMEOUT_S"] = str(timeout_s)
    _selftest_rank(rank, world, job, R, rounds, tmpdir)


def test_a_rank_that_dies_takes_the_others_down_within_the_timeout(built, tmp_path):
    """one of three ranks is killed in the middle of a long run: the other two must notice (the posts / heartbeats of
    the dead rank never arrive), raise the job's failure flag and exit NON-ZERO within DCORA_EXCHANGE_TIMEOUT_S --
    never hang (a first multi-GPU run must fail fast)"""
    import multiprocessing as mp
    import signal
    import uuid
    world, R, rounds, timeout_s = 3, 6, 50_000_000, 3.0
    job = "cpu%s" % uuid.uuid4().hex[:10]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_selftest_rank_slow, args=(k, world, job, R, rounds, str(tmp_path), timeout_s))
             for k in range(world)]
    for p in procs:
        p.start()
    time.sleep(2.0)  # bootstrap done, the rounds are running
    assert all(p.is_alive() for p in procs), "the run ended before a rank could be killed"
    os.kill(procs[1].pid, signal.SIGKILL)
    t0 = time.time()
    for k in (0, 2):
        procs[k].join(timeout_s + 20)
        assert procs[k].exitcode is not None, "rank %d still runs %.0f s after rank 1 died" % (k, time.time() - t0)
        assert procs[k].exitcode != 0
    procs[1].join(5)
    assert time.time() - t0 < timeout_s + 20
    for k in (0, 2):
        rc, _ = np.load(tmp_path / ("cs%d.npy" % k))
        assert rc != 0
