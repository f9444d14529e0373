"""one rank of a multi-process RBCD run through the library's neighbour exchange (dcora_exchange_*); started by
tests/test_exchange_gpu.py and by nothing else.  argv: rank world job dataset R r iters out_dir mode"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    job, name = sys.argv[3], sys.argv[4]
    R, r, iters = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    out_dir, mode = sys.argv[8], sys.argv[9]
    if os.environ.get("DCORA_TEST_WAIT_BY_RANK"):
        # ranks that differ in the form of their wait (what ranks sharing a GPU beside ranks with a GPU of their own do by
        # default): set before the library reads its environment
        os.environ["DCORA_EXCHANGE_WAIT"] = os.environ["DCORA_TEST_WAIT_BY_RANK"].split(",")[rank]
    import common
    import dcora_amd as da
    ds = common.product_dataset(name)
    X0 = np.load(os.path.join(out_dir, "X0.npy"))
    accel = mode == "greedy"
    # one GPU per rank where the box has them (peer access + cross-device IPC over xGMI); all ranks on device 0 otherwise
    device = rank % max(da.device_count(), 1)
    s = da.RbcdSession(ds, num_robots=R, r=r, acceleration=accel, rank=rank, world_size=world, device=device)
    if os.environ.get("DCORA_TEST_PROBE_FAULT"):
        # the library's test hook (not an environment switch of the library): the last rank reports its first n rounds
        # of the link check as failed
        from dcora_amd import capi
        assert capi.lib().dcora_debug_exchange_probe_fault(int(os.environ["DCORA_TEST_PROBE_FAULT"])) == 0
    if os.environ.get("DCORA_TEST_EXPECT_LINK_ERROR"):
        # the link check is made to fail on every transport: every rank gets the distinct error code, quickly
        import time
        t0 = time.time()
        try:
            da.Exchange(s, job)
        except Exception as e:
            np.savez(os.path.join(out_dir, "rank%d.npz" % rank), error=str(e), seconds=time.time() - t0)
            s.close()
            return
        raise SystemExit("the exchange was created although its link check cannot pass")
    ex = da.Exchange(s, job)
    link = ex.link_report()
    ex.set_X(X0)
    cost, gn, sel = [], [], []
    if mode == "greedy":
        selected = 0
        for _ in range(iters):
            c2, g, bn, nxt = ex.iterate(selected)
            cost.append(c2)
            gn.append(g)
            sel.append(selected)
            selected = nxt
    else:  # coloured ticks, one evaluation per sweep
        col, nc = s.colours()
        for _ in range(iters):
            for c in range(nc):
                ex.tick(np.flatnonzero(col == c).astype(np.int32))
            c2, g, bn, nxt = ex.evaluate()
            cost.append(c2)
            gn.append(g)
            sel.append(nxt)
    # the team's termination condition: everybody ready only when every rank says so
    assert ex.all_ready(True) is True
    assert ex.all_ready(rank != world - 1) is False
    cert = None
    if os.environ.get("DCORA_TEST_CERTIFY_USAGE"):
        # a usage error of certify (no global Q on rank 0) is reported on every rank and does NOT poison the job: the
        # exchange keeps working afterwards
        try:
            ex.certify(None, 1e-3, (ds.d + 1) * ds.n)
            raise SystemExit("certify accepted a call without the global Q")
        except RuntimeError as e:
            assert "global Q" in str(e), str(e)
        c2, g, bn, nxt = ex.evaluate()
        assert np.isfinite(c2) and np.isfinite(g)
    if os.environ.get("DCORA_TEST_CERTIFY"):
        eta = float(os.environ["DCORA_TEST_CERTIFY"])
        Q = da.build_Q_pgo(ds) if rank == 0 else None
        cert = ex.certify(Q, eta, (ds.d + 1) * ds.n)
    X = ex.gather_X()
    info = ex.info()
    ex.barrier()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), cost=np.array(cost), gradnorm=np.array(gn),
             selected=np.array(sel), X=X, mode=info["mode"], posts=info["posts"], waits=info["waits"],
             bytes_posted=info["bytes_posted"], peers=info["peers"], finegrained=info["halo_finegrained"],
             wait=info["wait"], device=device, link_rounds=link["rounds"], link_no_device_wait=link["gave_up_device_wait"],
             link_no_ipc=link["gave_up_ipc"], link_us=link["last_round_us"],
             **({} if cert is None else dict(cert_ok=cert[0], cert_theta=cert[1], cert_lambda=cert[2], cert_v=cert[3],
                                             cert_matvecs=cert[4], cert_distributed=cert[5])))
    ex.close()
    s.close()


if __name__ == "__main__":
    main()
