"""one rank of a multi-process range-aided RBCD run through the library's exchange (dcora_exchange_create_ra); started by
tests/test_ra_exchange_gpu.py and by nothing else.  argv: rank world job dataset r iters out_dir accel restart [eta]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    job, name = sys.argv[3], sys.argv[4]
    r, iters, out_dir = int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    accel, restart = bool(int(sys.argv[8])), int(sys.argv[9])
    import dcora_amd as da
    from test_raslam import ra_path
    ra = da.RADataset(ra_path(name))
    X0 = np.load(os.path.join(out_dir, "X0.npy"))
    device = rank % max(da.device_count(), 1)
    s = da.RaRbcdSession(ra, r, acceleration=accel, restart_interval=restart, rank=rank, world_size=world, device=device)
    ex = da.Exchange(s, job)
    ex.set_X(X0)
    cost, gn, sel = [], [], []
    selected = 0
    for _ in range(iters):
        c2, g, bn, nxt = ex.iterate(selected)
        cost.append(c2)
        gn.append(g)
        sel.append(selected)
        selected = nxt
    cert = None
    if len(sys.argv) > 10:  # fastVerification across the ranks: the global data matrix on rank 0 only
        cert = ex.certify(ra.Q if rank == 0 else None, float(sys.argv[10]), ra.k)
    X = ex.gather_X()
    info = ex.info()
    ex.barrier()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), cost=np.array(cost), gradnorm=np.array(gn),
             selected=np.array(sel), X=X, mode=info["mode"], posts=info["posts"], peers=info["peers"], wait=info["wait"],
             **({} if cert is None else dict(cert_ok=cert[0], cert_theta=cert[1], cert_lambda=cert[2], cert_v=cert[3],
                                             cert_matvecs=cert[4], cert_distributed=cert[5])))
    ex.close()
    s.close()


if __name__ == "__main__":
    main()
