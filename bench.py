#!/usr/bin/env python3
"""bench.py -- RBCD iterations/sec of the MI355X path on BASELINE.json's headline workload.

Workload (config.workload): sphere2500.g2o split across 5 agents (contiguous n/R partition of the reference
driver), rank r = 5, RBCD++ with Nesterov acceleration and restarts every 30, local solver = RTR with the
reference defaults (3 outer iterations, <= 50 tCG, Delta0 = 100, tol 1e-2).  One "step" = one pass of the loop
body of examples/MultiRobotExample.cpp:223-307 (non-selected Nesterov updates, public-pose pull, the selected
agent's QuadraticOptimizer::optimize, central cost / gradient evaluation, greedy selection).  The start point is
a seeded uniform-random matrix projected onto the manifold, as the reference driver's Random initialisation;
all inputs (Q, preconditioner, X) are resident in HBM when the timed region starts.

Launch: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the script expects to run under
torch.distributed.run (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment); started
bare with --gpus N > 1 it launches that itself as a child process and relays the result line.  Agent a lives on
rank a // ceil(R / N).  The data path between ranks is the library's neighbour exchange (dcora_exchange_*: peer
stores into IPC-mapped halo buffers over xGMI, flag words and the 2R evaluation scalars in a shared host segment);
torch.distributed (backend nccl = RCCL) only brackets the timed region (barrier, MAX over ranks).

Prints ONE JSON line on rank 0 (see the task contract) with two extra objects:
  roofline     -- the dominant kernel of the timed loop (k_fused_pc: step + vector updates + the dense preconditioner
                  product + projection of a tCG iteration in one launch) timed live with HIP events
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference algorithm) on the same iteration window of the
                  same workload, 1 host thread (the reference ships single-threaded: OpenMP off); r_threads: the same
                  with one host thread per agent (what the reference's asynchronous mode starts)
`value` is the MEDIAN of REPLAYS replays of the driver's window (value_samples lists them).
"""
import argparse
import json
import os
import subprocess
import sys
import time
import uuid

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md; about 6.3 TB/s is achievable by a stream)
SUSTAINED_STEPS = 300


CACHE_PATH = os.path.join(ROOT, "gpurun_out", "bench_n1_cache.json")
SCALING_R = 16  # ONE partition of the 100k lattice at every N: the strong-scaling series


def provenance(args):
    """what an N = 1 cache must match before an N > 1 line may quote it: the code (git HEAD, else the sha of this file),
    the workload and the box"""
    import hashlib
    import socket
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:
        head = ""
    with open(os.path.abspath(__file__), "rb") as fh:
        sha = hashlib.sha256(fh.read()).hexdigest()[:16]
    try:
        with open(os.path.join(ROOT, "dcora_amd", "lib", "libdcora_hip.so"), "rb") as fh:
            lib = hashlib.sha256(fh.read()).hexdigest()[:16]
    except Exception:
        lib = ""
    return {"git_head": head, "bench_py_sha16": sha, "lib_sha16": lib, "dataset": args.dataset, "rank_r": args.rank_r,
            "robots": args.robots, "steps": args.steps, "warmup": args.warmup, "hostname": socket.gethostname()}


def cache_store(obj):
    """the N = 1 run leaves what the N > 1 lines quote beside their own numbers (the CPU baseline, the one-GPU
    denominators of the scaling ratios): best effort, never an error"""
    try:
        os.makedirs(os.path.dirname(CACHE_PATH), exist_ok=True)
        obj = dict(obj)
        obj["written_at"] = time.time()
        with open(CACHE_PATH, "w") as fh:
            json.dump(obj, fh)
    except Exception:
        pass


def cache_load(args, max_age_s=6 * 3600):
    """the cache of the N = 1 run -- only when it was written by THIS code on THIS workload on THIS box, recently; a
    mismatch returns (None, reason) and the N > 1 line says so instead of quoting a stale figure"""
    try:
        with open(CACHE_PATH) as fh:
            c = json.load(fh)
    except Exception:
        return None, "no N = 1 run of this bench left gpurun_out/bench_n1_cache.json on this box"
    want, have = provenance(args), c.get("provenance") or {}
    for key, val in want.items():
        if key == "git_head" and (not val or not have.get(key)):
            continue  # no git on the box: bench_py_sha16 + lib_sha16 stand in
        if have.get(key) != val:
            return None, "N = 1 cache ignored: %s differs (%r there, %r here)" % (key, have.get(key), val)
    age = time.time() - float(c.get("written_at", 0))
    if age > max_age_s or age < 0:
        return None, "N = 1 cache ignored: written %.0f s ago" % age
    c["provenance"]["age_s"] = age
    return c, None


def scaling_block(world, strong, c5, cached, cache_note=None):
    """The workload that CAN scale, named at the top level of the line at every N: coloured simultaneous ticks over the
    100k-pose lattice split into R = 16 agents AT EVERY N (agent a on rank a // (16 / N): both colours on every rank),
    same graph, start point and sweeps -- one workload, so sweeps/s at N over sweeps/s at N = 1 is its strong scaling.
    `compact` is what the printed line carries; the R = 2 N series of earlier rounds stays as `secondary`."""
    key = "R=%d" % SCALING_R
    out = {"workload": "synthetic 50x50x40 SE(3) lattice (100000 poses, seed 20250310), r = 5, coloured ticks, R = %d "
                       "agents at every N; same graph, start point and number of sweeps" % SCALING_R,
           "n_gpus": world, "scaling": "strong",
           "how_to_read": "speed-up(N) = sweeps_per_s at N GPUs / sweeps_per_s of the N = 1 line (same R = %d partition)"
                          % SCALING_R}
    compact = {"workload": "100k-pose SE(3) lattice, r=5, coloured ticks, R=%d agents at every N" % SCALING_R,
               "agents": SCALING_R, "n_gpus": world, "unit": "sweeps/s"}
    if strong is not None:
        series = strong.get("one_gpu") if world == 1 else strong
        e = (series or {}).get(key) or {}
        out["coloured_ticks"] = e
        compact["sweeps_per_s"] = e.get("sweeps_per_s")
        compact["block_updates_per_s"] = e.get("block_updates_per_s")
        if e.get("error"):
            compact["error"] = _short(e["error"], 120)
        if world == 1:
            out["one_gpu"] = strong.get("one_gpu")
            compact["secondary_one_gpu_sweeps_per_s"] = {k: _g(v, "sweeps_per_s") for k, v in (series or {}).items()
                                                         if k != key}
        else:
            one = _g(cached, "strong_one_gpu", key, "sweeps_per_s")
            if one and e.get("sweeps_per_s"):
                out["one_gpu_same_R"] = {"sweeps_per_s": one, "provenance": (cached or {}).get("provenance")}
                compact["one_gpu_sweeps_per_s"] = one
                compact["speedup_vs_one_gpu"] = e["sweeps_per_s"] / one
            else:
                compact["one_gpu_sweeps_per_s"] = None
                compact["one_gpu_note"] = _short(cache_note or "the N = 1 line of this bench holds the denominator", 160)
            k2 = "R=%d" % (2 * world)
            if k2 != key and k2 in strong:
                out["secondary_R_2N"] = strong[k2]
                compact["secondary_R_2N"] = {"agents": 2 * world, "sweeps_per_s": _g(strong[k2], "sweeps_per_s"),
                                             "one_gpu_sweeps_per_s": _g(cached, "strong_one_gpu", k2, "sweeps_per_s")}
    if c5 is not None:
        out["sequential_rbcd_8_agents"] = {k: c5.get(k) for k in ("value", "unit", "ms_per_step", "parallelism", "exchange",
                                                                  "staircase_ranks", "error") if k in c5}
        compact["sequential_rbcd_8_agents_it_per_s"] = c5.get("value")
    out["compact"] = compact
    return out


LINE_LIMIT = 8000  # characters: the driver reads the LAST stdout line through an 8 KB tail window
DETAIL_PATH = os.path.join(ROOT, "gpurun_out", "bench_detail.json")


def _g(obj, *path):
    """obj[path[0]][path[1]]... or None (the side measurements are optional and may hold {"error": ...})"""
    for key in path:
        if not isinstance(obj, dict) or key not in obj:
            return None
        obj = obj[key]
    return obj


def _short(text, n=200):
    text = "" if text is None else str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def _num(x, digits=8):
    """floats at `digits` significant digits (the detail file keeps full precision)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    return x


KEEP_NULL = ("vs_baseline", "traffic")  # contract keys whose null carries meaning


def _prune(obj):
    """drop None values and round floats, recursively"""
    if isinstance(obj, dict):
        return {k: _prune(v) for k, v in obj.items() if v is not None or k in KEEP_NULL}
    if isinstance(obj, (list, tuple)):
        return [_prune(v) for v in obj]
    return _num(obj)


def compact_line(full, detail_path="gpurun_out/bench_detail.json", limit=LINE_LIMIT):
    """The ONE line bench.py prints last on stdout: the contract's keys, `roofline`, `cpu_baseline`, and one number per
    side measurement -- at most `limit` characters, strict JSON (no NaN / Infinity).  Everything else (`full`, as the
    stages built it) goes to gpurun_out/bench_detail.json and to stderr.  Pure function of a dict: tests/test_bench_line.py
    runs it on the committed lines of earlier rounds without a GPU."""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                     "higher_is_better", "scaling", "vs_baseline", "dtype")}
    line["data"] = _short(full.get("data"), 120)
    cfg = full.get("config") or {}
    line["config"] = {"workload": _short(cfg.get("workload"), 160), "parallelism": _short(cfg.get("parallelism"), 80),
                      "timed_iterations": _short(cfg.get("timed_iterations"), 80),
                      "window_note": "early iterations are cheaper on both sides (tCG exits early): a --steps 20 "
                                     "--warmup 5 window reads ~1.6x the sustained rate; see sustained and "
                                     "cpu_baseline.same_window_as_value",
                      "final_cost_2f": cfg.get("final_cost_2f"), "speedup_vs_cpu_port": cfg.get("speedup_vs_cpu_port")}
    vs = full.get("value_samples")
    if vs:
        line["value_samples"] = {"replays": vs.get("replays"), "statistic": "median", "min": vs.get("min"),
                                 "max": vs.get("max")}
    su = full.get("sustained")
    if su:
        line["sustained"] = {k: su.get(k) for k in ("value", "unit", "steps", "first_iteration", "ms_per_step",
                                                     "speedup_vs_cpu_port")}
    rf = full.get("roofline")
    if rf:
        line["roofline"] = {"bound": rf.get("bound"), "kernel": _short(rf.get("kernel"), 140),
                            "achieved": rf.get("achieved"), "peak": rf.get("peak"), "unit": rf.get("unit"),
                            "frac": rf.get("frac"), "traffic": rf.get("traffic"),
                            "avg_launch_us": rf.get("avg_launch_us"),
                            "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"),
                            "frac_streamed": rf.get("frac_streamed"), "bytes_streamed_per_launch": rf.get("bytes_per_launch"),
                            "measured_stream_triad_GBps": rf.get("measured_stream_triad_GBps"),
                            "note": _short(rf.get("compact_note") or "achieved = SURVEY 8(d) algorithmic bytes / live "
                                           "HIP-event launch time; frac_streamed counts the bytes the kernel streams on "
                                           "purpose; traffic = HBM bytes per launch from the committed PMC passes", 200)}
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                "host_cores_of_the_box": cb.get("host_cores_of_the_box"), "kind": cb.get("kind"),
                                "sample": _short(cb.get("sample"), 200),
                                "same_window_as_value": {"value": _g(cb, "same_window_as_value", "value")},
                                "source": _short(cb.get("source"), 120) if cb.get("source") else None}
    rq = full.get("roofline_qapply")
    if rq:
        line["roofline_qapply"] = {name: {"frac": _g(e, "frac"),
                                          "us": _g(e, "avg_launch_us") or _g(e, "avg_application_us")}
                                   for name, e in rq.items() if isinstance(e, dict)}
    sc = full.get("scaling_100k_lattice")
    if sc:
        line["scaling_100k_lattice"] = sc.get("compact") or {"n_gpus": sc.get("n_gpus")}
        line["scaling_value"] = _g(sc, "compact", "sweeps_per_s")
    col = full.get("coloured_rbcd")
    if isinstance(col, dict):
        line["coloured_rbcd"] = {"block_updates_per_s": col.get("block_updates_per_s"), "error": col.get("error")}
    mc = full.get("ms_to_certified_optimum")
    if mc:
        line["ms_to_certified_optimum"] = {"value": mc.get("total_ms"), "unit": "ms", "certified": mc.get("certified"),
                                           "rank": mc.get("rank"), "final_cost_2f": mc.get("final_cost_2f"),
                                           "cpu_port_ms": _g(mc, "cpu_port", "total_ms"),
                                           "error": _short(mc.get("error"), 120) if mc.get("error") else None}
    c2 = full.get("config2_sphere2500_single")
    if c2:
        line["config2_sphere2500_single"] = {"tcg_iterations_per_s": c2.get("tcg_iterations_per_s"),
                                             "cost_2f": c2.get("cost_2f"), "certified": c2.get("certified"),
                                             "error": _short(c2.get("error"), 120) if c2.get("error") else None}
    c3 = full.get("config3_torus3D_8agents")
    if c3:
        line["config3_torus3D_8agents"] = {"value": c3.get("value"), "unit": c3.get("unit"),
                                           "n_gpus": c3.get("n_gpus"), "cpu_port_value": _g(c3, "cpu_port", "value"),
                                           "error": _short(c3.get("error"), 120) if c3.get("error") else None}
    c4 = full.get("config4_tiers")
    if c4:
        line["config4_tiers"] = {"tcg_iterations_per_s": c4.get("tcg_iterations_per_s"),
                                 "ms_to_certified_optimum": _g(c4, "ms_to_certified_optimum", "value"),
                                 "certified_at_rank": _g(c4, "ms_to_certified_optimum", "certified_at_rank"),
                                 "multi_robot_rbcd_it_per_s": _g(c4, "multi_robot", "value"),
                                 "error": _short(c4.get("error"), 120) if c4.get("error") else None}
    c5 = full.get("config5_lattice100k")
    if c5:
        ranks = c5.get("staircase_ranks") or {}
        line["config5_lattice100k"] = {"value": c5.get("value"), "unit": c5.get("unit"),
                                       "hbm_frac": _g(c5, "hbm_roofline", "frac"),
                                       "staircase_ranks": {k: _g(v, "value") for k, v in ranks.items()},
                                       "cpu_port_value": _g(c5, "cpu_port", "value"),
                                       "error": _short(c5.get("error"), 120) if c5.get("error") else None}
    cc = full.get("config5_central_certified")
    if cc:
        line["config5_central_certified"] = {"seconds_to_certified_optimum": cc.get("seconds_to_certified_optimum"),
                                             "certified": cc.get("certified"), "rank": cc.get("rank"),
                                             "cost_2f": cc.get("cost_2f"),
                                             "from_the_chordal_start_s": _g(cc, "from_the_chordal_start", "seconds_to_certified_optimum"),
                                             "agents_loop_to_rgrad_0p1": {
                                                 mode: {"reached": _g(e, "reached"), "seconds": _g(e, "seconds"),
                                                        "gradnorm": _g(e, "gradnorm")}
                                                 for mode, e in (_g(cc, "from_the_chordal_start",
                                                                    "distributed_loop_to_the_drivers_stopping_rule") or {}).items()
                                                 if isinstance(e, dict)} or None,
                                             "error": _short(cc.get("error"), 120) if cc.get("error") else None}
    ps = full.get("psd_test")
    if ps:
        line["psd_test"] = {"sphere2500_ms": _g(ps, "sphere2500", "repeat_call_ms"),
                            "lattice100k_ms": _g(ps, "lattice100k", "repeat_call_ms"),
                            "lattice100k_numeric_ms": _g(ps, "lattice100k", "numeric_ms"),
                            "error": _short(ps.get("error"), 120) if ps.get("error") else None}
    pg = full.get("process_group")
    if pg:
        line["process_group"] = {k: pg.get(k) for k in ("backend", "ranks", "transport")}
    ex = full.get("exchange")
    if ex:
        line["exchange"] = {"transport": ex.get("transport"), "wait": ex.get("wait"),
                            "link_check_rounds": _g(ex, "link_check", "rounds"),
                            "post_us_per_iteration": ex.get("post_us_per_iteration"),
                            "wait_us_per_iteration": ex.get("wait_us_per_iteration")}
    line["detail"] = detail_path
    line = _prune(line)
    # never above the window: shed the optional blocks, least important first
    for key in ("psd_test", "coloured_rbcd", "value_samples", "exchange", "process_group", "config3_torus3D_8agents",
                "config5_central_certified", "config2_sphere2500_single", "roofline_qapply", "config4_tiers",
                "config5_lattice100k", "ms_to_certified_optimum", "scaling_100k_lattice", "sustained"):
        if len(json.dumps(line, allow_nan=False)) <= limit:
            break
        line.pop(key, None)
    return line


def emit(real_stdout, full):
    """detail -> gpurun_out/bench_detail.json + stderr; the compact line -> the last line of stdout"""
    def clean(o):  # strict JSON: NaN / Infinity become null
        if isinstance(o, dict):
            return {str(k): clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        if isinstance(o, (float, np.floating)):
            return float(o) if np.isfinite(o) else None
        if isinstance(o, np.integer):
            return int(o)
        if isinstance(o, np.ndarray):
            return clean(o.tolist())
        return o
    full = clean(full)
    try:
        os.makedirs(os.path.dirname(DETAIL_PATH), exist_ok=True)
        with open(DETAIL_PATH, "w") as fh:
            json.dump(full, fh)
            fh.write("\n")
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("bench.py: could not write %s: %s\n" % (DETAIL_PATH, e))
    sys.stderr.write("bench.py detail: " + json.dumps(full) + "\n")
    sys.stderr.flush()
    real_stdout.write(json.dumps(compact_line(full), allow_nan=False) + "\n")
    real_stdout.flush()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--dataset", default="sphere2500")
    ap.add_argument("--robots", type=int, default=5)
    ap.add_argument("--rank-r", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config5", action="store_true", help="skip the 100k-pose side measurement")
    ap.add_argument("--no-config4", action="store_true", help="skip the tiers.pyfg side measurement")
    ap.add_argument("--no-coloured", action="store_true",
                    help="skip the coloured simultaneous-update measurements (agents solving on concurrent host threads): "
                         "rocprofv3 --kernel-trace of the WHOLE default run dies in its own buffers there (DESIGN.md 5)")
    ap.add_argument("--no-config3", action="store_true", help="skip the torus3D 8-agent side measurement")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed loop (for a kernel trace of exactly that loop): no side measurements")
    ap.add_argument("--scaling-only", action="store_true",
                    help="the timed loop + the strong-scaling series (scaling_100k_lattice) and nothing else: the "
                         "rehearsal of the driver's N > 1 runs (tests/test_bench_gpu.py)")
    ap.add_argument("--cpu-steps", type=int, default=0, help="oracle iterations for cpu_baseline (0 = auto)")
    ap.add_argument("--cpu-c4-staircase", action="store_true",
                    help="also run the CPU port through the whole tiers.pyfg staircase (minutes)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (this process
    has not touched the GPU and never replaces itself) and relay its stdout"""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def initial_point(da, ds, r, seed=20250310):
    rng = np.random.default_rng(seed)
    M = rng.uniform(-1.0, 1.0, (r, (ds.d + 1) * ds.n))
    return da.manifold_project(r, ds.d, ds.n, M)


def run_single(args, da, torch, ds, X0):
    """the timed window the driver asks for (iterations warmup+2 .. warmup+1+steps of the trajectory from X0) and,
    continuing the same trajectory, a sustained window of SUSTAINED_STEPS iterations"""
    t0 = time.perf_counter()
    s = da.RbcdSession(ds, num_robots=args.robots, r=args.rank_r)
    setup_s = time.perf_counter() - t0

    def warm():
        s.set_X(X0)
        out = s.run(max_iters=args.warmup, rgrad_tol=0.0)
        # continue the trajectory: dcora_rbcd_run restarts selection at agent 0, so drive iterate() from here
        selected = int(out["selected"][-1]) if args.warmup > 0 else 0
        # one untimed pass to recover the greedy choice after warmup
        return s.iterate(selected)[3]

    def window(count, selected):
        s.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c2 = gn = 0.0
        for _ in range(count):
            c2, gn, bn, selected = s.iterate(selected)
        s.synchronize()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, c2, gn, selected

    # the driver's window is a few milliseconds: it is replayed REPLAYS times from the start point (same warm-up, same
    # K iterations of the same trajectory) and `value` is the median; every sample is printed
    samples = []
    for _ in range(REPLAYS):
        selected = warm()
        dt, c2, gn, selected = window(args.steps, selected)
        samples.append(dt)
    dt = float(np.median(samples))
    run_single.samples = samples
    sustained = None
    if not args.headline_only:
        first = args.warmup + 1 + args.steps
        dts, c2s, gns, selected = window(SUSTAINED_STEPS, selected)
        sustained = {"steps": SUSTAINED_STEPS, "first_iteration": first + 1, "value": SUSTAINED_STEPS / dts,
                     "unit": "RBCD iterations/s", "ms_per_step": 1e3 * dts / SUSTAINED_STEPS,
                     "final_cost_2f": c2s, "final_gradnorm": gns}
    return s, dt, c2, gn, sustained, setup_s


class RankDriver:
    """one process per GPU; agent a is hosted by rank a // ceil(R / world) (consecutive agents share a rank).  Every
    data-path step is a call into the C ABI: dcora_exchange_rbcd_iterate / _tick / _evaluate move the public poses
    between the ranks (neighbour to neighbour) and all-gather the evaluation scalars inside the library."""

    def __init__(self, da, torch, dist, ds, R, r, rank, world, job, acceleration=True):
        self.torch, self.dist, self.rank, self.world, self.R = torch, dist, rank, world, R
        ndev = torch.cuda.device_count()
        self.dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % max(ndev, 1))
        self.staged = dist.get_backend() != "nccl"
        t0 = time.perf_counter()
        self.s = da.RbcdSession(ds, num_robots=R, r=r, acceleration=acceleration, rank=rank, world_size=world,
                                device=self.dev.index)
        self.ex = da.Exchange(self.s, job)
        self.setup_s = time.perf_counter() - t0
        self.transport = self.ex.info()["transport"]
        self.link = self.ex.link_report()  # the start-up link check of dcora_exchange_create
        self._mark = self.ex.info()

    def set_X(self, X):
        self.ex.set_X(X)

    def step(self, selected):
        c2, gn, bn, nxt = self.ex.iterate(selected)
        return c2, gn, nxt

    def tick(self, agents):
        self.ex.tick(agents)

    def evaluate(self):
        c2, gn, bn, nxt = self.ex.evaluate()
        return c2, gn, nxt

    def exchange_stats(self, iterations):
        """host time this rank spent in the exchange since the last call, per iteration"""
        now, was = self.ex.info(), self._mark
        self._mark = now
        it = max(1, iterations)
        return {"transport": now["transport"], "halo_finegrained": now["halo_finegrained"], "wait": now["wait"],
                "ranks_this_rank_stores_to": now["peers"], "link_check": self.link,
                "post_us_per_iteration": 1e6 * (now["post_s"] - was["post_s"]) / it,
                "wait_us_per_iteration": 1e6 * (now["wait_s"] - was["wait_s"]) / it,
                "evaluation_allgather_wait_us_per_iteration": 1e6 * (now["eval_wait_s"] - was["eval_wait_s"]) / it,
                "bytes_posted_per_iteration": (now["bytes_posted"] - was["bytes_posted"]) / it,
                "note": "host wall time of rank 0 inside dcora_exchange_post / _wait / the evaluation all-gather; "
                        "the waits include waiting for the remote rank's kernels (the selected agent's solve)"}

    def close(self):
        self.ex.close()
        self.s.close()

    def timed(self, fn, count):
        torch, dist = self.torch, self.dist
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for _ in range(count):
            out = fn(out)
        self.s.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if self.staged else self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), out


_job_counter = [0]


def make_driver(da, torch, dist, ds, R, r, rank, world, acceleration=True):
    """the library exchange on every rank, or -- when any rank cannot create it -- the collective fallback on all"""
    _job_counter[0] += 1
    token = [uuid.uuid4().hex[:10] if rank == 0 else None]
    dist.broadcast_object_list(token, src=0)
    job = "%s_%d" % (token[0], _job_counter[0])
    drv, err = None, None
    if not os.environ.get("DCORA_BENCH_COLLECTIVES"):
        try:
            drv = RankDriver(da, torch, dist, ds, R, r, rank, world, job, acceleration)
        except Exception as e:  # noqa: BLE001 -- any failure: agree on the fallback below
            err = str(e)
    ok = [None] * world
    dist.all_gather_object(ok, drv is not None)
    if all(ok):
        return drv
    if drv is not None:
        drv.close()
    if rank == 0:
        print("bench: library exchange unavailable (%s): falling back to torch.distributed collectives" % err,
              file=sys.stderr)
    return CollectiveDriver(da, torch, dist, ds, R, r, rank, world, acceleration)


def run_multi(args, da, torch, dist, ds, X0, rank, world):
    drv = make_driver(da, torch, dist, ds, args.robots, args.rank_r, rank, world)
    drv.set_X(X0)
    state = (0.0, 0.0, 0)
    for _ in range(args.warmup + 1):
        state = drv.step(state[2])
    drv.exchange_stats(1)
    dt, state = drv.timed(lambda prev: drv.step((prev or state)[2]), args.steps)
    return drv, dt, state[0], state[1], drv.exchange_stats(args.steps)


class CollectiveDriver:
    """FALLBACK transport, used only when the library's neighbour exchange cannot be created on some rank: the
    exchanges between the phases of an iteration are RCCL collectives over packed public poses (all_gather of the
    non-selected agents' poses, broadcast of the selected agent's, all_reduce of the 2R evaluation scalars)."""
    transport = "torch.distributed collectives (fallback)"

    def __init__(self, da, torch, dist, ds, R, r, rank, world, acceleration=True):
        self.torch, self.dist, self.rank, self.world, self.R = torch, dist, rank, world, R
        ndev = torch.cuda.device_count()
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % max(ndev, 1))
        self.dev = dev
        self.staged = dist.get_backend() != "nccl"  # rehearsal on one GPU: gloo through host staging
        dh = ds.d + 1
        t0 = time.perf_counter()
        # the session enqueues on the stream the collectives are ordered on: pack -> RCCL -> unpack without host syncs
        self.ts = torch.cuda.Stream(device=dev)
        self.s = s = da.RbcdSession(ds, num_robots=R, r=r, acceleration=acceleration, rank=rank, world_size=world,
                                    device=dev.index, stream=self.ts.cuda_stream)
        self.setup_s = time.perf_counter() - t0
        counts = [s.public_count(a) for a in range(R)]
        self.slot = slot = r * dh * max(counts)       # doubles per agent in the exchange buffers
        self.per_rank = per_rank = (R + world - 1) // world  # agent a: rank a // per_rank, slot a % per_rank there
        self.owner = [a // per_rank for a in range(R)]
        self.mine = torch.zeros(per_rank * slot, dtype=torch.float64, device=dev)
        self.everyone = torch.zeros(world * per_rank * slot, dtype=torch.float64, device=dev)
        self.one = torch.zeros(slot, dtype=torch.float64, device=dev)
        self.evalbuf = torch.zeros(2 * R, dtype=torch.float64, device=dev)
        self.esz = self.mine.element_size()
        torch.cuda.synchronize()

    def allgather(self, dst, src):
        if self.staged:
            parts = [self.torch.zeros(src.numel(), dtype=src.dtype) for _ in range(self.world)]
            self.dist.all_gather(parts, src.cpu())
            dst.copy_(self.torch.cat(parts))
            self.torch.cuda.synchronize()
        else:
            with self.torch.cuda.stream(self.ts):
                self.dist.all_gather_into_tensor(dst, src)

    def bcast(self, t, src):
        if self.staged:
            h = t.cpu()
            self.dist.broadcast(h, src=src)
            t.copy_(h)
            self.torch.cuda.synchronize()
        else:
            with self.torch.cuda.stream(self.ts):
                self.dist.broadcast(t, src=src)

    def allreduce(self, t):
        if self.staged:
            h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h)
            self.torch.cuda.synchronize()
        else:
            with self.torch.cuda.stream(self.ts):
                self.dist.all_reduce(t)

    def exchange(self, agents):
        """getSharedStateDicts of `agents` -> ONE all_gather of the packed public poses -> updateNeighborStates on the
        ranks that do not host them (ref examples/MultiRobotExample.cpp:236-258)"""
        s, world, slot, esz = self.s, self.world, self.slot, self.esz
        for a in agents:
            if self.owner[a] == self.rank:
                s.pack_public_dev(a, self.mine.data_ptr() + (a % self.per_rank) * slot * esz)
        if self.staged:
            s.synchronize()
        self.allgather(self.everyone, self.mine)
        for a in agents:
            if self.owner[a] != self.rank:
                s.unpack_public_dev(a, self.everyone.data_ptr() + (self.owner[a] * self.per_rank + a % self.per_rank) * slot * esz)

    def push(self, selected):
        # the new block of the selected agent goes to everyone (their evaluation and their next G need it)
        s = self.s
        if self.owner[selected] == self.rank:
            s.pack_public_dev(selected, self.one.data_ptr())
            if self.staged:
                s.synchronize()
        self.bcast(self.one, self.owner[selected])
        if self.owner[selected] != self.rank:
            s.unpack_public_dev(selected, self.one.data_ptr())

    def evaluate(self):
        s = self.s
        s.phase_evaluate_dev(self.evalbuf.data_ptr())
        if self.staged:
            s.synchronize()
        self.allreduce(self.evalbuf)
        with self.torch.cuda.stream(self.ts):
            h = self.evalbuf.cpu().numpy()
        bn = np.sqrt(h[0::2])
        cost2 = float(h[1::2].sum())  # 2 f = sum_b <X_b, X_b Q_bb + G_b>
        return cost2, float(np.sqrt(h[0::2].sum())), int(np.argmax(bn))

    def step(self, selected):
        """one RBCD++ iteration, the loop body of the reference driver"""
        s = self.s
        s.phase_nonselected(selected)
        self.exchange([a for a in range(self.R) if a != selected])
        s.phase_selected(selected)
        self.push(selected)
        return self.evaluate()

    def set_X(self, X):
        self.s.set_X(X)

    def exchange_stats(self, iterations):
        return {"transport": self.transport}

    def close(self):
        self.s.close()

    def tick(self, agents):
        """the agents of the set update at the same time, each on the GPU of its rank; then their public poses travel"""
        self.s.iterate_set(agents)
        if self.world > 1 or self.staged:
            self.exchange([int(a) for a in agents])

    def timed(self, fn, count):
        torch, dist = self.torch, self.dist
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for _ in range(count):
            out = fn(out)
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if self.staged else self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), out


SKIP_COLOURED = False


def coloured_sweeps(drv, X0, sweeps, warm=2):
    """Block updates per second when the agents of one colour update at the same time (non-accelerated agents, as
    the reference's asynchronous mode; fixed colour order; one evaluation per sweep).  A separate mode, never
    `value`: an iteration here is a tick of several block updates."""
    if SKIP_COLOURED:
        raise RuntimeError("skipped (--no-coloured)")
    s = drv.s
    col, nc = s.colours()
    sets = [np.flatnonzero(col == c).astype(np.int32) for c in range(nc)]
    s.set_acceleration(False)
    drv.set_X(X0)

    def sweep(_prev):
        for S in sets:
            drv.tick(S)
        return drv.evaluate() if drv.dist is not None else s.evaluate()[:2]

    first = None
    for _ in range(warm):
        first = sweep(None)
    drv.set_X(X0)
    first = sweep(None)
    drv.set_X(X0)
    if drv.dist is not None:
        dt, last = drv.timed(sweep, sweeps)
    else:
        s.synchronize()
        t0 = time.perf_counter()
        last = None
        for _ in range(sweeps):
            last = sweep(last)
        s.synchronize()
        dt = time.perf_counter() - t0
    R = len(col)
    return {"mode": "coloured simultaneous updates (agents of one colour at once, non-accelerated, fixed order)",
            "colours": col.tolist(), "sweeps": sweeps, "block_updates": sweeps * R,
            "block_updates_per_s": sweeps * R / dt, "ms_per_sweep": 1e3 * dt / sweeps,
            "cost_2f_after_first_sweep": float(first[0]), "cost_2f_last": float(last[0]),
            "gradnorm_last": float(last[1])}


STRONG_SWEEPS = 10
REPLAYS = 5


def strong_scaling_entry(drv, X0, R, n_gpus):
    """one point of the strong-scaling curve: coloured sweeps over the WHOLE 100k lattice split into R agents (a sweep
    updates every block once; the graph, the start point and the number of sweeps do not depend on n_gpus)"""
    c = coloured_sweeps(drv, X0, sweeps=STRONG_SWEEPS, warm=1)
    return {"agents": R, "n_gpus": n_gpus, "sweeps": STRONG_SWEEPS, "sweeps_per_s": 1e3 / c["ms_per_sweep"],
            "block_updates_per_s": c["block_updates_per_s"], "ms_per_sweep": c["ms_per_sweep"],
            "cost_2f_after_first_sweep": c["cost_2f_after_first_sweep"],
            "cost_2f_after_%d_sweeps" % STRONG_SWEEPS: c["cost_2f_last"], "colours": c["colours"]}


STRONG_NOTE = ("strong scaling of the mode that CAN scale: coloured simultaneous updates (ref src/Agent.cpp:650-678 as "
               "ticks) on the 100k-pose lattice split into R = 16 agents at EVERY N (agent a on rank a // (16 / N), so both "
               "colours sit on every rank); same graph, start point and number of sweeps at every N: speed-up(N) = "
               "sweeps_per_s at N GPUs / sweeps_per_s of the N = 1 line.  Secondary: R = 2 N agents (agents 2g, 2g+1 on "
               "rank g), with the one-GPU figures for R = 4, 8 in the N = 1 line.  Sequential RBCD (`value`) updates one "
               "block per iteration and is not expected to rise with N.")


def strong_scaling_single(da):
    from dcora_amd import synth
    ds = synth.lattice_se3()
    r = 5
    rng = np.random.default_rng(20250310)
    X0 = da.manifold_project(r, ds.d, ds.n, rng.uniform(-1, 1, (r, (ds.d + 1) * ds.n)))
    out = {"note": STRONG_NOTE, "n_gpus": 1, "one_gpu": {}}
    for R in (SCALING_R, 4, 8):
        try:
            t0 = time.perf_counter()
            s = da.RbcdSession(ds, num_robots=R, r=r, acceleration=False)
            st = time.perf_counter() - t0
            e = strong_scaling_entry(SingleDriver(s), X0, R, 1)
            e["setup_s"] = st
            out["one_gpu"]["R=%d" % R] = e
            s.close()
        except Exception as e:  # noqa: BLE001
            out["one_gpu"]["R=%d" % R] = {"error": str(e)}
    return out


def strong_scaling_multi(da, torch, dist, rank, world):
    from dcora_amd import synth
    ds = synth.lattice_se3()
    r = 5
    rng = np.random.default_rng(20250310)
    X0 = da.manifold_project(r, ds.d, ds.n, rng.uniform(-1, 1, (r, (ds.d + 1) * ds.n)))
    out = {"note": STRONG_NOTE, "n_gpus": world}
    for R in dict.fromkeys((SCALING_R, 2 * world)):
        try:
            drv = make_driver(da, torch, dist, ds, R, r, rank, world, acceleration=False)
            e = strong_scaling_entry(drv, X0, R, world)
            e["setup_s"] = drv.setup_s
            e["transport"] = drv.transport
            e["exchange"] = drv.exchange_stats(STRONG_SWEEPS)
            out["R=%d" % R] = e
            drv.close()
        except Exception as e:  # noqa: BLE001
            out["R=%d" % R] = {"error": str(e)}
    return out


class SingleDriver:
    """the same interface on one process without a process group"""
    dist = None

    def __init__(self, s):
        self.s = s

    def set_X(self, X):
        self.s.set_X(X)

    def tick(self, agents):
        self.s.iterate_set(agents)


def agent_block(ds, R, b):
    """measurement arrays of agent b under the reference driver's contiguous partition (local indices, robot ids)"""
    per = ds.n // R
    robot = np.minimum(ds.ids[:, [1, 3]] // per, R - 1)
    start = robot * per
    ids = ds.ids.copy()
    ids[:, 0], ids[:, 2] = robot[:, 0], robot[:, 1]
    ids[:, 1] -= start[:, 0]
    ids[:, 3] -= start[:, 1]
    keep = (robot[:, 0] == b) | (robot[:, 1] == b)
    nb = ds.n - b * per if b == R - 1 else per
    return nb, ids[keep], ds.vals[keep]


def committed_profile(nbytes):
    """numbers read from files under profiles/ (a rocprofv3 kernel trace of `bench.py --headline-only` and the PMC
    passes): NOT measured in this run, reported under their own key with their source"""
    import csv
    out = {"note": "read from committed rocprofv3 summaries, not measured in this run"}
    for tag in ("r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_kernel_stats_headline_loop.csv" % tag)
        if not os.path.exists(path):
            continue
        try:
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    if "k_tcg_run" in row["Name"] or "k_fused_pc" in row["Name"] or "k_fused_precond" in row["Name"]:
                        us = float(row["AverageNs"]) / 1e3
                        out["in_loop"] = {"avg_launch_us_all_launches": us, "launches": int(row["Calls"]),
                                          "min_us": float(row["MinNs"]) / 1e3,
                                          "frac_if_every_launch_moved_the_bytes":
                                              nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                          "source": os.path.relpath(path, ROOT)}
                        break
        except Exception:
            pass
        break
    for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json",
                 "pmc_traffic.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                out["traffic_bytes_per_launch"] = j.get("k_fused_pc_bytes_per_launch",
                                                         j.get("k_fused_precond_bytes_per_launch"))
                out["traffic_bytes_per_launch_tcg_run"] = j.get("k_tcg_run_bytes_per_launch")
                out["traffic_source"] = "profiles/" + name
            except Exception:
                pass
            break
    return out


def qapply_in_loop():
    """the block Q-apply as it runs INSIDE the RBCD loop of the 100k lattice (agent blocks of 12 500 poses, with the
    evaluation's epilogue fused in, plus one whole-graph launch per iteration): from the committed per-iteration
    breakdown of a kernel trace (tools/trace_c5.sh), not measured in this run"""
    for tag in ("r05", "r04", "r03"):
        path = os.path.join(ROOT, "profiles", "%s_c5_loop_breakdown.txt" % tag)
        if not os.path.exists(path):
            continue
        try:
            for line in open(path):
                f = line.split()
                if f and f[0] in ("k_spmm_bsrq", "k_spmm_bsr2"):
                    return {"kernel": "%s in the RBCD loop of the 100k lattice (8 agents)" % f[0],
                            "launches_per_rbcd_iteration": float(f[1]), "avg_launch_us": float(f[2]),
                            "us_per_rbcd_iteration": float(f[3]),
                            "note": "mostly agent blocks of 15 MB: launch-bound, not a bandwidth figure",
                            "source": "profiles/%s_c5_loop_breakdown.txt (read from the committed file)" % tag}
        except Exception:
            pass
    return None


def qapply_entry(P, ms, nbytes, extra=None):
    ach = nbytes / (ms * 1e-3) / 1e9
    qi = P.qapply_info()
    e = {"kernel": "%s (Y = X Q + G)" % qi["kernel"], "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": ach / HBM_PEAK_GBPS, "bytes_per_launch": nbytes, "avg_launch_us": ms * 1e3, "nnz": qi["nnz"],
         "bytes_convention": "SURVEY 8(d) CSR figure 12 nnz + 4 (k+1) + 16 r k (+ 8 r k for G), whatever the stored form",
         "stored_matrix_bytes": qi["stored_matrix_bytes"]}
    if extra:
        e.update(extra)
    return e


def tcg_run_roofline(da, ds, r, robots, warmup, steps, sec8d_precond, nnz_block, kb):
    """The dominant kernel of the timed loop since round 5: k_tcg_run, ONE launch per tCG run of an RTR iteration (z0 = P
    grad, then Hessian product + step + dense preconditioner + projection per iteration, the retraction at the end).
    Measured live: the timed window is replayed with HIP events recorded on the solver's stream around every launch
    (dcora_rbcd_profile_tcg_runs); the tCG iterations inside the launches come from the solver's own statistics.
    Algorithmic bytes per launch, SURVEY 8(d): iterations x bytes_tCG + one more preconditioner application (z0), with
    bytes_tCG = bytes_QX + bytes_precond + 10 r k 8, bytes_QX = 12 nnz + 4 (k + 1) + 16 r k, bytes_precond =
    2 nnz(L) 12 + 2 r k 8 (a sparse-factor solve)."""
    X0 = initial_point(da, ds, r)
    s = da.RbcdSession(ds, num_robots=robots, r=r)
    try:
        s.set_X(X0)
        out = s.run(max_iters=warmup, rgrad_tol=0.0)
        selected = int(out["selected"][-1]) if warmup > 0 else 0
        selected = s.iterate(selected)[3]
        s.profile_tcg_runs(True)
        s.profile_tcg_read()
        tcg = outer = 0
        for _ in range(steps):
            selected = s.iterate(selected)[3]
            res = s.last_result()
            tcg += int(res["inner_iterations"])
            outer += int(res["outer_iterations"])
        prof = s.profile_tcg_read()
        s.profile_tcg_runs(False)
    finally:
        s.close()
    if prof["launches"] < 1:
        return None
    rk8 = 8.0 * r * kb
    bytes_qx = 12.0 * nnz_block + 4.0 * (kb + 1) + 2.0 * rk8
    bytes_tcg = bytes_qx + sec8d_precond + 10.0 * rk8
    it_per = tcg / prof["launches"]
    alg = it_per * bytes_tcg + sec8d_precond + 3.0 * rk8
    us = prof["total_us"] / prof["launches"]
    wgs = (kb // (ds.d + 1) + 1) // 2
    streamed = 8.0 * kb * kb + it_per * (wgs * rk8 + 6.0 * rk8) + 4.0 * rk8
    ach = alg / (us * 1e-6) / 1e9
    return {"kernel": "k_tcg_run (ONE launch per tCG run: z0 = P grad, then per iteration Hessian product, step, dense "
                      "preconditioner and projection, the retraction at the end; %d workgroups, grid-wide steps inside), "
                      "one agent, k=%d" % (wgs, kb),
            "achieved": ach, "frac": ach / HBM_PEAK_GBPS, "avg_launch_us": us, "launches_timed": prof["launches"],
            "tcg_iterations_per_launch": it_per, "rtr_outer_iterations": outer, "rbcd_iterations_replayed": steps,
            "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_tcg_iteration": bytes_tcg,
            "bytes_streamed_per_launch": streamed, "frac_streamed": streamed / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
            "measured": "HIP events on the solver's stream around each of the %d k_tcg_run launches of a replay of the "
                        "timed window (%d RBCD iterations), in this run" % (prof["launches"], steps)}


def roofline(da, ds, r, robots, warmup=30, steps=100):
    """HIP-event timing, live in this run, of the kernels that carry the bytes of the loop, each on its own stream:
    - the dominant kernel of the timed loop: k_fused_pc (step, vector updates, the dense (Q_bb + 0.1 I)^-1 product and
      the projection of one tCG iteration of one agent in one launch);
    - the Q-apply kernel (Y = X Q + G) on the whole sphere2500 graph (k_spmm) and on the synthetic 100k-pose lattice
      of BASELINE.json config 5 (k_spmm_bsr), there both cache-warm (one set re-read back to back) and HBM-cold
      (four distinct (Q, X, Y) sets in turn, 4 x 124 MB > the 256 MiB Infinity Cache)."""
    out = {}
    nb, ids, vals = agent_block(ds, robots, 0)
    Qb = da.build_Q_pgo(ds, n=nb, agent=0, ids=ids, vals=vals)
    kb = (ds.d + 1) * nb
    Pb = da.QuadraticProblem(r, ds.d, nb, Qb, G=np.zeros((r, kb)), reg=0.1)
    Pb.f(np.zeros((r, kb)))
    ms, nbytes = Pb.time_precond(reps=300)
    pinfo = Pb.precond_info()
    ach_streamed = nbytes / (ms * 1e-3) / 1e9
    # the ALGORITHMIC bytes of one tCG-iteration's preconditioner step are those of a sparse-factor solve,
    # 2 nnz(L) 12, plus the 7 passes over r x k vectors this launch also makes (r_old, H delta in; eta, H eta, r, z through)
    sec8d = 2.0 * pinfo["nnzL"] * 12 + 7.0 * r * kb * 8
    ach = sec8d / (ms * 1e-3) / 1e9
    one_launch = os.environ.get("DCORA_SOLVER_BC") != "split" and r <= 7 and r * kb <= 12800
    kname = "k_fused_pc (step length + vector updates + dense preconditioner product + projection + stopping rule, " \
            "one launch)" if one_launch else "k_fused_precond (step length + vector updates + dense preconditioner slices)"
    main = {"bound": "hbm", "kernel": "%s, one agent, k=%d" % (kname, kb),
            "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
            "algorithmic_bytes_per_launch": sec8d,
            "frac_algorithmic": ach / HBM_PEAK_GBPS,
            "frac_streamed": ach_streamed / HBM_PEAK_GBPS,
            "achieved_streamed": ach_streamed,
            "bytes_per_launch": nbytes, "avg_launch_us": ms * 1e3,
            "frac_note": "`achieved` / `frac` use SURVEY 8(d)'s algorithmic bytes (a sparse-factor solve: 2 nnz(L) 12 + "
                         "7 r k 8); `frac_streamed` counts the dense symmetric inverse (8 k^2) the kernel actually "
                         "streams: it trades bytes for launches at k = 2000, where the loop is latency-bound",
            "peak_note": "spec peak of HBM3E; measured_stream_triad is what a = b + s c over 6.4 GB reaches on this box in "
                         "this run, and at this kernel's size the operands sit in the 256 MiB Infinity Cache",
            "measured_stream_triad_GBps": da.stream_triad_gbps(),
            "bytes_counted": "streamed: the dense symmetric inverse (8 k^2) and 7 passes over r x k vectors (r_old, "
                             "H delta in; eta, H eta, r, z through); not counted: every workgroup re-reading r_old and "
                             "H delta from L2 to rebuild the residual",
            "survey_8d": {"formula": "bytes_precond = 2 nnz(L) 12 (a sparse-factor solve) + 7 r k 8",
                          "nnz_L": pinfo["nnzL"], "bytes_precond": sec8d,
                          "dense_bytes_over_8d_bytes": nbytes / sec8d},
            "measured": "HIP events on the solver's stream around 300 back-to-back launches of the kernel in its "
                        "in-loop form, in this run"}
    main["from_committed_profile"] = committed_profile(nbytes)
    run_form = False
    try:
        run_form = Pb.solver_info()["tcg"] == "one launch per run"
    except Exception:
        pass
    if run_form:
        # the loop's dominant kernel is the one-launch tCG run; the two-launch form's PC kernel stays as a side entry
        out["k_fused_pc_launch_form"] = {"kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                         "frac": ach / HBM_PEAK_GBPS, "avg_launch_us": ms * 1e3,
                                         "note": "the launch per tCG iteration the coloured mode and the fall-back use"}
        try:
            tr = tcg_run_roofline(da, ds, r, robots, warmup, steps, 2.0 * pinfo["nnzL"] * 12 + 2.0 * r * kb * 8,
                                  Pb.qapply_info()["nnz"], kb)
        except Exception as e:  # noqa: BLE001
            tr = None
            main["tcg_run_error"] = str(e)
        if tr:
            for key in ("kernel", "achieved", "frac", "avg_launch_us", "algorithmic_bytes_per_launch", "frac_streamed",
                        "measured"):
                main[key] = tr[key]
            main["frac_algorithmic"] = tr["frac"]
            main["bytes_per_launch"] = tr["bytes_streamed_per_launch"]
            main["achieved_streamed"] = tr["bytes_streamed_per_launch"] / (tr["avg_launch_us"] * 1e-6) / 1e9
            main["tcg_run"] = tr
            main["frac_note"] = ("`achieved` / `frac`: SURVEY 8(d)'s algorithmic bytes of the tCG iterations inside a launch "
                                 "(Q-apply + sparse-factor preconditioner + 10 vector passes each) over the launch's HIP-event "
                                 "time; `frac_streamed`: what the kernel moves by design (its rows of the dense inverse once "
                                 "per run, H delta gathered by every workgroup per iteration); the loop is latency-bound at "
                                 "k = 2000")
            main["bytes_counted"] = ("streamed: 8 k^2 once per launch + per iteration (workgroups + 6) r k 8; "
                                     "see tcg_run for the split")
            main["survey_8d"] = {"formula": "iterations x (bytes_QX + bytes_precond + 10 r k 8) + bytes_precond + 3 r k 8",
                                 "nnz_L": pinfo["nnzL"], "tcg_iterations_per_launch": tr["tcg_iterations_per_launch"]}
            main["compact_note"] = ("k_tcg_run: SURVEY 8(d) bytes of the %.1f tCG iterations per launch / live HIP-event time; "
                                    "frac_streamed = bytes moved by design; latency-bound at k=2000"
                                    % tr["tcg_iterations_per_launch"])
    tb = main["from_committed_profile"].get("traffic_bytes_per_launch_tcg_run") if run_form else \
        main["from_committed_profile"].get("traffic_bytes_per_launch")
    if tb:
        main["traffic"] = tb
        main["traffic_note"] = ("HBM-side bytes per launch of this kernel from the committed rocprofv3 PMC passes (%s; "
                                "FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE), not collected in this run; "
                                "%.1f x the algorithmic bytes: the dense inverse is streamed on purpose"
                                % (main["from_committed_profile"].get("traffic_source"),
                                   tb / main["algorithmic_bytes_per_launch"]))
    else:
        main["traffic_note"] = "no committed PMC pass found under profiles/"
    Pb.close()
    try:  # the dense kernel on a block 2.5 x larger (sphere2500 split in two, k = 5000): a longer launch.  The library
        # would take the sparse preconditioner at this size (crossover 2200 since round 4): dense is forced for this entry
        nb2, ids2, vals2 = agent_block(ds, 2, 0)
        Q2 = da.build_Q_pgo(ds, n=nb2, agent=0, ids=ids2, vals=vals2)
        k2 = (ds.d + 1) * nb2
        saved = os.environ.get("DCORA_PRECOND")
        os.environ["DCORA_PRECOND"] = "dense"
        try:
            P2 = da.QuadraticProblem(r, ds.d, nb2, Q2, G=np.zeros((r, k2)), reg=0.1)
        finally:
            if saved is None:
                os.environ.pop("DCORA_PRECOND", None)
            else:
                os.environ["DCORA_PRECOND"] = saved
        P2.f(np.zeros((r, k2)))
        ms2, nbytes2 = P2.time_precond(reps=100)
        kind2 = P2.precond_info()["kind"]
        P2.close()
        ach2 = nbytes2 / (ms2 * 1e-3) / 1e9
        out["precond_dense_k%d" % k2] = {"kernel": "k_fused_precond (%s preconditioner, one agent of 2)" % kind2,
                                         "achieved": ach2, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                         "frac": ach2 / HBM_PEAK_GBPS, "bytes_per_launch": nbytes2,
                                         "avg_launch_us": ms2 * 1e3, "k": k2}
    except Exception as e:
        out["precond_dense_k5000"] = {"error": str(e)}
    Q = da.build_Q_pgo(ds)
    k = (ds.d + 1) * ds.n
    P = da.QuadraticProblem(r, ds.d, ds.n, Q, G=np.zeros((r, k)), reg=-1.0)
    P.f(np.zeros((r, k)))
    ms, nbytes = P.time_qapply(reps=200)
    out["qapply_sphere2500"] = qapply_entry(P, ms, nbytes, {"k": k, "regime": "launch-bound (4 MB working set)"})
    P.close()
    try:
        from dcora_amd import synth
        big = synth.lattice_se3()
        Qg = da.build_Q_pgo(big)
        kg = (big.d + 1) * big.n
        Ps = [da.QuadraticProblem(r, big.d, big.n, Qg, G=np.zeros((r, kg)), reg=-1.0) for _ in range(4)]
        for Pg in Ps:
            Pg.f(np.zeros((r, kg)))
        ms, nbytes = Ps[0].time_qapply(reps=50)
        wl = "synthetic 50x50x40 SE(3) lattice, seed 20250310, r=%d" % r
        out["qapply_lattice100k"] = qapply_entry(Ps[0], ms, nbytes, {
            "k": kg, "workload": wl, "cache_state": "warm: one (Q, X, Y) set of %.0f MB re-read back to back inside the "
                                                    "256 MiB Infinity Cache" % (nbytes / 1e6)})
        msc = da.time_qapply_rotating(Ps, reps=48)
        out["qapply_lattice100k_cold"] = qapply_entry(Ps[0], msc, nbytes, {
            "k": kg, "workload": wl, "cache_state": "cold: 4 distinct (Q, X, Y) sets in turn (%.0f MB between two uses "
                                                    "of a set): every launch streams from HBM" % (4 * nbytes / 1e6)})
        for Pg in Ps:
            Pg.close()
        inl = qapply_in_loop()
        if inl:
            out["qapply_in_loop"] = inl
        # the preconditioner of one agent block of that lattice (k = 50 000): partitioned sparse inverse, one gather
        # kernel per dissection level; bytes = stored inverse factors + tables + the vector in / out of every tile
        nb, ids, vals = agent_block(big, 8, 0)
        Qa = da.build_Q_pgo(big, n=nb, agent=0, ids=ids, vals=vals)
        ka = (big.d + 1) * nb
        Pa = da.QuadraticProblem(r, big.d, nb, Qa, G=np.zeros((r, ka)), reg=0.1)
        Pa.f(np.zeros((r, ka)))
        ms, nbytes = Pa.time_precond(reps=50)
        info = Pa.precond_info()
        Pa.close()
        by_rank = {}
        for rr in (6, 7):  # the staircase ranks: the matrix-pipe kernel's time does not depend on r <= 8
            Pr = da.QuadraticProblem(rr, big.d, nb, Qa, G=np.zeros((rr, ka)), reg=0.1)
            Pr.f(np.zeros((rr, ka)))
            by_rank["r=%d" % rr] = {"avg_application_us": 1e3 * Pr.time_precond(reps=50)[0]}
            Pr.close()
        ach = nbytes / (ms * 1e-3) / 1e9
        s8 = 2.0 * info["nnzL"] * 12 + 2.0 * r * ka * 8
        out["precond_sparse_lattice100k_agent"] = {
            "kernel": "partitioned sparse inverse replay (z = r (Q + 0.1 I)^-1): k_sp_mtile (4-row tiles on "
                      "v_mfma_f64_4x4x4_4b, weights stored once in 4 x 4 micro-blocks), merged-level schedule, %d launches "
                      "(two of them the permutations the solver folds into its own kernels)" % info["launches"],
            "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
            "bytes_per_application": nbytes, "avg_application_us": ms * 1e3, "launches": info["launches"], "k": ka,
            "nnz_L": info["nnzL"], "survey_8d_bytes_precond": s8,
            "nnz_Q_agent": int(Qa.nnz), "survey_8d_bytes_qapply_agent": 12.0 * Qa.nnz + 4.0 * (ka + 1) + 16.0 * r * ka,
            "survey_8d_frac": s8 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "dense_inverse_bytes_avoided": 8.0 * ka * ka,
            "by_rank": by_rank}
    except Exception as e:  # the headline line must not depend on the side measurement
        out["qapply_lattice100k"] = {"error": str(e)}
    return main, out


def certified_run(args, da, torch, ds, with_cpu):
    """second half of BASELINE.json's metric: ms to certified optimum.  Start point = chordal initialisation lifted
    to rank r (the reference driver's InitializationMethod::Chordal, examples/MultiRobotExample.cpp:150-153); clock
    runs from the first RBCD iteration until fastVerification accepts the certificate (examples/...:223-348), file
    parsing and the initialisation excluded.  The same flow is timed on the CPU oracle."""
    r = args.rank_r
    t0 = time.perf_counter()
    T = da.chordal_initialization(ds)
    init_ms = 1e3 * (time.perf_counter() - t0)
    X0 = np.zeros((r, (ds.d + 1) * ds.n))
    X0[:ds.d] = T
    from dcora_amd import driver
    da.precond_cache_clear()  # cold: the agents' inverses are built inside the clock (the timed loop above left them cached)
    torch.cuda.synchronize()
    # the reference driver's loop incl. the staircase (dcora_amd/driver.py); per level: RBCD, certificate, escape
    out = driver.multi_robot_example(ds, X0, num_robots=args.robots, r_min=r, max_iters=1000, rgrad_tol=0.1,
                                     min_eig_tol=1e-3, refine_gap=True)
    lv = out["levels"]
    rbcd_ms = 1e3 * sum(x["rbcd_s"] for x in lv)
    cert_ms = 1e3 * sum(x["certification_s"] + x.get("escape_s", 0.0) for x in lv)  # gap refinement not included
    setup_ms = 1e3 * sum(x["setup_s"] for x in lv)
    res = {"init": "chordal", "init_ms": init_ms, "rbcd_iterations": int(out["total_iters"]),
           "agent_setup_ms": setup_ms, "rbcd_ms": rbcd_ms, "certification_ms": cert_ms,
           "total_ms": setup_ms + rbcd_ms + cert_ms, "total_ms_without_agent_setup": rbcd_ms + cert_ms,
           "clock": "SURVEY 8(d): file parsing and the chordal initialisation excluded; creation of the agents at "
                    "every staircase level (Q blocks, preconditioners built from an empty cache) included",
           "certified": bool(out["certified"]), "final_cost_2f": float(out["cost"][-1]),
           "final_gradnorm": float(out["gradnorm"][-1]), "rank": int(out["rank"]), "staircase_levels": len(lv),
           "certified_suboptimality_gap": {
               "gap_2f": 2.0 * out["suboptimality_gap_f"], "relative": 2.0 * out["suboptimality_gap_f"] /
               float(out["cost"][-1]), "n_eff": lv[-1]["n_eff"], "eta": 1e-3,
               "definition": "2 (f(X) - f*) <= eta n_eff after fastVerification(S, eta) accepted: S + eta I >= 0, "
                             "n_eff = tr(X^T X) with centred translations (dcora_cert_suboptimality_gap; an addition "
                             "of this build, the reference reports the boolean and theta only)",
               "refined": {"lambda_min_S": lv[-1].get("lambda_min_S"),
                           "gap_2f": 2.0 * lv[-1].get("suboptimality_gap_f_refined", float("nan")),
                           "ms": 1e3 * lv[-1].get("gap_refinement_s", 0.0),
                           "definition": "-lambda n_eff with lambda <= lambda_min(S) a lower bound VERIFIED by a "
                                         "Cholesky factorisation of S - lambda I (candidate from Lanczos on the "
                                         "accepted (S + eta I)^-1 minus its Ritz residual; "
                                         "dcora_cert_lambda_min_certified); outside the certification clock"}}}
    if with_cpu:
        from oracle import flows, orc
        dso = flows.oracle_dataset(args.dataset)
        tr = orc.run_rbcd(dso, X0, num_robots=args.robots, r_min=r, max_iters=1000, staircase=1)
        res["cpu_port"] = {"rbcd_iterations": int(tr["total_iters"]), "agent_setup_ms": 1e3 * tr["setup_seconds"],
                           "rbcd_ms": 1e3 * tr["rbcd_seconds"], "certification_ms": 1e3 * tr["cert_seconds"],
                           "total_ms": 1e3 * (tr["setup_seconds"] + tr["rbcd_seconds"] + tr["cert_seconds"]),
                           "certified": bool(tr["certified"] == 1), "final_cost_2f": float(tr["cost"][-1]),
                           "cores": 1}
        res["relative_cost_difference"] = abs(res["final_cost_2f"] - tr["cost"][-1]) / abs(tr["cost"][-1])
    return res


def config5_hbm_roofline(c5, rq, r=5):
    """the RBCD loop of the 100k lattice as a fraction of the HBM roofline (north_star: "iterations/sec and fraction of the
    HBM roofline"): SURVEY 8(d)'s ALGORITHMIC bytes of one RBCD iteration -- what its tCG iterations, RTR evaluations,
    the whole-graph evaluation and the RBCD++ bookkeeping must move, counted from the solver statistics of the timed
    stretch -- over the measured time per iteration.  Not a counter reading: the replay streams about 15 % more than the
    8(d) figure of a sparse-factor solve, the small launches are latency-bound."""
    a = (rq or {}).get("precond_sparse_lattice100k_agent") or {}
    g = (rq or {}).get("qapply_lattice100k") or {}
    w = c5.get("solver_work") or {}
    need = (a.get("survey_8d_bytes_qapply_agent"), a.get("survey_8d_bytes_precond"), a.get("k"), g.get("bytes_per_launch"),
            g.get("k"), w.get("tcg_per_iteration"), w.get("rtr_outer_per_iteration"), c5.get("ms_per_step"))
    if any(x is None for x in need):
        return None
    b_qx, b_pc, ka, b_glob, kg, tcg, outer, ms = need
    rk8 = 8.0 * r * ka
    per_tcg = b_qx + b_pc + 10.0 * rk8              # Hessian product, preconditioner, the vector updates of an iteration
    per_eval = b_qx + 1.0 * rk8                     # an evaluation of the agent's block: Q-apply with the linear term
    per_outer = b_pc + 3.0 * rk8                    # z0 = P grad at the start of a tCG run
    bookkeeping = 6.0 * 8.0 * r * kg                # RBCD++: X, V, Y of the whole graph read and written
    total = tcg * per_tcg + (outer + 1.0) * per_eval + outer * per_outer + b_glob + bookkeeping
    ach = total / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
            "algorithmic_bytes_per_rbcd_iteration": total,
            "parts": {"tcg_iterations": tcg, "bytes_per_tcg_iteration": per_tcg, "rtr_iterations": outer,
                      "bytes_per_evaluation": per_eval, "bytes_per_z0": per_outer, "whole_graph_evaluation": b_glob,
                      "rbcdpp_bookkeeping": bookkeeping},
            "note": "SURVEY 8(d) algorithmic bytes of one RBCD iteration (selected agent: 12 500 poses) / measured time"}


def config5_run(da, with_cpu, iters=60, cpu_iters=4):
    """BASELINE.json config 5 as a side measurement (never `value`): the synthetic 100k-pose SE(3) lattice split
    over 8 agents (12 500 poses, k = 50 000 per agent), r = 5, same RBCD++ loop on ONE GPU.  Each agent's
    preconditioner is the partitioned sparse inverse (sparse_precond.h), Q-apply runs on the block-CSR kernel.
    The CPU oracle repeats the first iterations of the same run: the traces must agree."""
    from dcora_amd import synth
    R, r = 8, 5
    ds = synth.lattice_se3()
    rng = np.random.default_rng(20250310)
    X0 = da.manifold_project(r, ds.d, ds.n, rng.uniform(-1, 1, (r, (ds.d + 1) * ds.n)))
    da.precond_cache_clear()  # cold: the set-up below builds the eight agents' preconditioners
    da.chol_cache_clear()
    t0 = time.perf_counter()
    s = da.RbcdSession(ds, num_robots=R, r=r)
    setup_s = time.perf_counter() - t0
    s.set_X(X0)
    s.run(max_iters=3, rgrad_tol=0.0)
    s.set_X(X0)
    t0 = time.perf_counter()
    out = s.run(max_iters=iters, rgrad_tol=0.0)
    dt = time.perf_counter() - t0
    res = {"workload": "synthetic 50x50x40 SE(3) lattice (100000 poses, %d edges, seed 20250310), 8 agents, r=5" % ds.m,
           "iterations": iters, "value": iters / dt, "unit": "RBCD iterations/s", "ms_per_step": 1e3 * dt / iters,
           "setup_s": setup_s, "cost_2f_first": float(out["cost"][0]), "cost_2f_last": float(out["cost"][-1])}
    def solver_work(sess, X, n):
        """untimed: the same iterations once more with the selected agent's solver statistics read back after each
        (dcora_rbcd_last_result) -- what an RBCD iteration of this stretch of the trajectory holds"""
        sess.set_X(X)
        sel = sess.evaluate()[3]
        outer = inner = 0
        for _ in range(n):
            sel = sess.iterate(sel)[3]
            lr = sess.last_result()
            outer += int(lr["outer_iterations"])
            inner += int(lr["inner_iterations"])
        return {"rtr_outer_per_iteration": outer / n, "tcg_per_iteration": inner / n}

    res["solver_work"] = solver_work(s, X0, iters)
    try:
        res["coloured_rbcd"] = coloured_sweeps(SingleDriver(s), X0, sweeps=8, warm=1)
    except Exception as e:
        res["coloured_rbcd"] = {"error": str(e)}
    Xend = s.get_X()
    s.close()
    # the next levels of the Riemannian staircase (BASELINE: r = 5..7): the same loop at r = 6 and 7 from the r = 5
    # iterate embedded in the higher rank (zero rows appended, as escapeSaddle's lift does before its step); the
    # agents' preconditioners come from the cache (they do not depend on r)
    res["staircase_ranks"] = {"5": {"value": res["value"], "ms_per_step": res["ms_per_step"], "setup_s": setup_s,
                                    **res["solver_work"]},
                              "note": "ranks 6 and 7 start where rank 5 stopped: a later stretch of the trajectory, whose "
                                      "local solves run about twice the tCG iterations (tcg_per_iteration) -- the lower "
                                      "rate is solver work, the kernels cost the same per call"}
    for rr in (6, 7):
        try:
            t0 = time.perf_counter()
            sr = da.RbcdSession(ds, num_robots=R, r=rr)
            st_s = time.perf_counter() - t0
            Xr = np.vstack([Xend, np.zeros((rr - r, Xend.shape[1]))])
            sr.set_X(Xr)
            sr.run(max_iters=3, rgrad_tol=0.0)
            sr.set_X(Xr)
            t0 = time.perf_counter()
            o2 = sr.run(max_iters=30, rgrad_tol=0.0)
            d2 = time.perf_counter() - t0
            work = solver_work(sr, Xr, 30)
            sr.close()
            res["staircase_ranks"][str(rr)] = {"value": 30 / d2, "ms_per_step": 1e3 * d2 / 30, "setup_s": st_s,
                                               "cost_2f_last": float(o2["cost"][-1]), **work}
        except Exception as e:
            res["staircase_ranks"][str(rr)] = {"error": str(e)}
    if with_cpu:
        from oracle import orc
        dso = orc.Dataset(ds.d, ds.n, ds.ids, ds.vals)
        tr = orc.run_rbcd(dso, X0, num_robots=R, r_min=r, max_iters=cpu_iters, staircase=0, rgrad_tol=0.0)
        n = min(cpu_iters, iters)
        res["cpu_port"] = {"iterations": int(tr["total_iters"]), "value": tr["total_iters"] / tr["rbcd_seconds"],
                           "unit": "RBCD iterations/s", "cores": 1,
                           "same_block_sequence": bool(np.array_equal(tr["selected"][:n], out["selected"][:n])),
                           "max_relative_cost_difference":
                               float(np.max(np.abs(tr["cost"][:n] - out["cost"][:n]) / np.abs(tr["cost"][:n])))}
    return res


def config2_run(da, ds, with_cpu, r=5):
    """BASELINE.json config 2 as a side measurement: sphere2500 as ONE problem (k = 10 000, sparse preconditioner),
    QuadraticOptimizer from the chordal start to |rgrad| < 1e-4 (RTR 40 x 100 tCG, what tests/test_configs_gpu.py
    checks against the oracle), then the certificate."""
    Q = da.build_Q_pgo(ds)
    t0 = time.perf_counter()
    P = da.QuadraticProblem(r, ds.d, ds.n, Q)
    setup_s = time.perf_counter() - t0
    T = da.chordal_initialization(ds)
    Xc = np.zeros((r, (ds.d + 1) * ds.n))
    Xc[:ds.d] = T
    prm = dict(RTR_iterations=40, RTR_tCG_iterations=100, gradnorm_tol=1e-4)
    opt = da.QuadraticOptimizer(P, da.ROptParameters(**prm))
    opt.optimize(Xc)  # warm-up
    t0 = time.perf_counter()
    X = opt.optimize(Xc)
    dt = time.perf_counter() - t0
    res = opt.getOptResult()
    t0 = time.perf_counter()
    S = da.dual_certificate(r, ds.d, ds.n, X, Q)
    psd, theta, x, lmin = da.fast_verification(S, 1e-3, block=ds.d + 1)
    cert_s = time.perf_counter() - t0
    out = {"workload": "sphere2500.g2o, one problem of k = 10000, r = 5: RTR 40 x 100 tCG from the chordal start",
           "preconditioner": P.precond_info(), "problem_setup_s": setup_s, "seconds": dt,
           "outer_iterations": int(res["outer_iterations"]), "tcg_iterations": int(res["inner_iterations"]),
           "tcg_iterations_per_s": res["inner_iterations"] / dt, "cost_2f": 2.0 * res["fOpt"],
           "gradnorm": res["gradNormOpt"], "certified": bool(psd), "certification_ms": 1e3 * cert_s}
    P.close()
    if with_cpu:
        from oracle import flows, orc
        dso = flows.oracle_dataset("sphere2500")
        Po = orc.Problem(r, dso.d, dso.n, orc.build_Q_pgo(dso))
        t0 = time.perf_counter()
        Xo, reso = Po.optimize(Xc, **prm)
        dto = time.perf_counter() - t0
        out["cpu_port"] = {"seconds": dto, "outer_iterations": int(reso["outer_iters"]),
                           "tcg_iterations": int(reso["inner_iters"]), "cost_2f": 2.0 * reso["fOpt"], "cores": 1}
        # the two runs need not take the same number of iterations: close to the optimum the trust-region ratio is a
        # difference of costs at the rounding level, and equally accurate preconditioners (host / device inverses) led
        # to 6, 18 and 40 outer iterations to the same optimum -- the per-iteration ratio is the like-for-like figure;
        # the wall-clock ratio is printed under a name that says what it includes
        out["speedup_per_tcg_iteration"] = (dto / max(1, reso["inner_iters"])) / (dt / max(1, res["inner_iterations"]))
        out["wall_clock_ratio_including_fewer_iterations"] = dto / dt
    return out


def psd_test_block(da, ds):
    """the PSD test of the certificate (isSparseSymmetricMatrixPSD, ref src/DCORA_utils.cpp:1737-1747) with the
    numeric factorisation on the device (dcora_cert_is_psd_device): sphere2500 (k = 10 000) and the whole 100k
    lattice (k = 400 000), positive verdicts, i.e. complete factorisations"""
    import scipy.sparse as sp
    from dcora_amd import synth
    out = {}
    for name, d_, blk, host in (("sphere2500", ds, ds.d + 1, True), ("lattice100k", synth.lattice_se3(), 4, False)):
        Q = da.build_Q_pgo(d_).to_scipy()
        A = da.Csr.from_scipy((Q + 1e-3 * sp.identity(Q.shape[0])).tocsr())
        da.chol_cache_clear()
        t0 = time.perf_counter()
        ok, cold = da.is_psd_device(A, blk, info=True)
        cold_ms = 1e3 * (time.perf_counter() - t0)
        t0 = time.perf_counter()
        ok2, warm = da.is_psd_device(A, blk, info=True)
        warm_ms = 1e3 * (time.perf_counter() - t0)
        e = {"k": A.n, "nnz": A.nnz, "positive_definite": bool(ok and ok2), "first_call_ms": cold_ms,
             "symbolic_ms": cold["symbolic_ms"], "repeat_call_ms": warm_ms, "numeric_ms": warm["numeric_ms"],
             "factorisation_gflop": warm["flops"] / 1e9, "achieved_tflops": warm["flops"] / warm["numeric_ms"] / 1e9,
             "front_arena_mb": warm["arena_bytes"] / 1e6, "tree_levels": warm["levels"], "launches": warm["launches"]}
        if host:
            t0 = time.perf_counter()
            okh = da.is_psd(A, blk)
            e["host_factorisation_ms"] = 1e3 * (time.perf_counter() - t0)
            e["host_agrees"] = bool(okh == ok)
        else:
            e["host_factorisation_ms"] = None
            e["host_note"] = "the host factorisation of this matrix did not finish in 7 minutes (DESIGN.md section 8)"
        out[name] = e
    return out


def config5_central(da, r=5):
    """BASELINE config 5 to its certified optimum on ONE GPU with the centralised solver (the single-robot flow of the
    reference, examples/SingleRobotExample.cpp, at k = 400 000): problem creation with the central preconditioner
    (partitioned inverse from the device factorisation), RTR rounds of 50 x 200 tCG from the seeded random start to
    |rgrad| < 1e-2, dual certificate + fastVerification.  The agents' RBCD++ converges sublinearly on this graph
    (config5_lattice100k); this block is what makes its optimum and its certificate known."""
    from dcora_amd import synth
    ds = synth.lattice_se3()
    k = (ds.d + 1) * ds.n
    Q = da.build_Q_pgo(ds)
    t0 = time.perf_counter()
    P = da.QuadraticProblem(r, ds.d, ds.n, Q)
    setup_s = time.perf_counter() - t0
    info = P.precond_info()
    pms, pbytes = P.time_precond(reps=10)
    rng = np.random.default_rng(20250310)
    X = da.manifold_project(r, ds.d, ds.n, rng.uniform(-1, 1, (r, k)))
    solve_s, outer, inner, rounds = 0.0, 0, 0, 0
    res = None
    for rounds in range(1, 41):
        opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=50, RTR_tCG_iterations=200, gradnorm_tol=1e-2))
        t0 = time.perf_counter()
        X = opt.optimize(X)
        solve_s += time.perf_counter() - t0
        res = opt.getOptResult()
        outer += int(res["outer_iterations"])
        inner += int(res["inner_iterations"])
        if res["gradNormOpt"] < 1e-2:
            break
    t0 = time.perf_counter()
    S = da.dual_certificate(r, ds.d, ds.n, X, Q)
    psd, theta, v, lmin = da.fast_verification(S, 1e-3, block=ds.d + 1)
    cert_s = time.perf_counter() - t0
    gap, n_eff = da.suboptimality_gap(r, ds.d, ds.n, X, psd, 1e-3, lmin)
    f_star2 = 2.0 * res["fOpt"]
    # the same from the chordal start (the reference driver's InitializationMethod::Chordal; its two SPD systems are
    # solved on the device, on the host they take minutes at this size) -- SURVEY 8(d) keeps the initialisation outside
    # the clock
    chordal = {}
    Y = None
    try:
        t0 = time.perf_counter()
        T = da.chordal_initialization(ds, device=0)
        init_s = time.perf_counter() - t0
        Xc = np.zeros((r, k))
        Xc[:ds.d] = T
        c_solve, c_outer, c_inner = 0.0, 0, 0
        Y = Xc
        for _ in range(40):
            opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=50, RTR_tCG_iterations=200, gradnorm_tol=1e-2))
            t0 = time.perf_counter()
            Y = opt.optimize(Y)
            c_solve += time.perf_counter() - t0
            rc = opt.getOptResult()
            c_outer += int(rc["outer_iterations"])
            c_inner += int(rc["inner_iterations"])
            if rc["gradNormOpt"] < 1e-2:
                break
        chordal = {"chordal_init_s_on_the_device": init_s, "cost_2f_of_the_start": 2.0 * P.f(Xc), "solve_s": c_solve,
                   "outer_iterations": c_outer, "tcg_iterations": c_inner, "cost_2f": 2.0 * rc["fOpt"],
                   "gradnorm": rc["gradNormOpt"],
                   "seconds_to_certified_optimum": setup_s + c_solve + cert_s,
                   "note": "same optimum; clock = problem creation + solve + certificate of the run above (8(d): the "
                           "initialisation is outside)"}
        # the agents' loop from the same start, against the optimum that is now known
        s8 = da.RbcdSession(ds, num_robots=8, r=r)
        s8.set_X(Xc)
        t0 = time.perf_counter()
        o8 = s8.run(max_iters=1000, rgrad_tol=0.1)
        d8 = time.perf_counter() - t0
        s8.close()
        c8 = np.asarray(o8["cost"])
        chordal["rbcd_8_agents"] = {"iterations": int(o8["iters"]), "seconds": d8, "cost_2f_last": float(c8[-1]),
                                    "gradnorm_last": float(o8["gradnorm"][-1]),
                                    "excess_over_optimum": {str(i): float(c8[min(i, len(c8)) - 1] / f_star2 - 1.0)
                                                            for i in (100, 300, 1000)}}
        chordal["distributed_loop_to_the_drivers_stopping_rule"] = distributed_to_tolerance(da, ds, Xc, r, f_star2)
    except Exception as e:
        chordal = {"error": str(e)}
        Y = None
    # recovered poses: rounding to SE(3)^n in the frame of pose 0 (ref src/Agent.cpp:1006-1056)
    t0 = time.perf_counter()
    Xr = Y if Y is not None else X  # (the solution from the chordal start has rank d exactly)
    Tr = da.align_lifted_trajectory_to_frame(Xr, Xr[:, :ds.d + 1], ds.d, ds.n, True)
    round_s = time.perf_counter() - t0
    Asp = Q.to_scipy()
    sv = np.linalg.svd(Xr, compute_uv=False)
    rounding = {"seconds": round_s, "cost_2f_of_the_rounded_trajectory": float(np.sum((Asp @ Tr.T).T * Tr)),
                "singular_values_of_X": [float(x) for x in sv],
                "of": "the solution from the chordal start" if Y is not None else "the solution from the random start",
                "note": "the certified solution has rank d: the relaxation is tight and rounding loses nothing"}
    P.close()
    return {"workload": "synthetic 50x50x40 SE(3) lattice as ONE problem (k = 400000), r = 5, RTR rounds of 50 x 200 tCG "
                        "from the seeded random start to |rgrad| < 1e-2",
            "problem_setup_s": setup_s, "preconditioner": {"kind": info["kind"], "launches": info["launches"],
                                                           "nnz_L": info["nnzL"], "application_us": 1e3 * pms,
                                                           "bytes_per_application": pbytes},
            "solve_s": solve_s, "rtr_rounds": rounds, "outer_iterations": outer, "tcg_iterations": inner,
            "tcg_iterations_per_s": inner / solve_s, "cost_2f": 2.0 * res["fOpt"], "gradnorm": res["gradNormOpt"],
            "certification_s": cert_s, "certified": bool(psd), "rank": r,
            "seconds_to_certified_optimum": setup_s + solve_s + cert_s,
            "certified_suboptimality_gap_2f": 2.0 * gap, "n_eff": n_eff, "recovered_poses": rounding,
            "from_the_chordal_start": chordal,
            "cpu_port": None, "cpu_note": "no CPU leg: the oracle's sparse Cholesky of this matrix does not finish in "
                                          "minutes (DESIGN.md section 8)"}


DIST_BUDGET_S = 25.0


def distributed_to_tolerance(da, ds, X0, r, f_star2, R=8, budget_s=DIST_BUDGET_S):
    """What the DISTRIBUTED loop needs on config 5 to reach the reference driver's stopping rule |rgrad| < 0.1
    (examples/MultiRobotExample.cpp:284) from the chordal start, or where it stands when the budget runs out: RBCD++ (greedy
    selection, acceleration with restarts) and coloured simultaneous ticks, 8 agents on one GPU.  Lets "ms to certified
    optimum" of config 5 be read without the centralised shortcut: the certificate itself adds
    config5_central_certified.certification_s."""
    out = {"stopping_rule": "|rgrad| < 0.1 (ref examples/MultiRobotExample.cpp:284)", "budget_s_per_mode": budget_s,
           "agents": R}
    s = da.RbcdSession(ds, num_robots=R, r=r)
    s.set_X(X0)
    t0 = time.perf_counter()
    iters, gn, c2 = 0, None, None
    while time.perf_counter() - t0 < budget_s:
        o = s.run(max_iters=500, rgrad_tol=0.1)  # continues from the session's state
        iters += int(o["iters"])
        gn, c2 = float(o["gradnorm"][-1]), float(o["cost"][-1])
        if gn < 0.1:
            break
    dt = time.perf_counter() - t0
    out["rbcd_pp"] = {"reached": bool(gn is not None and gn < 0.1), "iterations": iters, "seconds": dt, "gradnorm": gn,
                      "cost_2f": c2, "excess_over_optimum": None if c2 is None else c2 / f_star2 - 1.0}
    s.close()
    if SKIP_COLOURED:
        out["coloured_ticks"] = {"skipped": "--no-coloured"}
        return out
    s = da.RbcdSession(ds, num_robots=R, r=r, acceleration=False)
    s.set_X(X0)
    col, nc = s.colours()
    sets = [np.flatnonzero(col == c).astype(np.int32) for c in range(nc)]
    t0 = time.perf_counter()
    sweeps, gn, c2 = 0, None, None
    while time.perf_counter() - t0 < budget_s:
        for S in sets:
            s.iterate_set(S)
        sweeps += 1
        if sweeps % 5 == 0:
            c2, gn = [float(x) for x in s.evaluate()[:2]]
            if gn < 0.1:
                break
    c2, gn = [float(x) for x in s.evaluate()[:2]]
    dt = time.perf_counter() - t0
    out["coloured_ticks"] = {"reached": bool(gn < 0.1), "sweeps": sweeps, "block_updates": sweeps * R, "seconds": dt,
                             "gradnorm": gn, "cost_2f": c2, "excess_over_optimum": c2 / f_star2 - 1.0}
    s.close()
    return out


def side_multi(da, torch, dist, rank, world, ds, R, r, workload, iters=60, sweeps=8, more_ranks=()):
    """a BASELINE.json multi-agent config with one process per GPU (consecutive agents share a rank): same loop, the
    library's neighbour exchange between the ranks; a side measurement, never `value`"""
    rng = np.random.default_rng(20250310)
    X0 = da.manifold_project(r, ds.d, ds.n, rng.uniform(-1, 1, (r, (ds.d + 1) * ds.n)))
    drv = make_driver(da, torch, dist, ds, R, r, rank, world)
    drv.set_X(X0)
    state = (0.0, 0.0, 0)
    for _ in range(3):
        state = drv.step(state[2])
    drv.set_X(X0)
    drv.exchange_stats(1)
    first = (0.0, 0.0, 0)
    costs = []

    def one(prev):
        out = drv.step((prev or first)[2])
        costs.append(out[0])
        return out

    dt, state = drv.timed(one, iters)
    res = {"workload": workload, "n_gpus": world,
           "process_group": {"backend": dist.get_backend(), "ranks": dist.get_world_size()},
           "parallelism": "%d agents in consecutive groups over %d rank(s)" % (R, world),
           "iterations": iters, "value": iters / dt, "unit": "RBCD iterations/s", "ms_per_step": 1e3 * dt / iters,
           "setup_s": drv.setup_s, "cost_2f_first": float(costs[0]), "cost_2f_last": float(costs[-1]),
           "exchange": drv.exchange_stats(iters)}
    Xend = drv.ex.gather_X() if hasattr(drv, "ex") else None
    try:
        res["coloured_rbcd"] = coloured_sweeps(drv, X0, sweeps=sweeps, warm=1)
    except Exception as e:
        res["coloured_rbcd"] = {"error": str(e)}
    drv.close()
    if more_ranks and Xend is not None:  # the next staircase levels, from the r iterate embedded in the higher rank
        res["staircase_ranks"] = {str(r): {"value": res["value"], "ms_per_step": res["ms_per_step"]}}
        for rr in more_ranks:
            try:
                d2 = make_driver(da, torch, dist, ds, R, rr, rank, world)
                Xr = np.vstack([Xend, np.zeros((rr - r, Xend.shape[1]))])
                d2.set_X(Xr)
                st2 = (0.0, 0.0, 0)
                for _ in range(3):
                    st2 = d2.step(st2[2])
                d2.set_X(Xr)
                first2 = (0.0, 0.0, 0)
                dt2, _o = d2.timed(lambda prev: d2.step((prev or first2)[2]), 30)
                res["staircase_ranks"][str(rr)] = {"value": 30 / dt2, "ms_per_step": 1e3 * dt2 / 30, "setup_s": d2.setup_s}
                d2.close()
            except Exception as e:
                res["staircase_ranks"][str(rr)] = {"error": str(e)}
    return res


def config3_single(da, with_cpu, iters=200, cpu_iters=60):
    """BASELINE.json config 3 (torus3D.g2o, 8 agents) with all agents on ONE GPU: the same RBCD++ loop as the headline
    on the other split BASELINE names (k = 2500 per agent: the partitioned sparse preconditioner); a side measurement,
    never `value`.  With N > 1 the same key holds the run with the agents spread over the ranks."""
    from dcora_amd import datasets
    ds = datasets.product_dataset("torus3D")
    R, r = 8, 5
    rng = np.random.default_rng(20250310)
    X0 = da.manifold_project(r, ds.d, ds.n, rng.uniform(-1, 1, (r, (ds.d + 1) * ds.n)))
    t0 = time.perf_counter()
    s = da.RbcdSession(ds, num_robots=R, r=r)
    setup_s = time.perf_counter() - t0
    s.set_X(X0)
    s.run(max_iters=5, rgrad_tol=0.0)
    s.set_X(X0)
    t0 = time.perf_counter()
    out = s.run(max_iters=iters, rgrad_tol=0.0)
    dt = time.perf_counter() - t0
    s.close()
    res = {"workload": "torus3D.g2o, 8 agents, r=5, RBCD++ (accel, restart 30), RTR 3x50 tCG; all agents on one GPU",
           "n_gpus": 1, "iterations": iters, "value": iters / dt, "unit": "RBCD iterations/s",
           "ms_per_step": 1e3 * dt / iters, "setup_s": setup_s, "cost_2f_first": float(out["cost"][0]),
           "cost_2f_last": float(out["cost"][-1])}
    if with_cpu:
        from oracle import orc
        dso = orc.Dataset(ds.d, ds.n, ds.ids, ds.vals)
        tr = orc.run_rbcd(dso, X0, num_robots=R, r_min=r, max_iters=cpu_iters, staircase=0, rgrad_tol=0.0)
        n = min(cpu_iters, iters)
        res["cpu_port"] = {"iterations": int(tr["total_iters"]), "value": tr["total_iters"] / tr["rbcd_seconds"],
                           "unit": "RBCD iterations/s", "cores": 1,
                           "same_block_sequence": bool(np.array_equal(tr["selected"][:n], out["selected"][:n])),
                           "max_relative_cost_difference":
                               float(np.max(np.abs(tr["cost"][:n] - out["cost"][:n]) / np.abs(tr["cost"][:n])))}
    return res


def config5_multi(da, torch, dist, rank, world):
    from dcora_amd import synth
    ds = synth.lattice_se3()
    return side_multi(da, torch, dist, rank, world, ds, 8, 5,
                      "synthetic 50x50x40 SE(3) lattice (100000 poses, %d edges, seed 20250310), 8 agents, r=5" % ds.m,
                      iters=60, sweeps=8, more_ranks=(6, 7))


def config3_multi(da, torch, dist, rank, world):
    from dcora_amd import datasets
    ds = datasets.product_dataset("torus3D")
    return side_multi(da, torch, dist, rank, world, ds, 8, 5,
                      "torus3D.g2o, 8 agents, r=5, RBCD++ (accel, restart 30), RTR 3x50 tCG", iters=200, sweeps=20)


def config4_multi_robot(da, ra, with_cpu, r=3, iters=40, cpu_iters=3):
    """tiers.pyfg as the multi-robot driver sees it (examples/MultiRobotExample_RASLAM.cpp): 4 robots, every agent
    resident on the GPU (dcora_ra_rbcd_*), RBCD++ from the lifted odometry start with the agents' default local
    solver (RTR 3 x 50); the CPU figure is the same loop over the oracle's local solver for the first iterations"""
    X0 = np.zeros((r, ra.k))
    X0[:ra.d] = ra.X_odom
    t0 = time.perf_counter()
    s = da.RaRbcdSession(ra, r)
    setup_s = time.perf_counter() - t0
    s.set_X(X0)
    s.run(max_iters=2, rgrad_tol=0.0)
    s.set_X(X0)
    t0 = time.perf_counter()
    out = s.run(max_iters=iters, rgrad_tol=0.0)
    dt = time.perf_counter() - t0
    s.close()
    res = {"workload": "tiers.pyfg, %d robots, r=%d, RBCD++ (accel, restart 30), RTR 3x50 tCG" % (len(ra.robots), r),
           "iterations": iters, "value": iters / dt, "unit": "RBCD iterations/s", "ms_per_step": 1e3 * dt / iters,
           "setup_s": setup_s, "cost_2f_first": float(out["cost"][0]), "cost_2f_last": float(out["cost"][-1])}
    if with_cpu:
        from oracle import flows, orc
        opt = dict(RTR_iterations=3, RTR_tCG_iterations=50, gradnorm_tol=1e-2)
        flows.oracle_ra_rbcd_loop(da, orc, ra, X0, r, 1, True, 30, opt)  # builds / warms what the loop reuses
        t0 = time.perf_counter()
        Xo, tr = flows.oracle_ra_rbcd_loop(da, orc, ra, X0, r, cpu_iters, True, 30, opt)
        dtc = time.perf_counter() - t0
        res["cpu_port"] = {"iterations": cpu_iters, "value": cpu_iters / dtc, "unit": "RBCD iterations/s", "cores": 1,
                           "note": "numpy loop over the oracle's local solver, set-up of the agents included",
                           "same_block_sequence": bool(np.array_equal(tr[:, 0].astype(int),
                                                                      out["selected"][:cpu_iters]))}
    return res


def config4_run(da, with_cpu, cpu_staircase=False):
    """BASELINE.json config 4 as a side measurement (never `value`): tiers.pyfg (d = 2, 9768 poses, 7789 ranges, one
    landmark: k = 37 094), the first level of the centralised CORA flow -- RTR with the driver's parameters
    (200 x 200 tCG, tol 1e-4, ref examples/SingleRobotExample_RASLAM.cpp:59-79) at rank d from the odometry start.
    The range-aided layout runs the unfused solver path with the partitioned sparse preconditioner (the landmark
    is a hub: Schur complement).  Both sides keep the reference's 5 s TimeBound of one RTR run
    (ref src/QuadraticOptimizer.cpp:252): the CPU oracle stops on it, the GPU finishes its 200 outer iterations."""
    from dcora_amd import cora_flow, datasets
    path = os.path.join(datasets.DATA, "tiers.pyfg.gz")
    ra = da.RADataset(path)
    hip = cora_flow.ProductBackend(ra)
    t0 = time.perf_counter()
    P = hip.problem(ra.d)
    setup_s = time.perf_counter() - t0
    info = P.precond_info()
    t0 = time.perf_counter()
    X, f, gn, outer, inner = hip.optimize(P, ra.X_odom)
    dt = time.perf_counter() - t0
    P.close()
    res = {"workload": "tiers.pyfg, centralised CORA level r = d = 2: RTR 200 x 200, tol 1e-4, odometry start",
           "k": ra.k, "nnz_Q": ra.Q.nnz, "f": f, "gradnorm": gn, "outer_iterations": outer, "tcg_iterations": inner,
           "seconds": dt, "tcg_iterations_per_s": inner / dt, "problem_setup_s": setup_s,
           "preconditioner": {"kind": info["kind"], "launches": info["launches"], "nnz_L": info["nnzL"]}}
    if with_cpu:
        import gzip
        import shutil
        import tempfile
        from oracle import orc
        fd, tmp = tempfile.mkstemp(suffix=".pyfg")
        with os.fdopen(fd, "wb") as out, gzip.open(path, "rb") as src:
            shutil.copyfileobj(src, out)
        ro = orc.RADataset(tmp)
        os.unlink(tmp)
        from oracle import flows
        cpu = flows.OracleBackend(ro, hip.reg)
        Po = cpu.problem(ro.d)
        t0 = time.perf_counter()
        Xo, fo, gno, oo, io = cpu.optimize(Po, ro.X_odom)
        dtc = time.perf_counter() - t0
        res["cpu_port"] = {"f": fo, "gradnorm": gno, "outer_iterations": oo, "tcg_iterations": io, "seconds": dtc,
                           "tcg_iterations_per_s": io / dtc, "cores": 1,
                           "note": "stopped by the reference's 5 s TimeBound of one RTR run"}
        # like for like: the GPU run cut at the number of outer iterations the CPU completed
        P = hip.problem(ra.d)
        opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=int(oo), RTR_tCG_iterations=200,
                                                         gradnorm_tol=1e-4))
        t0 = time.perf_counter()
        opt.optimize(ra.X_odom)
        dts = time.perf_counter() - t0
        rs = opt.getOptResult()
        P.close()
        res["same_outer_iterations_as_cpu"] = {"f": rs["fOpt"], "gradnorm": rs["gradNormOpt"],
                                               "outer_iterations": rs["outer_iterations"],
                                               "tcg_iterations": rs["inner_iterations"], "seconds": dts,
                                               "speedup_vs_cpu_port": dtc / dts}
    try:
        res["multi_robot"] = config4_multi_robot(da, ra, with_cpu)
    except Exception as e:
        res["multi_robot"] = {"error": str(e)}
    # the second half of the metric on this configuration: the whole staircase of the reference's driver
    # (examples/SingleRobotExample_RASLAM.cpp:188-283) from the odometry start to the certified, rounded solution
    try:
        out = cora_flow.cora(hip, ra.X_odom, ra.d)
        lv = out["levels"]
        res["ms_to_certified_optimum"] = {
            "value": out["ms_total"], "unit": "ms", "certified": bool(out["certified"]), "certified_at_rank": out["r_final"],
            "ranks_visited": [a["r"] for a in lv], "f_per_rank": [a["f"] for a in lv],
            "theta_per_rank": [a["theta"] for a in lv], "tcg_iterations": int(sum(a["inner"] for a in lv)),
            "f_certified": lv[-1]["f"], "f_rounded": out["f_rounded"],
            "relative_gap_rounded_vs_certified": (out["f_rounded"] - lv[-1]["f"]) / abs(lv[-1]["f"]),
            "clock": "the driver's loop: every level creates its problem (preconditioner from the library's cache after "
                     "the first), RTR 200 x 200, dual certificate + fastVerification, escapeSaddle; then "
                     "projectSolutionRASLAM and the refinement at rank d"}
        if cpu_staircase and with_cpu:
            from oracle import flows as _fl
            t0 = time.perf_counter()
            ref = cora_flow.cora(_fl.OracleBackend(ro, hip.reg), ro.X_odom, ro.d)
            res["ms_to_certified_optimum"]["cpu_port"] = {
                "value": 1e3 * (time.perf_counter() - t0), "unit": "ms", "cores": 1, "certified": bool(ref["certified"]),
                "certified_at_rank": ref["r_final"], "f_certified": ref["levels"][-1]["f"], "f_rounded": ref["f_rounded"],
                "note": "every RTR run of the port stops on the reference's 5 s TimeBound"}
        else:
            res["ms_to_certified_optimum"]["cpu_port"] = (
                "not run by default (minutes: each of its RTR runs stops on the reference's 5 s TimeBound, its "
                "certificates take tens of seconds each); --cpu-c4-staircase runs it, DESIGN.md section 5 has the figure")
    except Exception as e:
        res["ms_to_certified_optimum"] = {"error": str(e)}
    return res


def cpu_baseline(args, ds_name, X0):
    """the CPU oracle (1 thread) on the same trajectory from the same start point; rates are taken over the SAME
    iteration windows as the GPU figures (the oracle stamps its loop clock after every iteration): the driver's window
    (iterations warmup+2 .. warmup+1+steps) and the sustained window that follows it"""
    from oracle import flows, orc
    dso = flows.oracle_dataset(ds_name)
    first = args.warmup + 1
    need = first + args.steps + SUSTAINED_STEPS
    n_it = max(need, args.cpu_steps or 1000)  # about 10 ms / iteration on one core: a 10-30 s sample
    tr = orc.run_rbcd(dso, X0, num_robots=args.robots, r_min=args.rank_r, max_iters=n_it, staircase=0,
                      rgrad_tol=0.0)
    t = tr["seconds"]
    same = args.steps / (t[first + args.steps - 1] - t[first - 1])
    lo, hi = first + args.steps, first + args.steps + SUSTAINED_STEPS
    sus = SUSTAINED_STEPS / (t[hi - 1] - t[lo - 1])
    return {"value": sus, "unit": "RBCD iterations/s", "cores": 1, "host_cores_of_the_box": os.cpu_count(), "kind": "port",
            "sample": "iterations %d..%d of the same workload and start point (the window of `sustained`); the oracle "
                      "ran %d iterations, %.1f s of loop time" % (lo + 1, hi, tr["total_iters"], t[-1]),
            "ms_per_step": 1e3 / sus,
            "same_window_as_value": {"iterations": "%d..%d" % (first + 1, first + args.steps), "value": same,
                                     "ms_per_step": 1e3 / same},
            "whole_sample": {"iterations": int(tr["total_iters"]), "value": tr["total_iters"] / t[-1]},
            "final_cost_2f_of_sample": float(tr["cost"][-1]),
            "r_threads": cpu_r_threads(args, dso, X0, orc, tr)}


def cpu_r_threads(args, dso, X0, orc, tr1):
    """the variant with one host thread per agent (ref src/Agent.cpp:660-662 starts one per agent in
    its asynchronous mode).  (i) the SAME synchronous loop with the non-selected agents' updates of a round on R
    threads -- one agent solves at a time, so threads buy next to nothing there; (ii) the mode in which they do:
    agents of one colour updating at the same time, block updates/s with 1 and with R threads (the CPU side of
    `coloured_rbcd`).  Bounded samples of the same workload and start point."""
    R = args.robots
    n_it = 300
    trR = orc.run_rbcd(dso, X0, num_robots=R, r_min=args.rank_r, max_iters=n_it, staircase=0, rgrad_tol=0.0, threads=R)
    same_trace = bool(np.array_equal(trR["selected"], tr1["selected"][:n_it]) and
                      np.allclose(trR["cost"], tr1["cost"][:n_it], rtol=1e-12, atol=0))
    t1 = tr1["seconds"]
    out = {"cores": R, "host_cores_of_the_box": os.cpu_count(),
           "sequential_loop": {"value": n_it / trR["seconds"][-1], "unit": "RBCD iterations/s",
                               "one_thread_over_the_same_iterations": n_it / t1[n_it - 1],
                               "same_trace_as_one_thread": same_trace,
                               "sample": "iterations 1..%d of the same workload" % n_it}}
    sweeps = 20
    c1 = orc.run_coloured(dso, X0, num_robots=R, r=args.rank_r, sweeps=sweeps, threads=1)
    cR = orc.run_coloured(dso, X0, num_robots=R, r=args.rank_r, sweeps=sweeps, threads=R)
    out["coloured_updates"] = {"unit": "block updates/s", "sweeps": sweeps, "colours": cR["colours"],
                               "one_thread": sweeps * R / c1["loop_seconds"],
                               "r_threads": sweeps * R / cR["loop_seconds"],
                               "same_costs": bool(np.array_equal(c1["cost"], cR["cost"])),
                               "cost_2f_last": float(cR["cost"][-1])}
    return out


def main():
    args = parse()
    global SKIP_COLOURED
    SKIP_COLOURED = bool(args.no_coloured)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no launcher: start one rank per GPU as a child process, before anything here has touched the GPU
        raise SystemExit(spawn_ranks(args))
    if env_world is not None and int(env_world) != args.gpus and not os.environ.get("DCORA_FORCE_MULTI"):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, env_world))
    # stdout carries exactly one JSON line: native libraries (RCCL prints a version banner) write to fd 1 too, so
    # point fd 1 at stderr for the run and keep the real stdout for the result
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    world = int(env_world or 1)
    import torch
    import dcora_amd as da
    from dcora_amd import datasets
    if da.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libdcora_hip has no CPU fallback")
    local = int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    ds = datasets.product_dataset(args.dataset)
    X0 = initial_point(da, ds, args.rank_r)
    multi = world > 1 or bool(os.environ.get("DCORA_FORCE_MULTI"))  # the latter: 1-rank rehearsal of the N>1 path
    sustained = exch = c3 = c5 = group = strong = None
    if multi:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("DCORA_DIST_BACKEND", "nccl"))
        group = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                 "used_for": "barrier + MAX of the timed region only"}
        drv, dt, c2, gn, exch = run_multi(args, da, torch, dist, ds, X0, rank, world)
        group["transport"] = drv.transport
        coloured = None if (args.headline_only or args.scaling_only) else coloured_sweeps(drv, X0, sweeps=40)
        drv.close()
        if not args.headline_only:
            c3 = None if (args.no_config3 or args.scaling_only) else config3_multi(da, torch, dist, rank, world)
            c5 = None if (args.no_config5 or args.scaling_only) else config5_multi(da, torch, dist, rank, world)
            strong = None if args.no_config5 else strong_scaling_multi(da, torch, dist, rank, world)
        dist.barrier()
        dist.destroy_process_group()
    else:
        s, dt, c2, gn, sustained, setup_s = run_single(args, da, torch, ds, X0)
        coloured = None if (args.headline_only or args.scaling_only or SKIP_COLOURED) else \
            coloured_sweeps(SingleDriver(s), X0, sweeps=40)
    if rank != 0:
        return
    ms = 1e3 * dt / args.steps
    line = {
        "metric": "RBCD iterations/sec, sphere2500 5-agent split",
        "value": args.steps / dt,
        "unit": "RBCD iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "%s.g2o (public dataset shipped as a fixture), seeded random start point" % args.dataset,
        "config": {"workload": "%s.g2o, %d agents, r=%d, RBCD++ (accel, restart 30), RTR 3x50 tCG" %
                               (args.dataset, args.robots, args.rank_r),
                   "parallelism": "agents in consecutive groups over %d rank(s)" % world,
                   "timed_iterations": "%d..%d of the trajectory from the start point" %
                                       (args.warmup + 2, args.warmup + 1 + args.steps),
                   "window_note": "iterations early in the trajectory are cheaper on BOTH sides (tCG exits early): a short "
                                  "window such as the driver's --steps 20 --warmup 5 reads ~1.6 x the sustained rate; "
                                  "`sustained` continues the same trajectory for %d more iterations and "
                                  "cpu_baseline.same_window_as_value is the CPU port over the same window" % SUSTAINED_STEPS,
                   "final_cost_2f": c2, "final_gradnorm": gn},
    }
    if getattr(run_single, "samples", None):
        sm = [args.steps / t for t in run_single.samples]
        line["value_samples"] = {"replays": len(sm), "statistic": "median (each replay: set X0, the same warm-up, the "
                                 "same K timed iterations)", "min": min(sm), "max": max(sm), "all": sm}
    if group is not None:
        line["process_group"] = group
        line["exchange"] = exch
    if args.headline_only:
        emit(real_stdout, line)
        return
    if sustained is not None:
        line["sustained"] = sustained
        line["config"]["session_setup_s"] = setup_s
    line["coloured_rbcd"] = coloured
    if c3 is not None:
        line["config3_torus3D_8agents"] = c3
    if c5 is not None:
        line["config5_lattice100k"] = c5
    if strong is not None:
        line["strong_scaling"] = strong
    if multi:
        cached, cache_note = cache_load(args)
        line["scaling_100k_lattice"] = scaling_block(world, strong, c5, cached, cache_note)
        if args.scaling_only:
            emit(real_stdout, line)
            return
        if not args.no_cpu_baseline:
            # the CPU port is timed ONCE (rank 0 of the N = 1 run); the N > 1 lines quote that figure when the cache
            # was written by this code on this workload on this box (provenance printed) -- or time it here, on rank 0
            # only and after the timed region
            if cached and cached.get("cpu_baseline"):
                cb = dict(cached["cpu_baseline"])
                cb["source"] = "N = 1 run of this bench on this box, %.0f s earlier (head %s)" % (
                    cached["provenance"]["age_s"], (cached["provenance"].get("git_head") or "?")[:8])
                cb["provenance"] = cached["provenance"]
            else:
                cb = cpu_baseline(args, args.dataset, X0)
                cb["source"] = "timed in this run on rank 0 after the timed region (%s)" % cache_note
            line["cpu_baseline"] = cb
            try:
                line["config"]["speedup_vs_cpu_port"] = line["value"] / cb["same_window_as_value"]["value"]
            except Exception:
                pass
    if args.scaling_only:
        line["strong_scaling"] = strong_scaling_single(da)
        line["scaling_100k_lattice"] = scaling_block(1, line["strong_scaling"], None, None)
        emit(real_stdout, line)
        return
    line["roofline"], line["roofline_qapply"] = roofline(da, ds, args.rank_r, args.robots, args.warmup,
                                                         min(args.steps, 100))
    if world == 1 and not multi:
        try:
            line["ms_to_certified_optimum"] = certified_run(args, da, torch, ds, not args.no_cpu_baseline)
        except Exception as e:  # never lose the headline line to the second measurement
            line["ms_to_certified_optimum"] = {"error": str(e)}
        if not args.no_config4:
            try:
                line["config2_sphere2500_single"] = config2_run(da, ds, not args.no_cpu_baseline)
            except Exception as e:
                line["config2_sphere2500_single"] = {"error": str(e)}
        if not args.no_config5:
            try:
                line["psd_test"] = psd_test_block(da, ds)
            except Exception as e:
                line["psd_test"] = {"error": str(e)}
        if not args.no_config3:
            try:
                line["config3_torus3D_8agents"] = config3_single(da, not args.no_cpu_baseline)
            except Exception as e:
                line["config3_torus3D_8agents"] = {"error": str(e)}
        if not args.no_config5:
            # (the throughput windows first: the centralised solve and the 2 x 25 s of the distributed loop behind them
            # leave the GPU at a lower clock -- the same 60-iteration window read 877 instead of 1160 it/s after them)
            try:
                line["config5_lattice100k"] = config5_run(da, not args.no_cpu_baseline)
                hr = config5_hbm_roofline(line["config5_lattice100k"], line.get("roofline_qapply"))
                if hr:
                    line["config5_lattice100k"]["hbm_roofline"] = hr
            except Exception as e:
                line["config5_lattice100k"] = {"error": str(e)}
            try:
                line["strong_scaling"] = None if SKIP_COLOURED else strong_scaling_single(da)
            except Exception as e:
                line["strong_scaling"] = {"error": str(e)}
        if not args.no_config5:
            try:
                line["config5_central_certified"] = config5_central(da)
            except Exception as e:
                line["config5_central_certified"] = {"error": str(e)}
        if not args.no_config4:
            try:
                line["config4_tiers"] = config4_run(da, not args.no_cpu_baseline, args.cpu_c4_staircase)
            except Exception as e:
                line["config4_tiers"] = {"error": str(e)}
        if not args.no_cpu_baseline:
            cb = line["cpu_baseline"] = cpu_baseline(args, args.dataset, X0)
            # like for like: each GPU window against the oracle's rate over the same iterations
            line["config"]["speedup_vs_cpu_port"] = line["value"] / cb["same_window_as_value"]["value"]
            line["sustained"]["speedup_vs_cpu_port"] = line["sustained"]["value"] / cb["value"]
        line["scaling_100k_lattice"] = scaling_block(1, line.get("strong_scaling"), line.get("config5_lattice100k"), None)
        cache_store({"cpu_baseline": line.get("cpu_baseline"),
                     "strong_one_gpu": (line.get("strong_scaling") or {}).get("one_gpu"),
                     "provenance": provenance(args)})
    emit(real_stdout, line)


if __name__ == "__main__":
    main()
