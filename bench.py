#!/usr/bin/env python3
"""bench.py -- GCUPS of the query x database Smith-Waterman scan (BASELINE.json metric) on N MI355X of one node.

  python bench.py --gpus N --steps K --warmup W          (N > 1: spawns its N ranks itself, before any GPU call)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full pass of the hot path over the workload held in HBM: every query of the rank's shard scored against the
resident database (s4g_scan), the per-query top-400 selection on the device, and for N > 1 the RCCL all-gather of the
per-query top-hit lists.  Workloads (BASELINE.json `configs`, generator of SURVEY.md 8d):

  N = 1   configs[1]: 1000 synthetic queries U[300,400] x 100k-sequence synthetic DB, BLOSUM62 gap 10/1     (the metric's config)
  N > 1   configs[2]: the same 1000 queries in total x 570k sequences (Swiss-Prot scale), DB replicated in every GPU's HBM,
          queries split over the ranks: total work is fixed as N grows ("strong"); --workload configs2 runs it on one GPU
  --mode db-shard   configs[3]: a database shard GENERATED ON EACH DEVICE (--targets per GPU, default 150M/8), every rank scans
          the same query batch with the streaming scan + running top-k, all-gather + K-way merge; time-boxed: a batch of
          --queries per step out of the config's 10 000, wall time of the full config extrapolated
  --workload configs4   long-query stress: 64 queries U[2000,5000] x the configs[1] database

Rank 0 prints ONE JSON line (the last line of stdout).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
BYTES_PER_CELL = 16.0          # SURVEY.md 8d contract figure: streamed H/E column model, 2 int32 loads + 2 stores per cell
TOPK = 400                     # max_alignments default (/root/reference/sift4g/src/main.cpp:75)
LANES_PER_CLK_PER_SIMD = 16    # a wave64 vector instruction occupies its SIMD for 4 clocks (profiles/r03_valu_issue.txt)
GUIDE_CLOCK_MHZ = 2400.0       # MI355X_MICROARCH.md: peak engine clock; the roofline's `peak` is priced at it
# Algorithmic vector instructions per DP cell of each scan kernel, counted from the recurrence, as
# (all, of which "half-slot" class).  A gfx950 SIMD issues one vector instruction per 4-clock slot (16 lanes per clock) -- except
# that 32-bit add / sub / logic / mov / f32 add-mul-fma with two VGPR sources can share a slot two at a time when two waves present
# them together (tools/valu_issue.hip, profiles/r03_valu_issue.txt).  The roofline counts such an instruction as HALF a slot: the
# bound holds even if every one of them found a partner (in the kernels' mixes few do: 0.85-0.95 of a slot each, measured).
VALU_OPS_PER_CELL = {"sw_scan_pkc<32>": (3.25, 1.5), "sw_scan_pkd<32>": (3.25, 1.0), "sw_scan_pk<ArithI16M3,32>": (3.75, 2.0), "sw_scan_pk<ArithF16,32>": (3.75, 0.0),
                     "sw_scan_pk<ArithI16,32>": (4.5, 0.0), "sw_scan_i32<32>": (7.5, 3.0), "sw_scan_i32d<32>": (6.5, 1.0)}
WORKLOADS = {
    "configs1": dict(cfg=2, queries=1000, targets=100_000, qlen=(300, 400), label="BASELINE configs[1]"),
    "configs2": dict(cfg=3, queries=1000, targets=570_000, qlen=(300, 400), label="BASELINE configs[2] (Swiss-Prot scale)"),
    "configs4": dict(cfg=5, queries=64, targets=100_000, qlen=(2000, 5000), label="BASELINE configs[4] (long queries)"),
    # configs[2] with the length tail a real Swiss-Prot has and SURVEY 8d's generator clips at 3000 (synth.SWISSPROT_TAIL: the lognormal
    # unclipped, 0.3 % of the sequences from a 2-10 k component, eight planted giants of 10 200 .. 35 213 residues)
    "swissprot-tail": dict(cfg=3, queries=1000, targets=570_000, qlen=(300, 400), tail=True,
                           label="BASELINE configs[2] with an unclipped Swiss-Prot-like length tail"),
}
CONFIGS3_TARGETS, CONFIGS3_QUERIES = 150_000_000, 10_000


def log(msg):
    print(f"bench.py: {msg}", file=sys.stderr, flush=True)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (torch.distributed.run), relay rank 0's JSON
    line.  Nothing in this process has touched the GPU (a process that has initialised HIP must not spawn-and-replace)."""
    # fewer visible devices than ranks: say so and stop HERE, in seconds, instead of after N children have each imported torch
    # (torch.cuda.device_count() counts without initialising HIP on this image, so spawning afterwards stays legal)
    if os.environ.get("S4G_BENCH_SHARE_GPU") != "1":
        import torch
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            log(f"{args.gpus} ranks but {n_dev} GPU(s) visible (set S4G_BENCH_SHARE_GPU=1 to emulate the N-rank job on fewer devices)")
            sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log(f"spawning {args.gpus} ranks: {' '.join(cmd[1:8])} ...")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for l in p.stdout.splitlines():
        t = l.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr)
    if p.returncode != 0 or line is None:
        log(f"ranks exited with code {p.returncode}" + ("" if line else " and printed no result line"))
        sys.exit(p.returncode or 1)
    print(line, flush=True)
    sys.exit(0)


def _cpu_topology():
    """physical cores among the CPUs this process may run on (sysfs thread_siblings_list), or None"""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        cores = set()
        for c in cpus:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                cores.add(f.read().strip())
        return len(cpus), len(cores)
    except Exception:
        return None


def _cpu_limits():
    """what bounds the CPU baseline on this box besides the core count: the cgroup CPU quota (cpu.max of cgroup v2, cfs_quota of v1), the
    affinity mask, the physical-core map -> dict; every field None where the box does not say"""
    out = {"affinity_cpus": None, "physical_cores_in_affinity": None, "online_cpus": os.cpu_count(), "quota_cpus": None, "quota_source": None}
    try:
        out["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = _core_map()
    if cores:
        out["physical_cores_in_affinity"] = len(cores)
    try:                                                          # cgroup v2: the cgroup of this process, then its ancestors up to the root
        rel = [l.strip().split("::", 1)[1] for l in open("/proc/self/cgroup") if l.startswith("0::")]
        path = os.path.normpath("/sys/fs/cgroup/" + (rel[0] if rel else "/"))
        while path.startswith("/sys/fs/cgroup"):
            f = os.path.join(path, "cpu.max")
            if os.path.exists(f):
                q, per = open(f).read().split()[:2]
                if q != "max":
                    cpus = float(q) / float(per)
                    if out["quota_cpus"] is None or cpus < out["quota_cpus"]:
                        out["quota_cpus"], out["quota_source"] = round(cpus, 2), f
            if path == "/sys/fs/cgroup":
                break
            path = os.path.dirname(path)
    except Exception:
        pass
    if out["quota_cpus"] is None:
        try:                                                      # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                out["quota_cpus"], out["quota_source"] = round(q / per, 2), "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"
        except Exception:
            pass
    return out


def _core_map():
    """physical cores among the CPUs this process may run on -> list of sorted sibling lists (sysfs thread_siblings_list), or None"""
    try:
        from sift4g_amd.numa import _parse_cpulist
        allowed = os.sched_getaffinity(0)
        cores = {}
        for c in sorted(allowed):
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                sib = tuple(sorted(_parse_cpulist(f.read().strip()) & allowed))
            cores[sib] = list(sib)
        return sorted(cores.values())
    except Exception:
        return None


def _spread_mask(cores, n):
    """n hardware threads for n worker threads: one sibling of n physical cores spread evenly over the core list (so over sockets and
    CCDs) while n fits the physical cores, both siblings of as many cores as needed beyond that"""
    if n <= len(cores):
        step = len(cores) / float(n)
        return {cores[int(i * step)][0] for i in range(n)}
    mask = {c[0] for c in cores}
    for c in cores:
        for sib in c[1:]:
            if len(mask) < n:
                mask.add(sib)
    return mask


def cpu_baseline(cfg, n_threads, target_seconds=12.0):
    """The CPU port of the same recurrence timed on this host's cores on a bounded sample of the same workload
    (kind "port": a restatement, NOT swsharp's own SSE code, which is an absent submodule).  Uses the AVX2 inter-sequence port
    (oracle/sw_simd.c) in the precision ladder the reference's class of CPU scorer uses (swimd / OPAL): 32 targets
    per register in unsigned 8-bit lanes, int16 for the pairs that reach the 8-bit saturation mark, scalar beyond int16; falls
    back to the scalar oracle on CPUs without AVX2.  The thread count is swept first, every count under an affinity mask of that many
    hardware threads (one per physical core, spread over the sockets, while the count fits the cores): `cores` = the count the reported
    figure was taken at.  `limits` says what bounds the figure on this box besides the core count -- the cgroup CPU quota (a pod capped
    at 32 CPUs on a 128-core host scales NEGATIVELY beyond 32 threads: they share the quota), the affinity mask -- so that nobody reads
    the value as what the whole host does."""
    import oracle
    M = oracle.blosum62()
    q_off, t_off = cfg["q_off"], cfg["t_off"]
    simd = oracle.simd_available()
    nq = min(128 if simd else 16, len(q_off) - 1)
    qo = q_off[:nq + 1].copy()
    n_t = len(t_off) - 1
    limits = _cpu_limits()
    cores = _core_map()
    full_mask = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None

    def run(scan, lo, hi, threads, pin=True):
        to = (t_off[lo:hi + 1] - t_off[lo]).copy()
        tc = np.ascontiguousarray(cfg["t_codes"][int(t_off[lo]):int(t_off[hi])])
        pinned = False
        if pin and cores and full_mask is not None:
            try:
                os.sched_setaffinity(0, _spread_mask(cores, threads))          # the scan's pthreads inherit the mask of the thread that creates them
                pinned = True
            except OSError:
                pinned = False
        try:
            t0 = time.perf_counter()
            r = scan(cfg["q_codes"], qo, tc, to, M, 10, 1, threads)
            dt = time.perf_counter() - t0
        finally:
            if pinned:
                os.sched_setaffinity(0, full_mask)
        run.pinned = pinned
        return dt, float(qo[nq]) * float(to[-1]), r

    if not simd:
        scan = lambda *a: (oracle.scan_flat(*a), 0)
        sweep, best_threads = {}, n_threads
    else:
        scan = oracle.scan_simd_ladder_flat
        # thread sweep on a short sample each: all hardware threads, the physical cores, the quota, halves of that
        n_phys = len(cores) if cores else n_threads
        counts = {n_threads, n_phys, max(1, n_threads // 2), max(1, n_threads // 4), max(1, n_threads // 8), max(1, n_threads // 16)}
        if limits["quota_cpus"]:
            counts.add(max(1, min(n_threads, int(round(limits["quota_cpus"])))))
        sweep = {}
        for c in sorted(counts, reverse=True):
            m = min(n_t, max(8192, 256 * c))      # >= 8 batches of 32 targets per thread: the queue balances unequal batches
            dt, cells, _ = run(scan, 0, m, c)
            sweep[c] = round(cells / dt / 1e9, 2)
        best_threads = max(sweep, key=sweep.get)
    chunk = max(256, min(n_t, 64 * best_threads))
    cells, dt, nt, escalated = 0.0, 0.0, 0, 0
    while nt < n_t and dt < target_seconds:           # time-boxed: whole chunks of targets until ~target_seconds
        hi = min(n_t, nt + chunk)
        d, c, r = run(scan, nt, hi, best_threads)
        dt += d; cells += c; escalated += r[1]
        nt = hi
        chunk = min(chunk * 2, 1 << 18)
    out = dict(value=round(cells / dt / 1e9, 3), unit="GCUPS", cores=best_threads, kind="port")
    # what the figure is bounded by, in words
    n_phys = len(cores) if cores else None
    if limits["quota_cpus"] and limits["quota_cpus"] < (limits["affinity_cpus"] or 1e9):
        limits["bound_by"] = (f"capped: the cgroup CPU quota is {limits['quota_cpus']} CPUs of the {limits['affinity_cpus']} hardware threads the affinity mask shows "
                              f"({limits['quota_source']}); threads beyond the quota share it")
    elif sweep and best_threads < min(n_phys or n_threads, n_threads) and sweep[best_threads] > 1.05 * sweep.get(n_phys or n_threads, sweep[max(sweep)]):
        limits["bound_by"] = (f"no cgroup quota visible, yet the sweep peaks at {best_threads} of {n_phys} physical cores / {n_threads} hardware threads: a limit "
                              "the container does not expose (hypervisor CPU share, a shared host) or memory bandwidth; NOT what the whole host does")
    else:
        limits["bound_by"] = "the cores the affinity mask shows (no quota visible, the sweep does not peak early)"
    out["limits"] = limits
    out["pinned"] = bool(getattr(run, "pinned", False))
    out["pin_policy"] = ("an affinity mask of as many hardware threads as worker threads, one per physical core spread evenly over the core list while the "
                         "count fits the cores (the scan's pthreads inherit it)") if out["pinned"] else None
    out["quota_cpus"] = limits["quota_cpus"]
    if simd:
        # what the tiers are worth on ONE thread, and the int16 tier alone (rounds 1-4's baseline) at the same thread count
        m1 = min(n_t, 1024)
        d8, c8, _ = run(scan, 0, m1, 1)
        d16, c16, _ = run(oracle.scan_simd16_flat, 0, m1, 1)
        m16 = min(n_t, 16 * best_threads * 8)
        dA, cA, _ = run(oracle.scan_simd16_flat, 0, m16, best_threads)
        topo = _cpu_topology()
        out.update(per_thread_gcups=round(cells / dt / 1e9 / best_threads, 3),
                   one_thread_gcups={"ladder_8bit_first": round(c8 / d8 / 1e9, 2), "int16_only": round(c16 / d16 / 1e9, 2)},
                   int16_only_gcups=round(cA / dA / 1e9, 2),
                   thread_sweep_gcups={str(k): v for k, v in sweep.items()},
                   host=dict(hardware_threads=topo[0], physical_cores=topo[1], smt_siblings_used=bool(best_threads > topo[1])) if topo else None,
                   tiers=dict(pairs=int(nq) * int(nt), escalated_to_int16=int(escalated)))
    impl = ("AVX2 inter-sequence port (oracle/sw_simd.c): 32 unsigned 8-bit lanes first, int16 lanes for the pairs that reach the 8-bit saturation "
            "mark, scalar beyond int16 (the swimd / OPAL ladder)" if simd else "scalar C oracle (oracle/sw_oracle.c)")
    out["sample"] = (f"first {nq} queries x first {nt} targets of the same workload, {cells:.3e} cells in {dt:.1f} s, {impl}, "
                     f"{best_threads} pthreads (best of the sweep over {sorted(sweep) if sweep else [best_threads]})"
                     + (f"; {limits['bound_by']}" if limits.get("bound_by") else ""))
    return out


def prefilter_leg(cfg, dev_index, reps=3, check_targets=2000):
    """The stage in front of the scan: sift4g's k-mer prefilter over the same database on the device
    (s4g_prefilter_*: k = 5, 5000 candidates per query, the CLI's defaults, main.cpp:68-69), database already in HBM; the first
    `check_targets` targets are also run through oracle/kmer_prefilter.c and the two candidate lists compared."""
    import torch
    import oracle
    import sift4g_amd.device as s4gd
    q_off, t_off = cfg["q_off"], cfg["t_off"]
    nq, nt = len(q_off) - 1, len(t_off) - 1
    queries = [cfg["q_codes"][int(q_off[i]):int(q_off[i + 1])] for i in range(nq)]
    residues = int(t_off[-1])
    d_codes = torch.from_numpy(cfg["t_codes"][:max(residues, 1)]).cuda(dev_index)
    d_off = torch.from_numpy(np.asarray(t_off).astype(np.int64)).cuda(dev_index)
    best = dict(index=1e9, add=1e9, finish=1e9, call=1e9, finish_ids=1e9)
    stats = None
    # the caller's result arrays are allocated once, as the drop-in's are (a std::vector sized before the call): a fresh, never
    # touched 40 MB pays 3 ms of page faults inside the call (tools/prefilter_finish_probe.py: 4.4 ms fresh, 1.3 ms reused)
    out_bufs = (np.zeros((max(nq, 1), 5000), np.uint32), np.zeros((max(nq, 1), 5000), np.float32))
    for _ in range(reps + 1):                          # the first round grows the buffers
        torch.cuda.synchronize()
        t0 = time.perf_counter(); p = s4gd.Prefilter(queries, 5, 5000, device=dev_index); t1 = time.perf_counter()
        p.add_device(d_codes, d_off, first_id=0)
        torch.cuda.synchronize()                       # (the merge of a chunk's candidates into the lists runs on behind the call: count it)
        t2 = time.perf_counter()
        p.finish(out=out_bufs); t3 = time.perf_counter()
        p.finish(scores=False, out=out_bufs); t4 = time.perf_counter()      # ids only: what searchDatabase returns and the drop-in asks for
        stats = p.stats()
        p.close()
        best = dict(index=min(best["index"], t1 - t0), add=min(best["add"], t2 - t1), finish=min(best["finish"], t3 - t2),
                    call=min(best["call"], t3 - t0), finish_ids=min(best["finish_ids"], t4 - t3),
                    call_ids=min(best.get("call_ids", 1e9), (t2 - t0) + (t4 - t3)))
    m = min(check_targets, nt)
    sample = [cfg["t_codes"][int(t_off[i]):int(t_off[i + 1])] for i in range(m)]
    t0 = time.perf_counter()
    r = oracle.Prefilter(queries, 5, 5000)
    r.add(sample)
    r_ids, r_counts, r_scores, _ = r.finish()
    cpu_s = time.perf_counter() - t0
    p = s4gd.Prefilter(queries, 5, 5000, device=dev_index)
    p.add(sample)
    ids, counts, scores = p.finish()
    p.close()
    bad = int((counts != r_counts).sum())
    for q in range(nq):
        n = int(min(counts[q], r_counts[q]))
        bad += int((ids[q, :n] != r_ids[q, :n]).sum()) + int((scores[q, :n].view(np.uint32) != r_scores[q, :n].view(np.uint32)).sum())
    # against the HBM roof: what the stage has to move at the least -- every residue in once, one 12-byte candidate (query, score, id)
    # out per (query, target) pair with a hit -- over the time of the call; the counter traffic of the same call (rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE passes of tools/prefilter_bench.py, profiles/r05_prefilter) rides along
    alg_bytes = residues + 12.0 * stats["pairs"]
    traffic = None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "prefilter_traffic.json")))
        if pj.get("residues") == residues and pj.get("queries") == nq:
            traffic = pj
    except Exception:
        traffic = None
    # (the rate is over the WHOLE call -- index + add + finish, host to host -- not over the add alone)
    roof = {"bound": "hbm", "achieved": round(alg_bytes / best["call"] / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg_bytes / best["call"] / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": alg_bytes,
            "traffic": traffic["hbm_bytes_per_add"] if traffic else None, "traffic_measured_in_run": False,
            "traffic_source": traffic["source"] if traffic else None,
            "traffic_full_size_add": traffic.get("hbm_bytes_per_full_size_add") if traffic else None,
            "note": "integer gather work: the target kernel is bound by vector issue (sorting and walking a target's hits in LDS / registers), its index "
                    "lookups stay in L2 (1 MB of slots + 1.3 MB of bucket words); traffic = FETCH_SIZE + WRITE_SIZE of every kernel of one add call, "
                    "averaged over the add calls of the profiled run as in rounds 3-4 (traffic_full_size_add: the 100 000-target calls alone)"}
    return {"stage": "k-mer prefilter (database_search.cpp:66-253) on the device, database resident in HBM",
            "kmer_length": 5, "max_candidates": 5000, "queries": nq, "targets": nt, "residues": residues, "roofline": roof,
            "path": "fused" if stats.get("fused") else "general", "general_path_targets": stats.get("general_path_targets"),
            "index_ms": round(best["index"] * 1e3, 3), "add_ms": round(best["add"] * 1e3, 3), "finish_ms": round(best["finish"] * 1e3, 3),
            "finish_ids_only_ms": round(best["finish_ids"] * 1e3, 3),
            "call_ms": round(best["call"] * 1e3, 3), "residues_per_s": round(residues / best["call"], 1),
            "residues_per_s_basis": "database residues / (index + add + finish with ids AND scores) of the fastest whole call",
            "call_ids_only_ms": round(best["call_ids"] * 1e3, 3), "residues_per_s_ids_only": round(residues / best["call_ids"], 1),
            "ids_only_basis": "index + add + finish with ids only: what searchDatabase returns (database_search.cpp:166-178) and the drop-in asks for",
            "add_residues_per_s": round(residues / best["add"], 1), "kmer_hits": stats["hits"], "pairs_with_a_hit": stats["pairs"],
            "check": {"targets": m, "mismatches": bad, "against": "oracle/kmer_prefilter.c (CPU restatement, 1 thread)",
                      "cpu_residues_per_s": round(int(t_off[m]) / cpu_s, 1)}}


def end_to_end_leg(s4g, torch, db, q_codes, q_off, scorer, scores, dev, cells, reps=3):
    """"report kernel-only and end-to-end (incl. H2D of queries, D2H of results; DB resident)".  Wall clock, host to host,
    of what a caller with host buffers pays per batch once the database is in HBM: (a) the query set packed and uploaded
    (s4g_queries_create), the scan, the device top-400, the three result arrays downloaded; (b) the same with the WHOLE score matrix
    downloaded into pinned host memory instead of the lists (what the per-query swsharp contract implies: step 2's callback sees every
    score on the host).  Outside the timed region; never `value`."""
    host = torch.empty(scores.shape, dtype=torch.int32, pin_memory=True)
    legs = {"top_lists": [], "full_matrix": []}
    parts = {"queries_h2d_ms": [], "scan_ms": [], "topk_ms": [], "lists_d2h_ms": [], "matrix_d2h_ms": []}
    for it in range(reps + 1):                         # the first round warms the query-block cache and the pinned buffer's pages
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        q = s4g.QuerySet(codes=q_codes, offsets=q_off, device=dev.index)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        s4g.scan(db, q, scorer, scores)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ids, sc, cnt = s4g.topk_device(db, scores, TOPK, 1)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        res = (ids.cpu(), sc.cpu(), cnt.cpu())
        t4 = time.perf_counter()
        host.copy_(scores, non_blocking=True)
        torch.cuda.synchronize(); t5 = time.perf_counter()
        del q, res
        if it:
            legs["top_lists"].append(t4 - t0); legs["full_matrix"].append((t3 - t0) + (t5 - t4))
            for k, v in zip(parts, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                parts[k].append(v * 1e3)
    del host
    mean = lambda v: float(np.mean(v))
    return {"what": "host to host per batch, database resident in HBM: queries packed + uploaded, scan, device top-400, results downloaded; wall clock, "
                    f"mean of {reps} after one warm-up; never `value`",
            "top_lists": {"ms": round(mean(legs["top_lists"]) * 1e3, 3), "gcups": round(cells / mean(legs["top_lists"]) / 1e9, 2),
                          "d2h_bytes": int(scores.shape[0]) * (2 * TOPK + 1) * 4},
            "full_matrix": {"ms": round(mean(legs["full_matrix"]) * 1e3, 3), "gcups": round(cells / mean(legs["full_matrix"]) / 1e9, 2),
                            "d2h_bytes": int(scores.numel()) * 4, "host_memory": "pinned",
                            "d2h_gb_per_s": round(scores.numel() * 4 / (mean(parts["matrix_d2h_ms"]) * 1e-3) / 1e9, 2)},
            "parts_ms": {k: round(mean(v), 3) for k, v in parts.items()}}


def oracle_selfcheck(q_codes, q_off, fetch_target, pairs, gpu_scores, n_threads):
    """Re-score (query, target) pairs with the CPU oracle (outside the timed region) and compare with what the GPU reported."""
    import oracle
    pq = np.asarray([p[0] for p in pairs], dtype=np.uint32)
    uniq, inv = np.unique(np.asarray([p[1] for p in pairs], dtype=np.int64), return_inverse=True)
    seqs = [fetch_target(int(t)) for t in uniq]
    t_off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    t_off[1:] = np.cumsum([len(s) for s in seqs]).astype(np.uint64)
    t_codes = np.ascontiguousarray(np.concatenate(seqs)) if seqs else np.zeros(1, np.uint8)
    t0 = time.perf_counter()
    want = oracle.pairs_flat(np.ascontiguousarray(q_codes), np.ascontiguousarray(q_off), t_codes, t_off, pq, inv.astype(np.uint32),
                             oracle.blosum62(), 10, 1, n_threads)
    bad = int((want != np.asarray(gpu_scores, dtype=np.int32)).sum())
    return bad, time.perf_counter() - t0


def tail_leg(s4g, s4gd, synth, torch, cfg, allq, scorer, dev, nq, nt, qlen, wl, db, kern_ms, topk_ms, cells_rank, gcups):
    """--workload swissprot-tail, after the timed region: length statistics of the database, the same scan (a) without the sequences
    beyond 12 288 residues and (b) on the set clipped at 3000 (BASELINE configs[2] as SURVEY 8d generates it)."""
    t_off = cfg["t_off"].astype(np.int64)
    lens = np.diff(t_off)
    info = {"lengths": {"mean": round(float(lens.mean()), 1), "max": int(lens.max()), "over_3000": int((lens > 3000).sum()),
                        "over_10000": int((lens > 10000).sum()), "over_12288": int((lens > 12288).sum()),
                        "residues_over_3000_frac": round(float(lens[lens > 3000].sum() / lens.sum()), 4)},
            "tile_seqs": [int(n) for n in db.tiles()],
            "giants": [int(lens[t]) for t in cfg["giants"]]}

    def timed(database, queries, reps=2):
        buf = torch.empty((queries.n, database.n), dtype=torch.int32, device=dev)
        cells = s4gd.scan_cells(database, queries, 0, queries.n)
        ms = []
        for it in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            s4g.scan(database, queries, scorer, buf)
            e1.record()
            torch.cuda.synchronize()
            if it:
                ms.append(e0.elapsed_time(e1))
        st = s4gd.scan_stats(dev.index)
        del buf
        return float(np.mean(ms)), cells, st

    # (a) the same sequences minus the long-sequence tile
    keep = np.nonzero(lens <= 12288)[0]
    sel_off = np.zeros(len(keep) + 1, dtype=np.uint64)
    sel_off[1:] = np.cumsum(lens[keep]).astype(np.uint64)
    idx = np.concatenate([np.arange(t_off[t], t_off[t + 1]) for t in np.nonzero(lens > 12288)[0]]) if len(keep) < len(lens) else np.zeros(0, dtype=np.int64)
    mask = np.ones(int(t_off[-1]), dtype=bool)
    mask[idx] = False
    db_a = s4g.Database(codes=np.ascontiguousarray(cfg["t_codes"][:int(t_off[-1])][mask]), offsets=sel_off, device=dev.index)
    ms_a, cells_a, st_a = timed(db_a, allq)
    del db_a
    info["without_sequences_over_12288"] = {"scan_ms": round(ms_a, 3), "gcups": round(cells_a / ms_a / 1e6, 2), "cells": float(cells_a)}
    info["long_tile"] = {"cells": float(cells_rank - cells_a), "cells_frac": round(float(cells_rank - cells_a) / float(cells_rank), 5),
                         "scan_ms_added": round(kern_ms - ms_a, 3), "scan_ms_added_frac": round((kern_ms - ms_a) / kern_ms, 4),
                         "note": "scan time of the timed steps minus the scan time of the same database without its sequences beyond 12 288 residues; "
                                 "those sit in a tile of their own that runs on a side stream underneath the main tile (DESIGN.md 5)"}
    # (b) the clipped set: BASELINE configs[2] as SURVEY 8d generates it
    cfg_c = synth.make_config(wl["cfg"], n_queries=nq, n_targets=nt, q_len=qlen)
    db_c = s4g.Database(codes=cfg_c["t_codes"], offsets=cfg_c["t_off"], device=dev.index)
    ms_c, cells_c, st_c = timed(db_c, allq)
    del db_c, cfg_c
    info["clipped_at_3000"] = {"scan_ms": round(ms_c, 3), "gcups": round(cells_c / ms_c / 1e6, 2), "cells": float(cells_c)}
    info["scan_gcups"] = round(cells_rank / kern_ms / 1e6, 2)
    info["ratio_to_clipped"] = round((cells_rank / kern_ms) / (cells_c / ms_c), 4)
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["auto"] + sorted(WORKLOADS), default="auto",
                    help="auto: configs1 at --gpus 1, configs2 (strong scaling) at --gpus > 1")
    ap.add_argument("--queries", type=int, default=None, help="queries in TOTAL (query-shard) / per step (db-shard); default: the workload's")
    ap.add_argument("--targets", type=int, default=None, help="database sequences (db-shard: per GPU); default: the workload's")
    ap.add_argument("--qlen", type=int, nargs=2, default=None, help="query length range; default: the workload's")
    ap.add_argument("--force-i32", action="store_true", help="run every query through the int32 kernel")
    ap.add_argument("--no-f16", action="store_true", help="skip the maximum3-based tiers (plain packed int16 instead)")
    ap.add_argument("--no-drift", action="store_true", help="skip the drifting-frame kernel (maximum3 tier instead)")
    ap.add_argument("--diag-frame", action="store_true", help="the drifting-frame tier in its anti-diagonal form (sw_scan_pkd, rounds 1-4) instead of the column frame (sw_scan_pkc)")
    ap.add_argument("--no-spec", action="store_true", help="tiers by score bound only: no speculative use of the drifting-frame kernel for long queries")
    ap.add_argument("--mode", choices=["query-shard", "db-shard"], default="query-shard")
    ap.add_argument("--step-seconds", type=float, default=120.0, help="db-shard without --targets: size of the shard a step scans, in seconds per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-selfcheck", action="store_true")
    ap.add_argument("--no-prefilter", action="store_true", help="skip the k-mer prefilter leg after the timed region")
    ap.add_argument("--no-tiers", action="store_true", help="skip the two extra int32-only steps after the timed region")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-to-host leg (queries H2D + scan + top-k + results D2H) after the timed region")
    ap.add_argument("--no-scale-ref", action="store_true", help="skip the configs[2]-on-one-GPU leg (the N > 1 lines' denominator) after the timed region")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)       # does not return

    # The contract is ONE line on stdout.  RCCL prints a version banner to the C stdout of every rank when a process group starts
    # (seen with the single-rank RCCL test: four lines in front of the JSON), and any library may chatter there: from here on file
    # descriptor 1 is stderr for everybody, and the result line goes to a private copy of the real stdout.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            log(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        sys.exit(2)
    # S4G_BENCH_SHARE_GPU=1: ranks may share devices (a 1-GPU test box running the N-rank job): rendezvous and gather go through
    # gloo, since RCCL refuses two ranks on one device.  Reported in the JSON ("emulated"); never a scaling measurement.
    share = os.environ.get("S4G_BENCH_SHARE_GPU") == "1"
    n_dev = torch.cuda.device_count()
    if n_dev < 1 or (n_dev < world and not share):
        if rank == 0:
            log(f"{world} ranks but {n_dev} GPU(s) visible (set S4G_BENCH_SHARE_GPU=1 to emulate the N-rank job on fewer devices)")
        sys.exit(2)
    dev_index = local_rank % n_dev
    # one process per GPU: keep this rank's host threads (the library's worker pool, staging copies, the process group's helpers)
    # on the NUMA node its GPU hangs off -- before the first HIP call, so the runtime's own threads inherit the mask.  Only for
    # N > 1: the N = 1 line times the CPU baseline on ALL host cores.
    affinity = None
    if world > 1 and os.environ.get("S4G_BENCH_NO_PIN") != "1":
        from sift4g_amd import numa
        affinity = numa.pin_to_gpu(dev_index)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import sift4g_amd as s4g
    from sift4g_amd import synth, shard
    from sift4g_amd import device as s4gd

    use_dist = world > 1 or os.environ.get("S4G_BENCH_FORCE_DIST") == "1"     # the latter exercises the RCCL path on one GPU
    backend = None
    rccl_ranks = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = "gloo" if (share and n_dev < world) else "nccl"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
            ones = torch.ones(1, dtype=torch.int32, device=dev)
        else:
            dist.init_process_group("gloo")
            ones = torch.ones(1, dtype=torch.int32)
        dist.all_reduce(ones)                       # how many ranks really joined the collective
        rccl_ranks = int(ones.item())
    # the gather of the hit lists runs BELOW Python when the ranks are on RCCL: s4g_gather_topk of the C ABI (its own communicator, made
    # from an id rank 0 creates and the process group hands round; pack / all-gather / unpack or K-way merge on the step's stream).
    # S4G_BENCH_TORCH_GATHER=1 keeps the torch.distributed form of sift4g_amd/shard.py; gloo (emulated) runs always use it.
    gather_comm = None
    if use_dist and backend == "nccl" and os.environ.get("S4G_BENCH_TORCH_GATHER") != "1":
        # every rank must end up on the same form: the ranks agree (an all-reduce of "I have a communicator") before anybody uses
        # it, and all of them fall back to the torch.distributed gather if any of them could not make one
        try:
            gather_comm = shard.Comm.from_process_group(dev_index)
            ok = 1
        except Exception as exc:          # librccl not loadable, ncclCommInitRank refused ...
            log(f"rank {rank}: no C-ABI communicator ({exc}); falling back to the torch.distributed gather")
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and gather_comm is not None:
            gather_comm.close()
            gather_comm = None

    db_shard = args.mode == "db-shard"
    tail_wl = (not db_shard) and args.workload != "auto" and bool(WORKLOADS[args.workload].get("tail"))
    flags = dict(force_i32=args.force_i32, no_f16=args.no_f16, no_drift=args.no_drift, no_spec=args.no_spec, diag_frame=args.diag_frame)
    scorer = s4g.Scorer("BLOSUM_62", 10, 1)
    n_cpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t_gen = time.perf_counter()
    cfg = None

    if db_shard:
        # ---- configs[3]: this rank's shard of a UniRef90-scale database, generated where it will be scanned ----
        nq = args.queries if args.queries is not None else 32
        # time-boxed ("run N DB tiles, report measured GCUPS and extrapolated wall time"): the shard a step scans is cut
        # so that one step is about --step-seconds at ~11 TCUPS, but never below 1.25 M targets (two database tiles): 32 queries
        # scan the whole 18.75 M-target share, the config's own 10 000 queries scan 1.25 M targets of it per step
        if args.targets is not None:
            n_per = args.targets
        else:
            fit = int(args.step_seconds * 11.0e12 / (nq * 350.0 * 340.0))
            n_per = max(1_250_000, min(CONFIGS3_TARGETS // 8, fit))
        qlen = tuple(args.qlen) if args.qlen else (300, 400)
        wl_label = "BASELINE configs[3] (UniRef90 scale, DB sharded)"
        qseed = synth.SEED_BASE + 4
        ql = synth.query_lengths(qseed + 0x100, nq, *qlen)
        q_codes, q_off = synth.random_set(qseed + 0x200, ql)
        codes_d, offsets_d = synth.device_database(n_per, synth.SEED_BASE + 4 + 0x1000 * (rank + 1), dev)
        # one full-length copy of every query planted in exactly one rank's shard (query i -> rank i % world), at the head of a
        # target that is long enough; the owners tell everybody where (rank 0 checks the merged lists)
        lens_h = (offsets_d[1:] - offsets_d[:-1]).cpu().numpy()
        where = torch.zeros(nq, dtype=torch.int64)
        taken = set()
        for i in range(nq):
            if i % world != rank:
                continue
            p = (i + 1) * (n_per // (nq + 1))
            while lens_h[p] < ql[i] or p in taken:
                p += 1
            taken.add(p)
            where[i] = p
            o = int(offsets_d[p])
            codes_d[o:o + int(ql[i])] = torch.from_numpy(q_codes[int(q_off[i]):int(q_off[i + 1])].copy()).to(dev)
        if use_dist:
            if backend == "nccl":
                where = where.to(dev)
            dist.all_reduce(where)
        planted = {i: int(where[i]) for i in range(nq)}
        torch.cuda.synchronize()
        t_up = time.perf_counter()
        db = s4g.Database.from_device(codes_d, offsets_d)
        allq = s4g.QuerySet(codes=q_codes, offsets=q_off, device=dev_index)
        q_begin, q_count = 0, nq
        residues_rank = int(offsets_d[-1])
        n_targets_total = n_per * world
        if world == 1 and not args.no_cpu_baseline:     # the CPU baseline's bounded sample: the shard's first sequences, on the host
            m = min(n_per, 1 << 18)
            cfg = dict(q_codes=q_codes, q_off=q_off, t_codes=codes_d[:int(offsets_d[m])].cpu().numpy(),
                       t_off=offsets_d[:m + 1].cpu().numpy().astype(np.uint64))
    else:
        # ---- configs[1] / [2] / [4]: DB replicated, the queries split over the ranks ----
        wl_name = args.workload if args.workload != "auto" else ("configs1" if world == 1 else "configs2")
        wl = WORKLOADS[wl_name]
        wl_label = wl["label"]
        nq = args.queries if args.queries is not None else wl["queries"]
        nt = args.targets if args.targets is not None else wl["targets"]
        qlen = tuple(args.qlen) if args.qlen else wl["qlen"]
        cfg = synth.make_config(wl["cfg"], n_queries=nq, n_targets=nt, q_len=qlen, tail=synth.SWISSPROT_TAIL if wl.get("tail") else None)
        q_codes, q_off = cfg["q_codes"], cfg["q_off"]
        t_up = time.perf_counter()
        db = s4g.Database(codes=cfg["t_codes"], offsets=cfg["t_off"], device=dev_index)
        allq = s4g.QuerySet(codes=q_codes, offsets=q_off, device=dev_index)
        bounds = shard.shard_bounds(nq, world)
        q_begin, q_count = bounds[rank], bounds[rank + 1] - bounds[rank]
        residues_rank = int(cfg["t_off"][-1])
        n_targets_total = nt
    torch.cuda.synchronize()
    t_ready = time.perf_counter()
    if rank == 0:
        log(f"generate {t_up - t_gen:.1f} s; pack DB ({residues_rank} residues) + queries into HBM {t_ready - t_up:.2f} s "
            f"(outside the timed region)")

    cells_rank = s4gd.scan_cells(db, allq, q_begin, q_count) if q_count else 0
    plan = s4gd.scan_plan(db, allq, scorer)
    if args.force_i32:
        plan["kernel"], plan["strip_rows"] = ("sw_scan_i32<32>" if args.no_drift else "sw_scan_i32d<32>"), 32   # BLOSUM62 10/1 fits the drifting int32 frame
    elif args.no_f16 and ("M3" in plan["kernel"] or "pkd" in plan["kernel"] or "pkc" in plan["kernel"]):
        plan["kernel"] = "sw_scan_pk<ArithI16,32>"
    elif args.no_drift and ("pkd" in plan["kernel"] or "pkc" in plan["kernel"]):
        plan["kernel"] = "sw_scan_pk<ArithI16M3,32>"
    elif args.diag_frame and "pkc" in plan["kernel"]:
        plan["kernel"] = "sw_scan_pkd<32>"
    scores = None if db_shard else torch.empty((max(q_count, 1), db.n), dtype=torch.int32, device=dev)
    # per step: scan start, scan end, top-k end, gather (+ merge) end -- on the stream every launch of the step goes to
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(args.steps)]
    gather_host_s = []              # gloo (emulated runs): the gather is host-synchronous, events do not see it
    gather_group = None

    def step(i=None, fl=flags):
        if i is not None:
            ev[i][0].record()
        def gathered(fn):
            t_g = time.perf_counter()
            g = fn()
            if i is not None:
                ev[i][3].record()
                gather_host_s.append(time.perf_counter() - t_g)
            return g
        if db_shard:
            ids, sc, cnt = s4g.scan_topk(db, allq, scorer, TOPK, 1, **fl)             # streaming scan + tile top-k + merge
            if i is not None:
                ev[i][1].record(); ev[i][2].record()                                  # (the top-k is inside the streaming scan)
            return (ids, sc, cnt), gathered(lambda: shard.gather_db_shards(ids, sc, cnt, rank * n_per, TOPK, gather_group, comm=gather_comm))   # all-gather + K-way merge
        if q_count:
            s4g.scan(db, allq, scorer, scores, q_begin=q_begin, q_count=q_count, **fl)
        if i is not None:
            ev[i][1].record()
        if q_count:
            ids, sc, cnt = s4g.topk_device(db, scores[:q_count], TOPK, 1)             # steps 2-3 on the device
        else:
            ids = torch.full((0, TOPK), -1, dtype=torch.int32, device=dev)
            sc = torch.zeros((0, TOPK), dtype=torch.int32, device=dev)
            cnt = torch.zeros((0,), dtype=torch.int32, device=dev)
        if i is not None:
            ev[i][2].record()
        return (ids, sc, cnt), gathered(lambda: shard.gather_query_shards(ids, sc, cnt, allq.n, gather_group, comm=gather_comm))        # all-gather of the hit lists (N > 1)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = step(i)
    fence()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev])) if args.steps else 0.0
    topk_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) if args.steps else 0.0
    gather_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev])) if args.steps else 0.0
    if backend == "gloo" and gather_host_s:
        gather_ms = float(np.mean(gather_host_s)) * 1e3
    stats = s4gd.scan_stats(dev_index)
    clk = s4gd.scan_clock(dev_index)                # sampled inside the scan kernel of the last timed step
    # The gather below Python (s4g_gather_topk: its own RCCL communicator, pack / all-gather / unpack or merge kernels) against the
    # torch.distributed form of the same exchange on the last step's lists, once, outside the timed region: every rank must hold the
    # same merged lists either way.  (The multi-rank RCCL path has only ever run with one rank on the boxes this was built on.)
    gather_check = None
    if gather_comm is not None and last is not None:
        (l_ids_, l_sc_, l_cnt_), g_c = last
        if db_shard:
            g_t = shard.gather_db_shards(l_ids_, l_sc_, l_cnt_, rank * n_per, TOPK, gather_group, comm=None)
        else:
            g_t = shard.gather_query_shards(l_ids_, l_sc_, l_cnt_, allq.n, gather_group, comm=None)
        torch.cuda.synchronize()
        same = all(torch.equal(a.to(torch.int32), b.to(torch.int32)) for a, b in zip(g_c, g_t))
        flag = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_check = {"c_abi_equals_torch_distributed_on_every_rank": bool(int(flag.item())), "lists": int(g_c[0].shape[0])}
        if not gather_check["c_abi_equals_torch_distributed_on_every_rank"]:
            log(f"rank {rank}: GATHER CROSS-CHECK FAILED: s4g_gather_topk and the torch.distributed gather disagree")

    red_dev = dev if backend != "gloo" else torch.device("cpu")
    t = torch.tensor([dt, kern_ms, topk_ms, gather_ms], dtype=torch.float64, device=red_dev)
    tmin = t.clone()
    c = torch.tensor([float(cells_rank)], dtype=torch.float64, device=red_dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    dt = float(t[0])
    phases = {"scan_ms": {"max": round(float(t[1]), 3), "min": round(float(tmin[1]), 3)},
              "topk_ms": {"max": round(float(t[2]), 3), "min": round(float(tmin[2]), 3)},
              "gather_ms": {"max": round(float(t[3]), 3), "min": round(float(tmin[3]), 3)},
              "note": "per rank, mean over the timed steps (HIP events on the step's stream; gloo gathers by host clock); max / min over ranks"}
    cells_total = float(c[0])

    line = None
    if rank == 0:
        gcups = cells_total * args.steps / dt / 1e9
        kern_gcups = cells_rank / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        ops, half_ops = VALU_OPS_PER_CELL.get(plan["kernel"], (0.0, 0.0))
        slots = ops - 0.5 * half_ops                 # issue slots per cell if every half-slot instruction found a partner
        # ---- the roof that binds: vector-ALU issue.  `peak` = CUs x 4 SIMDs x 16 lanes per clock x the guide's 2.4 GHz;
        #      the clock the timed kernel really ran at is sampled INSIDE it (s4g_scan_clock) and prices `peak_at_measured_clock` ----
        import ctypes as _ct
        cus = torch.cuda.get_device_properties(dev_index).multi_processor_count
        clock_mhz = clk["shader_mhz"] if clk["launches"] and clk["shader_mhz"] > 500.0 else 0.0
        valu_peak = cus * 4 * LANES_PER_CLK_PER_SIMD * GUIDE_CLOCK_MHZ * 1e6 / 1e12             # T lane-ops/s at 2.4 GHz
        valu_peak_meas = cus * 4 * LANES_PER_CLK_PER_SIMD * clock_mhz * 1e6 / 1e12 if clock_mhz else None
        valu_achieved = kern_gcups * slots / 1e3
        valu_all_full = kern_gcups * ops / 1e3       # round 2's figure: every instruction a full slot
        # executed (not algorithmic) lane-ops per cell: SQ_INSTS_VALU x 64 / cells of a committed rocprofv3 --pmc pass of this
        # command (profiles/valu_executed.json, tagged with the commit it was collected on)
        # (both stored figures are keyed on a hash of the scan kernels' sources -- sift4g_amd/srchash.py -- and stay null when the
        # build that is running was made from other sources than the build that was profiled)
        from sift4g_amd.srchash import kernel_source_sha
        src_sha = kernel_source_sha()
        exec_ops, exec_src = None, None
        try:
            ve = json.load(open(os.path.join(ROOT, "profiles", "valu_executed.json"))).get(plan["kernel"])
            if ve and ve.get("workload_cells") == cells_rank and ve.get("kernel_source_sha") == src_sha:
                exec_ops, exec_src = ve["lane_ops_per_cell"], ve.get("source")
        except Exception:
            exec_ops = None
        valu_executed = kern_gcups * (exec_ops - 0.5 * half_ops) / 1e3 if exec_ops else None      # executed slots (the half-slot share as counted above)
        traffic, traffic_src = None, None
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # written from rocprofv3 --pmc passes of this command
        if os.path.exists(prof):
            try:
                pj = json.load(open(prof))
                if pj.get("kernel") == plan["kernel"] and pj.get("workload_cells") == cells_rank and pj.get("kernel_source_sha") == src_sha:
                    traffic, traffic_src = pj.get("hbm_bytes_per_launch"), pj.get("source")
            except Exception:
                traffic = None
        hbm_achieved = kern_gcups * BYTES_PER_CELL
        tier = "int32" if plan["kernel"].startswith("sw_scan_i32") else ("f16" if "F16" in plan["kernel"] else "int16")
        dtype = {"int32": "int32",
                 "int16": "int16x2 packed, exact by score-bound routing; int32 out",
                 "f16": "f16x2 arithmetic on integers (exact below 4096 by score-bound routing); int32 out"}[tier]
        out = {
            "metric": "GCUPS (SW cell updates/s) on query x DB scan",
            "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3), "higher_is_better": True,
            "scaling": "weak" if db_shard else "strong",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "rccl_ranks": rccl_ranks, "backend": ("RCCL (nccl)" if backend == "nccl" else backend), "emulated": bool(share and n_dev < world),
            "gather": ("s4g_gather_topk (C ABI: RCCL all-gather + merge kernels below Python)" if gather_comm is not None else
                       ("torch.distributed (sift4g_amd/shard.py)" if use_dist else None)),
            "config": {
                "workload": (f"{wl_label}: {nq} synthetic queries per step len U[{qlen[0]},{qlen[1]}] x {n_per} device-generated targets per GPU "
                             f"({n_targets_total} in total) lognormal(5.68,0.55) clip[30,3000], BLOSUM62 gap 10/1, streaming scan + running top-{TOPK}"
                             if db_shard else
                             f"{wl_label}: {nq} synthetic queries len U[{qlen[0]},{qlen[1]}] x {n_targets_total} synthetic targets "
                             + ("lognormal(5.68,0.55) clip[30,36000] + 0.3 % lognormal(7.6,0.5) + 8 giants of 10200..35213 residues"
                                if tail_wl else "lognormal(5.68,0.55) clip[30,3000]") +
                             f" + planted homologs, BLOSUM62 gap 10/1, full scan + top-{TOPK}"),
                "cells_per_step": cells_total,
                "parallelism": (f"db-shard x{world}: every GPU scans all queries against its own {n_per} targets, all-gather + K-way merge" if db_shard
                                else f"query-shard x{world}: DB replicated, {nq} queries split over the ranks, all-gather of the hit lists"),
                "kernel": plan["kernel"], "grid_blocks": plan["grid_blocks"], "strip_rows": plan["strip_rows"]},
            "roofline": {
                "bound": "valu", "achieved": round(valu_achieved, 3), "peak": round(valu_peak, 3), "unit": "T lane-ops/s",
                "frac": round(valu_achieved / valu_peak, 4) if valu_peak else None,
                # the same launch under the other issue model (every instruction a full slot, none paired): read the two together
                "frac_every_instruction_a_full_slot": round(valu_all_full / valu_peak, 4) if valu_peak else None,
                "frac_model": "frac: two-VGPR-source 32-bit add/sub count half an issue slot (a lower bound on what the kernel uses); "
                              "frac_every_instruction_a_full_slot: no pairing at all (an upper bound)",
                "traffic": traffic, "traffic_measured_in_run": False, "kernel_source_sha": src_sha,
                "kernel": plan["kernel"], "kernel_ms": round(kern_ms, 3), "kernel_gcups": round(kern_gcups, 2),
                "ops_per_cell": ops, "half_slot_ops_per_cell": half_ops, "issue_slots_per_cell": slots,
                "peak_basis": f"{cus} CU x 4 SIMD x {LANES_PER_CLK_PER_SIMD} lanes/clk x {GUIDE_CLOCK_MHZ:.0f} MHz (guide peak engine clock): one vector "
                              "instruction per SIMD and 4-clock issue slot; two-VGPR-source add/sub/logic/mov count half a slot (they can pair up across waves)",
                # achieved = algorithmic issue slots per cell (x 64 lanes) x cells / the kernel's HIP-event time; frac_algorithmic = frac
                "frac_algorithmic": round(valu_achieved / valu_peak, 4) if valu_peak else None,
                "executed_ops_per_cell": exec_ops, "executed": round(valu_executed, 3) if valu_executed else None,
                "frac_executed": round(valu_executed / valu_peak, 4) if valu_executed else None,
                "executed_source": exec_src,
                # the clock the timed kernel ran at, from s_memtime / s_memrealtime samples taken by its own blocks
                "clock_in_kernel": {"shader_mhz": round(clock_mhz, 1), "min_block": round(clk["shader_mhz_min"], 1),
                                    "max_block": round(clk["shader_mhz_max"], 1), "blocks": clk["blocks"],
                                    "kernel_ms_by_100MHz_counter": round(clk["kernel_ms"], 3), "tier": clk["tier"],
                                    "launches_sampled": clk["launches"],
                                    # how long the persistent blocks were busy, mean over blocks / the longest: what is missing to 1 is the
                                    # drain of the work queue (few long items: configs[4] 0.94; configs[1] 0.999)
                                    "block_busy_ms": {"mean": round(clk["block_ms_mean"], 3), "min": round(clk["block_ms_min"], 3)},
                                    "blocks_busy_frac": round(clk["block_ms_mean"] / clk["kernel_ms"], 4) if clk["kernel_ms"] > 0 else None},
                "peak_at_measured_clock": round(valu_peak_meas, 3) if valu_peak_meas else None,
                "frac_algorithmic_at_measured_clock": round(valu_achieved / valu_peak_meas, 4) if valu_peak_meas else None,
                "frac_executed_at_measured_clock": round(valu_executed / valu_peak_meas, 4) if valu_executed and valu_peak_meas else None,
                "traffic_source": traffic_src,
                "note": "a register-tiled kernel moves ~8/R bytes per cell, so it sits far above the HBM contract roofline by construction; "
                        "vector-ALU issue is the limit that binds (achieved = algorithmic issue slots per cell x 64 lanes x measured cells/s)",
                # the contract figure of SURVEY.md 8d / BASELINE.md: 16 algorithmic bytes per cell against 8 TB/s
                "hbm": {"bound": "hbm", "achieved": round(hbm_achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(hbm_achieved / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_cell": BYTES_PER_CELL,
                        "traffic": traffic,
                        "traffic_frac_of_peak": (round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and kern_ms else None)}},
            "phases": phases,
            "gather_payload_bytes": (allq.n * 2 * TOPK * 4 * world if db_shard else (max(shard.shard_bounds(allq.n, world)[r + 1] - shard.shard_bounds(allq.n, world)[r] for r in range(world)) * (2 * TOPK + 1) * 4 * world)) if use_dist else 0,
            "host_affinity": affinity,
            "gather_crosscheck": gather_check,
            "speculative": {"queries": stats["speculative"], "rechecked": stats["rechecked"]},
            "routing": {k: stats[k] for k in ("tier_drift", "tier_m3", "tier_i16", "tier_i32", "pair_items")},
        }
        if db_shard:
            full_cells = float(CONFIGS3_QUERIES) * (cells_total / nq)
            full_cells *= CONFIGS3_TARGETS / float(n_targets_total)
            out["config"]["queries"] = nq
            out["config"]["targets_per_gpu"] = n_per
            out["config"]["database_tiles_per_gpu"] = len(db.tiles())
            out["config"]["time_boxed"] = (n_per < CONFIGS3_TARGETS // 8 or nq < CONFIGS3_QUERIES)
            out["config"]["extrapolated"] = {"full_config": f"{CONFIGS3_QUERIES} queries x {CONFIGS3_TARGETS} targets on 8 GPUs",
                                             "cells": full_cells, "wall_s_at_this_rate_per_gpu_x8": round(full_cells / (gcups / world * 8 * 1e9), 1)}

        # ---- outside the timed region: tie the number to correctness and to the int32 kernel ----
        (l_ids, l_sc, l_cnt), (g_ids, g_sc, g_cnt) = last
        if not args.no_selfcheck and args.steps and q_count:
            rng = np.random.default_rng(12345)
            ids_h, sc_h, cnt_h = l_ids.cpu().numpy(), l_sc.cpu().numpy(), l_cnt.cpu().numpy()
            pairs, got = [], []
            for qi in np.linspace(0, q_count - 1, num=min(8, q_count)).astype(int):     # every entry of 8 queries' top lists
                for j in range(int(cnt_h[qi])):
                    pairs.append((q_begin + int(qi), int(ids_h[qi, j]))); got.append(int(sc_h[qi, j]))
            n_list = len(pairs)
            if not db_shard:                                                            # + random entries of the score matrix
                rq = rng.integers(0, q_count, 2000)
                rt = rng.integers(0, db.n, 2000)
                vals = scores[torch.from_numpy(rq).to(dev), torch.from_numpy(rt).to(dev)].cpu().numpy()
                pairs += [(q_begin + int(a), int(b)) for a, b in zip(rq, rt)]
                got += [int(v) for v in vals]
                fetch = lambda tid: cfg["t_codes"][int(cfg["t_off"][tid]):int(cfg["t_off"][tid + 1])]
            else:
                rq = rng.integers(0, q_count, 2000)
                rt = rng.integers(0, db.n, 2000)
                vals = s4g.score_pairs(db, allq, scorer, rq.astype(np.uint32), rt.astype(np.uint32), route="wave")
                pairs += [(int(a), int(b)) for a, b in zip(rq, rt)]
                got += [int(v) for v in vals]
                off_h = offsets_d.cpu().numpy()
                fetch = lambda tid: codes_d[int(off_h[tid]):int(off_h[tid + 1])].cpu().numpy()
            bad, secs = oracle_selfcheck(q_codes, q_off, fetch, pairs, got, n_cpu)
            chk = {"pairs": len(pairs), "mismatches": bad, "top_list_entries": n_list, "random_pairs": len(pairs) - n_list,
                   "against": "oracle/sw_oracle.c (scalar CPU restatement)", "seconds": round(secs, 2)}
            if db_shard:            # every planted copy must lead the MERGED list of its query with the query's self-score
                M = s4gd.matrix_by_name("BLOSUM_62")
                gi, gs = g_ids.cpu().numpy(), g_sc.cpu().numpy()
                miss = 0
                for i in range(nq):
                    qc = q_codes[int(q_off[i]):int(q_off[i + 1])]
                    if gi[i, 0] != (i % world) * n_per + planted[i] or gs[i, 0] != int(M[qc, qc].sum()):
                        miss += 1
                chk["planted_copies"] = nq
                chk["planted_not_first"] = miss
                chk["mismatches"] += miss
            out["selfcheck"] = chk
            if chk["mismatches"]:
                log(f"SELFCHECK FAILED: {chk}")
        if world == 1 and not db_shard and not args.no_tiers and not args.force_i32 and args.steps:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            step(None, dict(force_i32=True))
            ms32 = []
            for _ in range(2):
                e0.record()
                s4g.scan(db, allq, scorer, scores, q_begin=q_begin, q_count=q_count, force_i32=True)
                e1.record()
                torch.cuda.synchronize()
                ms32.append(e0.elapsed_time(e1))
            k32 = float(np.mean(ms32))
            g32 = cells_rank / (k32 * 1e-3) / 1e9
            clk32 = s4gd.scan_clock(dev_index)
            mhz32 = clk32["shader_mhz"] if clk32["launches"] else 0.0
            peak32 = cus * 4 * LANES_PER_CLK_PER_SIMD * mhz32 * 1e6 / 1e12 if mhz32 else None
            slots32 = VALU_OPS_PER_CELL["sw_scan_i32d<32>"][0] - 0.5 * VALU_OPS_PER_CELL["sw_scan_i32d<32>"][1]
            out["tiers"] = {"int32": {"gcups": round(g32, 2), "kernel_ms": round(k32, 3), "kernel": "sw_scan_i32d<32>", "steps": 2,
                                      "valu_frac": round(g32 * slots32 / 1e3 / valu_peak, 4) if valu_peak else None,
                                      "shader_mhz_in_kernel": round(mhz32, 1),
                                      "valu_frac_at_measured_clock": round(g32 * slots32 / 1e3 / peak32, 4) if peak32 else None,
                                      "valu_frac_every_instruction_a_full_slot": round(g32 * VALU_OPS_PER_CELL["sw_scan_i32d<32>"][0] / 1e3 / valu_peak, 4) if valu_peak else None},
                            tier: {"gcups": round(kern_gcups, 2), "kernel_ms": round(kern_ms, 3), "kernel": plan["kernel"]}}
        if world == 1 and not db_shard and not args.no_end_to_end and args.steps and q_count:
            try:                                   # an extra leg must never cost the line the headline number
                out["end_to_end"] = end_to_end_leg(s4g, torch, db, q_codes, q_off, scorer, scores, dev, cells_rank)
                out["end_to_end"]["kernel_only_gcups"] = round(kern_gcups, 2)
            except Exception as exc:
                out["end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"}
                log(f"end_to_end leg failed: {exc}")
        if world == 1 and tail_wl and args.steps:
            # ---- what the tail costs: the same step on the same database WITHOUT its sequences beyond 12 288 residues (the tile of
            #      their own that s4g_db_create gives them, scored on a side stream), and on the set clipped at 3000 as SURVEY 8d has it ----
            try:
                out["tail"] = tail_leg(s4g, s4gd, synth, torch, cfg, allq, scorer, dev, nq, nt, qlen, wl, db, kern_ms, topk_ms, cells_rank, gcups)
            except Exception as exc:
                out["tail"] = {"error": f"{type(exc).__name__}: {exc}"}
                log(f"tail leg failed: {exc}")
        if world == 1 and not db_shard and args.workload == "auto" and not args.no_scale_ref and args.queries is None and args.targets is None:
            # ---- like-for-like denominator of the N > 1 lines: the workload they run (configs[2], strong-scaled) on this one GPU,
            #      two steps (scan + top-k, as a timed step) outside the timed region ----
            try:
                wl2 = WORKLOADS["configs2"]
                t_g = time.perf_counter()
                cfg2 = synth.make_config(wl2["cfg"], n_queries=wl2["queries"], n_targets=wl2["targets"], q_len=wl2["qlen"])
                t_g = time.perf_counter() - t_g
                del scores
                torch.cuda.empty_cache()
                db2 = s4g.Database(codes=cfg2["t_codes"], offsets=cfg2["t_off"], device=dev_index)
                q2 = s4g.QuerySet(codes=cfg2["q_codes"], offsets=cfg2["q_off"], device=dev_index)
                sc2 = torch.empty((q2.n, db2.n), dtype=torch.int32, device=dev)
                cells2 = s4gd.scan_cells(db2, q2, 0, q2.n)
                ms2 = []
                for it in range(3):                         # one warm-up (workspace allocation), two measured
                    torch.cuda.synchronize()
                    t_s = time.perf_counter()
                    s4g.scan(db2, q2, scorer, sc2)
                    s4g.topk_device(db2, sc2, TOPK, 1)
                    torch.cuda.synchronize()
                    if it:
                        ms2.append((time.perf_counter() - t_s) * 1e3)
                out["scale_ref"] = {"workload": f"{wl2['label']}: {q2.n} queries x {db2.n} targets on ONE GPU (what --gpus N > 1 strong-scales)",
                                    "gcups": round(cells2 / (float(np.mean(ms2)) * 1e-3) / 1e9, 2), "ms_per_step": round(float(np.mean(ms2)), 3),
                                    "steps": 2, "cells_per_step": float(cells2), "generate_s": round(t_g, 1),
                                    "shader_mhz_in_kernel": round(s4gd.scan_clock(dev_index)["shader_mhz"], 1)}
                del sc2, db2, q2, cfg2
            except Exception as exc:               # an extra leg must never cost the line the headline number
                out["scale_ref"] = {"error": f"{type(exc).__name__}: {exc}"}
                log(f"scale_ref leg failed: {exc}")
        if world == 1 and not db_shard and not args.no_prefilter and cfg is not None:
            try:                                   # an extra leg must never cost the line the headline number
                out["prefilter"] = prefilter_leg(cfg, dev_index)
                if out["prefilter"]["check"]["mismatches"]:
                    log(f"PREFILTER CHECK FAILED: {out['prefilter']['check']}")
            except Exception as exc:               # reported, not hidden: the leg's absence shows in the line
                out["prefilter"] = {"error": f"{type(exc).__name__}: {exc}"}
                log(f"prefilter leg failed: {exc}")
        if world == 1 and not args.no_cpu_baseline and cfg is not None:
            out["cpu_baseline"] = cpu_baseline(cfg, n_cpu)
        line = json.dumps(out)
    if gather_comm is not None:
        torch.cuda.synchronize()
        gather_comm.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # (the C stdout buffer -- RCCL's banner -- goes out first, to what is now stderr)
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        result_out.write(line + "\n")
        result_out.flush()
        if json.loads(line).get("selfcheck", {}).get("mismatches") or json.loads(line).get("prefilter", {}).get("check", {}).get("mismatches") \
                or (gather_check and not gather_check["c_abi_equals_torch_distributed_on_every_rank"]):
            sys.exit(3)
        # Figures that look wrong on THIS box are said on stderr, never turned into an exit code: the line stands as measured, and a
        # time-sliced or down-clocked GPU must not fail the run (GPUTEST_r05 died on a busy-fraction threshold, not on a mismatch).
        res = json.loads(line)
        rf = res["roofline"]
        if rf.get("executed") and rf.get("peak_at_measured_clock") and rf["executed"] > 1.01 * rf["peak_at_measured_clock"]:
            log(f"WARNING roofline inconsistent: executed {rf['executed']} T lane-ops/s above the issue peak {rf['peak_at_measured_clock']} at the clock measured in the kernel")
        ck = rf["clock_in_kernel"]
        if world == 1 and not res["emulated"] and res["config"]["cells_per_step"] >= 1e12 and ck["launches_sampled"]:
            if ck["blocks_busy_frac"] is not None and ck["blocks_busy_frac"] < 0.9:
                log(f"WARNING the persistent blocks were busy {ck['blocks_busy_frac']} of the longest one's time: another process on this GPU, or a queue that drains badly")
            if ck["launches_sampled"] == 1 and rf["kernel_ms"] > 50.0 and abs(ck["kernel_ms_by_100MHz_counter"] / rf["kernel_ms"] - 1.0) > 0.05:
                log(f"WARNING the kernel's own 100 MHz counter reads {ck['kernel_ms_by_100MHz_counter']} ms, the HIP events {rf['kernel_ms']} ms")
            if 0.0 < ck["shader_mhz"] < 1800.0:
                log(f"WARNING the scan kernel ran at {ck['shader_mhz']} MHz (guide peak {GUIDE_CLOCK_MHZ:.0f}): a throttled or power-capped box")

if __name__ == "__main__":
    main()
