"""bench.py contract: one JSON line with the keys the driver reads (GPU), and a stable synthetic workload (CPU)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_synthetic_workload_is_deterministic():
    """the bench workload must not drift between runs, machines or rounds: checksum of a small instance of the generator."""
    from sift4g_amd import synth
    c = synth.make_config(2, n_queries=6, n_targets=400, homologs_per_query=5)
    # digest over the parts produced by integer / exactly-rounded arithmetic only (queries, planted homologs); the background
    # target lengths go through libm's exp/log/cos, whose last ulp may differ between CPUs
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(c["q_codes"]).tobytes()); h.update(np.ascontiguousarray(c["q_off"]).tobytes())
    for _, t in c["planted"]:
        h.update(np.ascontiguousarray(c["t_codes"][int(c["t_off"][t]):int(c["t_off"][t + 1])]).tobytes())
    assert len(c["planted"]) == 30
    ql = np.diff(c["q_off"].astype(np.int64)); tl = np.diff(c["t_off"].astype(np.int64))
    assert ql.min() >= 300 and ql.max() <= 400 and tl.min() >= 30 and tl.max() <= 3000
    assert set(np.unique(c["t_codes"][: int(c["t_off"][-1])]).tolist()) <= set(np.nonzero(synth.AA_FREQUENCY)[0].tolist())
    c2 = synth.make_config(2, n_queries=6, n_targets=400, homologs_per_query=5)
    assert all(np.array_equal(c[k], c2[k]) for k in ("q_codes", "q_off", "t_codes", "t_off"))
    assert h.hexdigest() == "79275571666105ecacc64341f82cf4ff6b415edd831295a851524caccc517dac"


def test_swissprot_tail_generator():
    """synth.SWISSPROT_TAIL (bench.py --workload swissprot-tail): the clipped set's queries and planted homologs unchanged, the
    lognormal unclipped, a thin 2-10 k component, eight giants of fixed lengths at ids no homolog uses, each holding a copy of a query."""
    from sift4g_amd import synth
    base = synth.make_config(3, n_queries=40, n_targets=20000)
    tail = synth.make_config(3, n_queries=40, n_targets=20000, tail=synth.SWISSPROT_TAIL)
    assert np.array_equal(base["q_codes"], tail["q_codes"]) and np.array_equal(base["q_off"], tail["q_off"])
    assert tail["planted"][:len(base["planted"])] == base["planted"]
    lens = np.diff(tail["t_off"].astype(np.int64))
    giants = tail["giants"]
    assert [int(lens[t]) for t in giants] == list(synth.SWISSPROT_TAIL["giants"]) and len(set(giants)) == 8
    assert not set(giants) & {t for _, t in base["planted"]}
    assert int(lens.max()) == 35213 and int((lens > 3000).sum()) >= 8 and np.diff(base["t_off"].astype(np.int64)).max() <= 3000
    for q, t in tail["planted"][len(base["planted"]):]:          # the copies inside the giants
        assert t in giants
        qc = tail["q_codes"][int(tail["q_off"][q]):int(tail["q_off"][q + 1])]
        tc = tail["t_codes"][int(tail["t_off"][t]):int(tail["t_off"][t + 1])]
        assert any(np.array_equal(tc[i:i + len(qc)], qc) for i in np.nonzero(tc[:len(tc) - len(qc) + 1] == qc[0])[0])
    again = synth.make_config(3, n_queries=40, n_targets=20000, tail=synth.SWISSPROT_TAIL)
    assert all(np.array_equal(tail[k], again[k]) for k in ("q_codes", "q_off", "t_codes", "t_off")) and again["giants"] == giants


def _run_bench(*argv, env=None, timeout=1500):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def _check_common(d, n_gpus, steps, warmup):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "rccl_ranks", "speculative", "routing"):
        assert key in d, key
    assert d["unit"] == "GCUPS" and d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert d["rccl_ranks"] == n_gpus
    rf = d["roofline"]
    # the limit that binds is vector-ALU issue, priced at the clock measured in the same run; the HBM contract figure of
    # SURVEY.md 8d (16 B per cell against 8 TB/s) rides along
    assert rf["bound"] == "valu" and rf["unit"] == "T lane-ops/s" and 20.0 < rf["peak"] < 45.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["achieved"] > 0
    # `peak` is priced at the guide's 2.4 GHz; the clock the timed kernel ran at is sampled inside it and prices the second peak
    assert "2400 MHz" in rf["peak_basis"] and rf["frac_algorithmic"] == rf["frac"] and rf["traffic_measured_in_run"] is False
    # round 4: what the arithmetic type guarantees is in `dtype`; the two issue models sit side by side; stored counter figures are
    # keyed on the hash of the kernel sources (and are null whenever that hash is not the profiled build's)
    assert d["dtype"] == "int32" or d["dtype"].endswith("int32 out")
    assert rf["frac_every_instruction_a_full_slot"] >= rf["frac"] and "frac_model" in rf
    from sift4g_amd.srchash import kernel_source_sha
    assert rf["kernel_source_sha"] == kernel_source_sha()
    if rf["traffic"] is not None:
        import json as _json
        assert _json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))["kernel_source_sha"] == rf["kernel_source_sha"]
    ck = rf["clock_in_kernel"]
    if ck["launches_sampled"]:           # a call that went to the pair waves as a whole launches no scan kernel
        # ORDER and SIGN only: what the clock, the busy fraction or the two timers read depends on the box and on who else is on the GPU
        # (GPUTEST_r05: two ranks time-slicing one device read blocks_busy_frac 0.456); bench.py prints a warning on stderr where a
        # figure looks wrong on a full-size single-GPU run, a correctness suite never fails on one
        assert ck["blocks"] >= 1 and 0.0 < ck["min_block"] <= ck["max_block"]
        if rf["peak_at_measured_clock"]:          # (null when the sampled clock reads below 500 MHz: bench.py does not price a peak at such a reading)
            assert ck["min_block"] <= ck["shader_mhz"] <= ck["max_block"]
            assert abs(rf["peak_at_measured_clock"] / rf["peak"] - ck["shader_mhz"] / 2400.0) < 1e-3
        if ck["blocks_busy_frac"] is not None:
            assert 0.0 < ck["blocks_busy_frac"] <= 1.0001 and ck["block_busy_ms"]["min"] <= ck["block_busy_ms"]["mean"] <= ck["kernel_ms_by_100MHz_counter"] * 1.0001
    for ph in ("scan_ms", "topk_ms", "gather_ms"):
        assert d["phases"][ph]["max"] >= d["phases"][ph]["min"] >= 0.0
    hb = rf["hbm"]
    assert hb["bound"] == "hbm" and hb["unit"] == "GB/s" and hb["peak"] == 8000.0 and abs(hb["frac"] - hb["achieved"] / hb["peak"]) < 1e-3
    if "selfcheck" in d:
        assert d["selfcheck"]["mismatches"] == 0 and d["selfcheck"]["pairs"] > 0, d["selfcheck"]


@pytest.mark.gpu
def test_bench_prints_one_contract_json_line():
    d = _run_bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--queries", "16", "--targets", "3000")
    _check_common(d, 1, 2, 1)
    assert "configs[1]" in d["config"]["workload"] and d["scaling"] == "strong"
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "GCUPS" and cb["sample"]
    # what bounds the baseline on the box is IN the line (presence only): cgroup quota or null, the affinity mask, whether threads were pinned
    assert "quota_cpus" in cb and isinstance(cb["pinned"], bool) and cb["limits"]["bound_by"] and cb["limits"]["affinity_cpus"] >= 1
    # the end-to-end figure rides along (presence only: queries H2D + scan + top-k + results D2H, lists and full matrix)
    e2e = d["end_to_end"]
    assert e2e["top_lists"]["ms"] > 0 and e2e["full_matrix"]["ms"] > 0 and e2e["full_matrix"]["d2h_bytes"] == 16 * 3000 * 4
    assert set(e2e["parts_ms"]) == {"queries_h2d_ms", "scan_ms", "topk_ms", "lists_d2h_ms", "matrix_d2h_ms"}
    # the same line carries the int32-only figure (north_star: "int32 scores") and the oracle re-score of the run's own results
    assert d["tiers"]["int32"]["gcups"] > 0 and d["tiers"]["int32"]["kernel"] == "sw_scan_i32d<32>"
    chk = d["selfcheck"]
    assert chk["random_pairs"] == 2000 and chk["top_list_entries"] > 0 and chk["mismatches"] == 0
    assert d["speculative"] == {"queries": 0, "rechecked": 0}
    # ... and the stage in front of the scan, checked against its own restatement
    pf = d["prefilter"]
    assert pf["targets"] == 3000 and pf["residues_per_s"] > 0 and pf["kmer_hits"] >= pf["pairs_with_a_hit"] > 0
    assert pf["check"] == dict(pf["check"], targets=2000, mismatches=0)


@pytest.mark.gpu
def test_bench_swissprot_tail_workload_line():
    """--workload swissprot-tail at test size: the line names the workload, carries the `tail` leg (length statistics, the scan without
    the sequences beyond 12 288 residues, the clipped set, their ratio) and the oracle re-score of its own results."""
    d = _run_bench("--gpus", "1", "--workload", "swissprot-tail", "--steps", "1", "--warmup", "1", "--queries", "24", "--targets", "30000",
                   "--no-cpu-baseline", "--no-tiers", "--no-prefilter")
    _check_common(d, 1, 1, 1)
    assert "unclipped Swiss-Prot-like length tail" in d["config"]["workload"] and "giants" in d["config"]["workload"]
    t = d["tail"]
    assert t["lengths"]["max"] == 35213 and t["lengths"]["over_12288"] >= 6 and t["giants"] == [10200, 12288, 12289, 14507, 18141, 22152, 27118, 35213]
    assert len(t["tile_seqs"]) >= 2 and t["long_tile"]["cells"] > 0 and t["ratio_to_clipped"] > 0
    assert t["clipped_at_3000"]["gcups"] > 0 and t["without_sequences_over_12288"]["gcups"] > 0
    assert d["selfcheck"]["mismatches"] == 0


@pytest.mark.gpu
def test_bench_launches_its_own_ranks_and_strong_scales_configs2():
    """`python bench.py --gpus 2` as a plain command: the parent spawns the ranks before touching the GPU and relays rank 0's
    line.  On the one-GPU test box the two ranks share the device (S4G_BENCH_SHARE_GPU=1: gloo instead of RCCL, reported as
    "emulated"); the job itself -- configs[2] generator, queries split over the ranks, gather of the hit lists -- is the real one."""
    d = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--queries", "24", "--targets", "6000", "--no-cpu-baseline",
                   env={"S4G_BENCH_SHARE_GPU": "1"})
    _check_common(d, 2, 2, 1)
    assert d["scaling"] == "strong" and "configs[2]" in d["config"]["workload"] and "query-shard x2" in d["config"]["parallelism"]
    import torch
    assert d["emulated"] == (torch.cuda.device_count() < 2)
    one = _run_bench("--gpus", "1", "--workload", "configs2", "--steps", "1", "--warmup", "0", "--queries", "24", "--targets", "6000",
                     "--no-cpu-baseline", "--no-tiers")
    assert one["config"]["cells_per_step"] == d["config"]["cells_per_step"]        # total work is the same at N = 1 and N = 2


@pytest.mark.gpu
def test_bench_db_shard_mode_device_generated_shards():
    """configs[3] shape at test size: a shard generated on each device, streaming scan + running top-k, all-gather + K-way merge;
    the line carries the oracle re-score of rank 0's lists and the planted-copy check on the MERGED lists."""
    for n, env in ((1, {}), (2, {"S4G_BENCH_SHARE_GPU": "1"})):
        d = _run_bench("--gpus", str(n), "--mode", "db-shard", "--steps", "1", "--warmup", "1", "--queries", "6", "--targets", "40000",
                       "--no-cpu-baseline", env=env)
        _check_common(d, n, 1, 1)
        assert d["scaling"] == "weak" and "configs[3]" in d["config"]["workload"]
        chk = d["selfcheck"]
        assert chk["planted_copies"] == 6 and chk["planted_not_first"] == 0 and chk["mismatches"] == 0
        assert d["config"]["extrapolated"]["wall_s_at_this_rate_per_gpu_x8"] > 0


@pytest.mark.gpu
def test_bench_rccl_path_with_one_rank():
    """The collectives of the N > 1 job on the backend the driver's 8-GPU run uses -- RCCL ("nccl"), device tensors -- exercised
    with a process group of ONE rank on this box's GPU (S4G_BENCH_FORCE_DIST=1): init with a device id, the all-reduce that
    counts the ranks, the all-gather of the packed hit lists and the merge, in both sharding modes.  (The emulated N-rank tests
    above run the same job over gloo; two ranks cannot share a device under RCCL.)"""
    env = {"S4G_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"}
    d = _run_bench("--gpus", "1", "--workload", "configs2", "--steps", "2", "--warmup", "1", "--queries", "20", "--targets", "5000",
                   "--no-cpu-baseline", "--no-tiers", "--no-prefilter", "--no-scale-ref", env=env)
    _check_common(d, 1, 2, 1)
    assert d["backend"] == "RCCL (nccl)" and d["rccl_ranks"] == 1 and not d["emulated"]
    # the gather below Python against the torch.distributed form of the same exchange, on the lists of the last step
    assert d["gather_crosscheck"] == {"c_abi_equals_torch_distributed_on_every_rank": True, "lists": 20}
    assert d["selfcheck"]["mismatches"] == 0 and d["selfcheck"]["top_list_entries"] > 0
    d = _run_bench("--gpus", "1", "--mode", "db-shard", "--steps", "1", "--warmup", "1", "--queries", "6", "--targets", "30000",
                   "--no-cpu-baseline", env=env)
    _check_common(d, 1, 1, 1)
    assert d["backend"] == "RCCL (nccl)" and d["selfcheck"]["planted_not_first"] == 0 and d["selfcheck"]["mismatches"] == 0
    assert d["gather_crosscheck"]["c_abi_equals_torch_distributed_on_every_rank"] is True


@pytest.mark.gpu
def test_bench_eight_rank_rehearsal_on_one_gpu():
    """The job the driver launches on an 8-GPU node, rehearsed with its eight ranks sharing this box's GPU(s): a query count that
    does not divide by 8 (ragged shards), per-phase timings and the all-gather payload in the line, self-check of rank 0's lists;
    then the same for --mode db-shard.  Emulated (gloo) unless the box really has eight GPUs; never a scaling measurement."""
    d = _run_bench("--gpus", "8", "--steps", "2", "--warmup", "1", "--queries", "45", "--targets", "5000", "--no-cpu-baseline",
                   env={"S4G_BENCH_SHARE_GPU": "1"})
    _check_common(d, 8, 2, 1)
    assert "query-shard x8" in d["config"]["parallelism"] and d["scaling"] == "strong"
    # 45 queries over 8 ranks: shards of 5 or 6 queries; every rank pads its lists to the longest shard for the all-gather
    assert d["gather_payload_bytes"] == 6 * (2 * 400 + 1) * 4 * 8
    assert d["phases"]["scan_ms"]["max"] > 0 and d["phases"]["gather_ms"]["max"] > 0
    assert d["selfcheck"]["mismatches"] == 0 and d["selfcheck"]["top_list_entries"] > 0
    one = _run_bench("--gpus", "1", "--workload", "configs2", "--steps", "1", "--warmup", "0", "--queries", "45", "--targets", "5000",
                     "--no-cpu-baseline", "--no-tiers", "--no-prefilter")
    assert one["config"]["cells_per_step"] == d["config"]["cells_per_step"]
    d = _run_bench("--gpus", "8", "--mode", "db-shard", "--steps", "1", "--warmup", "1", "--queries", "11", "--targets", "12000",
                   "--no-cpu-baseline", env={"S4G_BENCH_SHARE_GPU": "1"})
    _check_common(d, 8, 1, 1)
    assert d["scaling"] == "weak" and "db-shard x8" in d["config"]["parallelism"]
    assert d["gather_payload_bytes"] == 11 * 2 * 400 * 4 * 8
    chk = d["selfcheck"]
    assert chk["planted_copies"] == 11 and chk["planted_not_first"] == 0 and chk["mismatches"] == 0


def test_cpu_baseline_says_what_bounds_it(tmp_path, monkeypatch):
    """bench.py's cpu_baseline helpers on this host: the core map and the spread masks (one hardware thread per physical core, evenly
    over the core list, siblings only beyond the cores), the cgroup quota reader on a synthetic cpu.max, and the baseline itself on a toy
    workload -- `limits.bound_by`, `quota_cpus`, `pinned` are in the result and the affinity mask is back to what it was."""
    sys.path.insert(0, ROOT)
    import bench
    cores = bench._core_map()
    assert cores and all(c == sorted(c) for c in cores)
    for n in (1, 2, len(cores)):
        m = bench._spread_mask(cores, n)
        assert len(m) == n and all(sum(1 for x in c if x in m) <= 1 for c in cores)       # never two siblings of one core while n fits the cores
    fake = [[0, 8], [1, 9], [2, 10], [3, 11]]
    assert bench._spread_mask(fake, 2) == {0, 2} and bench._spread_mask(fake, 4) == {0, 1, 2, 3} and bench._spread_mask(fake, 6) == {0, 1, 2, 3, 8, 9}
    lim = bench._cpu_limits()
    assert lim["affinity_cpus"] == len(os.sched_getaffinity(0)) and "quota_cpus" in lim
    before = os.sched_getaffinity(0)
    from sift4g_amd import synth
    cfg = synth.make_config(2, n_queries=8, n_targets=600, homologs_per_query=4)
    cb = bench.cpu_baseline(cfg, len(before), target_seconds=0.5)
    assert os.sched_getaffinity(0) == before
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["limits"]["bound_by"] and "quota_cpus" in cb and isinstance(cb["pinned"], bool)
    assert cb["limits"]["bound_by"] in cb["sample"]


def test_bench_refuses_more_ranks_than_devices_in_seconds():
    """`python bench.py --gpus 8` where fewer than 8 GPUs are visible (this container: none; the 1-GPU test box: one): exit code 2 and the
    reason on stderr before a single rank is spawned -- not a hang, not a JSON line.  (Runs on CPU and on the GPU box alike.)"""
    import time
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("eight GPUs visible: the command would run the real job")
    env = {k: v for k, v in os.environ.items() if k not in ("S4G_BENCH_SHARE_GPU", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 2 and "8 ranks but" in r.stderr and "GPU(s) visible" in r.stderr and not r.stdout.strip(), (r.returncode, r.stdout[-300:], r.stderr[-800:])
    assert "spawning" not in r.stderr                      # refused in the parent: no torch.distributed.run children
    print(f"refused in {time.perf_counter() - t0:.1f} s")


def test_numa_pinning_reads_a_kfd_topology(tmp_path):
    """sift4g_amd/numa.py on a synthetic sysfs tree: two CPU nodes, four GPUs (two per socket), io_links to the socket."""
    from sift4g_amd import numa
    kfd, nodes = tmp_path / "kfd", tmp_path / "node"
    for i in range(2):
        (kfd / str(i)).mkdir(parents=True)
        (kfd / str(i) / "properties").write_text("cpu_cores_count 4\nsimd_count 0\n")
        (nodes / f"node{i}").mkdir(parents=True)
        (nodes / f"node{i}" / "cpulist").write_text("0-3\n" if i == 0 else "4-7\n")
    for g in range(4):
        d = kfd / str(2 + g)
        (d / "io_links" / "0").mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\ndomain 0\nlocation_id 256\n")
        (d / "io_links" / "0" / "properties").write_text(f"type 2\nnode_from {2 + g}\nnode_to {g // 2}\n")
    assert numa.gpu_numa_nodes(str(kfd)) == [0, 0, 1, 1]
    assert numa._parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
    before = os.sched_getaffinity(0)
    try:
        info = numa.pin_to_gpu(2, str(kfd), str(nodes))
        if before & {4, 5, 6, 7}:
            assert info["pinned"] and info["numa_node"] == 1 and os.sched_getaffinity(0) == before & {4, 5, 6, 7}
        else:
            assert not info["pinned"] and "affinity" in info["reason"]
        assert numa.pin_to_gpu(9, str(kfd), str(nodes))["pinned"] is False
    finally:
        os.sched_setaffinity(0, before)
