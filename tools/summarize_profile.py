#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of one kernel (a --kernel-trace --stats run and three --pmc passes under gpurun_out/<prefix>_*)
into profiles/<name>/{kernel_stats.csv, bench_under_rocprof.json, pmc_summary.json} and profiles/hbm_traffic.json.
usage: tools/summarize_profile.py <prefix> <name> <kernel label> [git sha of the build that was profiled]
(the raw files come from tools/profile_bench.sh <prefix> ..., run on the GPU box through gpurun)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sift4g_amd.srchash import kernel_source_sha          # the sources the profiled build was made from (run this on the same tree)

prefix, name, label = sys.argv[1:4]
src_sha = kernel_source_sha()
sha = sys.argv[4] if len(sys.argv) > 4 else None
cells = 12076509042144          # BASELINE configs[1]: sum Lq * sum Lt of the synthetic workload
key = label.split("<")[0]        # kernel symbol to pick out of the traces ("sw_scan_pkd", "sw_scan_i32", ...)
dst = f"profiles/{name}"
os.makedirs(dst, exist_ok=True)
shutil.copy(glob.glob(f"gpurun_out/{prefix}_trace/runc/*_kernel_stats.csv")[0], f"{dst}/kernel_stats.csv")
shutil.copy(f"gpurun_out/{prefix}_bench_trace.txt", f"{dst}/bench_under_rocprof.json")
out, meta = {}, {}
for d, tag in ((f"{prefix}_fetch", "fetch"), (f"{prefix}_write", "write"), (f"{prefix}_sq", "sq")):
    f = glob.glob(f"gpurun_out/{d}/runc/*_counter_collection.csv")[0]
    agg, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if key + "<" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            meta = dict(kernel_symbol=r["Kernel_Name"], vgpr=r["VGPR_Count"], lds_bytes=r["LDS_Block_Size"], scratch=r["Scratch_Size"])
    # PER LAUNCH: a pass may dispatch the kernel more than once on the same workload (round 6: bench.py's end_to_end leg adds four scans
    # unless --no-end-to-end is given); every figure below is the mean over the pass's dispatches, "all_dispatches" keeps the raw sum
    out[tag] = {k: {"sum": v / n[k], "dispatches": n[k], "all_dispatches": v} for k, v in agg.items()}
fetch_kb, write_kb = out["fetch"]["FETCH_SIZE"]["sum"], out["write"]["WRITE_SIZE"]["sum"]
avg_ns = [float(r["AverageNs"]) for r in csv.DictReader(open(f"{dst}/kernel_stats.csv")) if key + "<" in r["Name"]][0]
t = avg_ns / 1e9
sq = out["sq"]
s = dict(kernel=label, workload_cells=cells, git_sha=sha, kernel_source_sha=src_sha, **meta,
         command="tools/profile_bench.sh: rocprofv3 --pmc <COUNTER SET> --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                 "--no-cpu-baseline --no-selfcheck --no-tiers (one pass per counter set: FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_WAVES "
                 "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES); rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 "
                 "--warmup 1 in its own run",
         FETCH_SIZE_KB=fetch_kb, WRITE_SIZE_KB=write_kb,
         correction="MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 64 B per 128-B request on wide (16 B/lane) coalesced reads -> read "
                    "bytes = FETCH_SIZE*1024*2; WRITE_SIZE*1024 exact for 16 B/lane stores. Strip-boundary rows are the kernel's only bulk "
                    "traffic: 16 B/lane vectors in sw_scan_pkd / sw_scan_i32d, 4 B/lane dwords (one whole 256-byte line per wave instruction, "
                    "two 128-B requests) in sw_scan_pkc -- the same bytes per chunk either way.",
         hbm_read_bytes_per_launch=fetch_kb * 2048, hbm_write_bytes_per_launch=write_kb * 1024,
         hbm_bytes_per_launch=fetch_kb * 2048 + write_kb * 1024, kernel_ms_rocprof_avg=avg_ns / 1e6, sq=sq)
s["derived"] = dict(
    kernel_gcups=cells / t / 1e9, hbm_GBps=s["hbm_bytes_per_launch"] / t / 1e9, hbm_bytes_per_cell=s["hbm_bytes_per_launch"] / cells,
    valu_lane_ops_per_cell=sq["SQ_INSTS_VALU"]["sum"] * 64 / cells, valu_T_lane_ops_per_s=sq["SQ_INSTS_VALU"]["sum"] * 64 / t / 1e12,
    clock_GHz=sq["GRBM_GUI_ACTIVE"]["sum"] / 8 / t / 1e9,
    cycles_per_valu_wave_instr_per_simd=(sq["GRBM_GUI_ACTIVE"]["sum"] / 8) / (sq["SQ_INSTS_VALU"]["sum"] / 1024),
    lds_bank_conflict_cycles_frac_of_wave_cycles=sq["SQ_LDS_BANK_CONFLICT"]["sum"] / sq["SQ_WAVE_CYCLES"]["sum"])
json.dump(s, open(f"{dst}/pmc_summary.json", "w"), indent=1)
if label.startswith("sw_scan_pkc"):        # the default kernel: what bench.py copies into roofline.traffic
    json.dump(dict(kernel=label, workload_cells=cells, hbm_bytes_per_launch=s["hbm_bytes_per_launch"], source=f"{dst}/pmc_summary.json",
                   git_sha=sha, kernel_source_sha=src_sha), open("profiles/hbm_traffic.json", "w"), indent=1)
# executed vector lane-ops per cell of every profiled scan kernel: what bench.py's roofline.frac_executed is priced with
ve_path = "profiles/valu_executed.json"
ve = json.load(open(ve_path)) if os.path.exists(ve_path) else {}
ve[label] = dict(lane_ops_per_cell=s["derived"]["valu_lane_ops_per_cell"], workload_cells=cells, SQ_INSTS_VALU=sq["SQ_INSTS_VALU"]["sum"],
                 source=f"{dst}/pmc_summary.json", git_sha=sha, kernel_source_sha=src_sha)
json.dump(ve, open(ve_path, "w"), indent=1)
print(json.dumps(s["derived"], indent=1))
