#!/usr/bin/env python3
"""profiles/<name>/ from the raw rocprofv3 files of tools/profile_lists.sh (or any run laid out the same way):
kernel_stats.csv (kernel trace), bench output, and pmc_summary.json = per kernel the summed counters of ONE dispatch (the
timed one: the last dispatch of each kernel in every pass) with a few derived figures.
usage: tools/summarize_pmc.py <prefix> <name> <cells of the workload> [git sha] [kernel substrings, comma separated]"""
import collections, csv, glob, json, os, shutil, sys

prefix, name, cells = sys.argv[1], sys.argv[2], float(sys.argv[3])
sha = sys.argv[4] if len(sys.argv) > 4 else None
want = sys.argv[5].split(",") if len(sys.argv) > 5 else ["sw_"]
dst = f"profiles/{name}"
os.makedirs(dst, exist_ok=True)
shutil.copy(glob.glob(f"gpurun_out/{prefix}_trace/runc/*_kernel_stats.csv")[0], f"{dst}/kernel_stats.csv")
if os.path.exists(f"gpurun_out/{prefix}_bench.txt"):
    shutil.copy(f"gpurun_out/{prefix}_bench.txt", f"{dst}/bench_output.txt")
avg_ms = {r["Name"]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(f"{dst}/kernel_stats.csv"))}
per = collections.defaultdict(dict)
meta = {}
for f in glob.glob(f"gpurun_out/{prefix}_pmc/*/runc/*_counter_collection.csv"):
    last = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(w in k for w in want):
            continue
        last.setdefault((k, r["Counter_Name"]), {})[r["Dispatch_Id"]] = float(r["Counter_Value"])
        meta[k] = dict(vgpr=r["VGPR_Count"], lds_bytes=r["LDS_Block_Size"], scratch=r["Scratch_Size"])
    for (k, c), by_dispatch in last.items():
        per[k][c] = by_dispatch[max(by_dispatch, key=int)]          # the last (timed) dispatch of the pass
out = dict(git_sha=sha, workload_cells=cells, note="counters of the last dispatch of each kernel in each pass; SQ_* cycle counters are in quad-cycles "
           "(MI355X_MICROARCH.md); FETCH_SIZE in KB, x2 for wide coalesced reads", kernels={})
for k, c in per.items():
    ms = avg_ms.get(k)
    d = dict(counters=c, kernel_ms_rocprof_avg=ms, **meta[k])
    if ms and "SQ_INSTS_VALU" in c:
        d["valu_T_lane_ops_per_s"] = c["SQ_INSTS_VALU"] * 64 / (ms * 1e-3) / 1e12
        d["valu_lane_ops_per_workload_cell"] = c["SQ_INSTS_VALU"] * 64 / cells
    if "GRBM_GUI_ACTIVE" in c and ms:
        d["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_WAVE_CYCLES" in c:
        d["lds_bank_conflict_frac_of_wave_cycles"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_WAVE_CYCLES"]
    if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
        d["wait_inst_any_frac_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    out["kernels"][k] = d
json.dump(out, open(f"{dst}/pmc_summary.json", "w"), indent=1)
for k, d in out["kernels"].items():
    print(k[:60], {x: round(v, 3) for x, v in d.items() if isinstance(v, float)})
