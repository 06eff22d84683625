#!/usr/bin/env python3
"""profiles/<name>/ from the raw rocprofv3 files of tools/profile_prefilter.sh: kernel_stats.csv (kernel trace), the bench
output, and summary.json = per prefilter kernel the average duration and the FETCH_SIZE / WRITE_SIZE counters per call (KB, as
the counters report them; MI355X_MICROARCH.md: double FETCH_SIZE for wide coalesced reads, not for the random 4-byte lookups
that dominate here).  usage: tools/summarize_prefilter.py <prefix> <name> [git sha]"""
import collections, csv, glob, json, os, shutil, sys

prefix, name = sys.argv[1], sys.argv[2]
sha = sys.argv[3] if len(sys.argv) > 3 else None
dst = f"profiles/{name}"
os.makedirs(dst, exist_ok=True)
shutil.copy(sorted(glob.glob(f"gpurun_out/{prefix}_trace/runc/*_kernel_stats.csv"))[-1], f"{dst}/kernel_stats.csv")
lines = [l for l in open(f"gpurun_out/{prefix}_bench.txt") if not l.startswith(("W2026", "E2026")) and "amdgpu.ids" not in l]
open(f"{dst}/bench_output.txt", "w").writelines(lines)


def short(n):
    if "rocprim" in n:
        for key in ("radix_sort", "onesweep", "scan", "reduce_by_key", "histogram", "transform"):
            if key in n:
                return "rocprim:" + key
        return "rocprim:other"
    return n.split("(")[0].replace("s4g::", "")


stats = collections.defaultdict(lambda: dict(calls=0, total_us=0.0))
for r in csv.DictReader(open(f"{dst}/kernel_stats.csv")):
    s = stats[short(r["Name"])]
    s["calls"] += int(r["Calls"]); s["total_us"] += float(r["TotalDurationNs"]) / 1e3
pmc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/{prefix}_pmc/{cnt}/runc/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cnt:
                e = pmc[short(r["Kernel_Name"])][cnt]; e[0] += 1; e[1] += float(r["Counter_Value"])
out = dict(git_sha=sha, note="durations: rocprofv3 kernel trace of tools/prefilter_bench.py 1000 100000 5 5000 3 2000 (8 add calls of 100 000 "
           "targets, 4 finish calls, one add of 2000); FETCH/WRITE: KB per call from separate --pmc passes", kernels={})
for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["total_us"]):
    d = dict(calls=s["calls"], avg_us=round(s["total_us"] / s["calls"], 1), total_ms=round(s["total_us"] / 1e3, 3))
    for cnt, (n, v) in pmc.get(k, {}).items():
        d[cnt + "_KB_per_call"] = round(v / n, 1)
    out["kernels"][k] = d
json.dump(out, open(f"{dst}/summary.json", "w"), indent=1)
for k, d in out["kernels"].items():
    print(f"{k:28s} {d}")
