"""Copy what is judged out of the rocprofv3 output of tools/profile_bench.sh into profiles/.

  python tools/summarize_prof.py <tag> <name>     e.g.  r1e r01_e
    gpurun_out/prof_<tag>/**/_kernel_stats.csv  -> profiles/<name>_bench_default_kernel_stats.csv
    gpurun_out/prof_<tag>_bench.log             -> profiles/<name>_bench_default.log
    gpurun_out/pmc_<tag>_{FETCH,WRITE}_SIZE/**/_counter_collection.csv
        -> profiles/<name>_pmc_fetch_write_summary.csv (per kernel: launches, raw KiB) and
           profiles/<name>_pmc_traffic.json (bytes per launch = raw x 1024, as MI355X_MICROARCH.md prescribes)
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")

ks = glob.glob(os.path.join(G, "prof_" + tag, "**", "*_kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(P, name + "_bench_default_kernel_stats.csv"))
    print("kernel stats:", ks[0])
log = os.path.join(G, "prof_%s_bench.log" % tag)
if os.path.exists(log):
    shutil.copy(log, os.path.join(P, name + "_bench_default.log"))


def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k)
    # the search kernel has one instantiation per admission rule: <1> = ordered (the product's default) keeps the plain name, <0> = exact
    m = re.match(r"decode_kernel<(\d)>", k)
    if m:
        return "decode_kernel" if m.group(1) == "1" else "decode_kernel<exact admission>"
    return re.split(r"[<(]", k)[0]


rows, per, per_max = [], {}, {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(G, "pmc_%s_%s" % (tag, c), "**", "*_counter_collection.csv"), recursive=True)
    if not files:
        continue
    tot, n = defaultdict(float), defaultdict(set)
    by_dispatch = defaultdict(float)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != c:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
        by_dispatch[(k, r["Dispatch_Id"])] += float(r["Counter_Value"])
    per[c] = {}
    mx = defaultdict(float)
    for (k, _d), v in by_dispatch.items():
        mx[k] = max(mx[k], v)
    per_max[c] = {k: v * 1024.0 for k, v in mx.items()}
    for k in tot:
        rows.append((c, k, len(n[k]), tot[k], tot[k] / len(n[k])))
        per[c][k] = tot[k] / len(n[k]) * 1024.0
if rows:
    with open(os.path.join(P, name + "_pmc_fetch_write_summary.csv"), "w") as f:
        f.write("counter,kernel,launches,sum_raw_KiB,per_launch_raw_KiB\n")
        for r in rows:
            f.write("%s,%s,%d,%.1f,%.1f\n" % r)
    fe, wr = per.get("FETCH_SIZE", {}), per.get("WRITE_SIZE", {})
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-step-api --no-determinize",
           "batch": 1024, "source": "profiles/%s_pmc_fetch_write_summary.csv (FETCH_SIZE + WRITE_SIZE, raw KiB x 1024)" % name,
           "note": "raw counter values x 1024 B; MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced streaming reads by 2x "
                   "and is uncalibrated for scattered access",
           "bytes_per_launch": {k: fe.get(k, 0.0) + wr.get(k, 0.0) for k in set(fe) | set(wr)},
           # the largest launch of every kernel (the front-end kernels also run on the 64-utterance training batches: their mean says little)
           "bytes_of_largest_launch": {k: per_max.get("FETCH_SIZE", {}).get(k, 0.0) + per_max.get("WRITE_SIZE", {}).get(k, 0.0) for k in set(fe) | set(wr)},
           "fetch_bytes_per_launch": fe, "write_bytes_per_launch": wr}
    json.dump(out, open(os.path.join(P, name + "_pmc_traffic.json"), "w"), indent=1)
    print("traffic: decode_kernel %.1f GB/launch" % (out["bytes_per_launch"].get("decode_kernel", 0) / 1e9))
