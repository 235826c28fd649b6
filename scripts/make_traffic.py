"""profiles/<round>_traffic.json + profiles/<round>_<workload>_kernel_stats.csv from the rocprofv3 outputs of one profiling round.

    python scripts/make_traffic.py <gpurun_out/prof_dir>

expects, per workload W: <dir>/W_stats (--kernel-trace --stats), <dir>/W_fetch (--pmc FETCH_SIZE), <dir>/W_write
(--pmc WRITE_SIZE), <dir>/W_bench.json (the bench line of an unprofiled run of the same command).  FETCH_SIZE /
WRITE_SIZE are in KB (MI355X_MICROARCH.md, HBM section); FETCH_SIZE is doubled on gfx950 as that guide says (128-B requests
tallied at 64 B) -- calibrated there for wide streaming reads only, so for 8/16-byte scattered loads it is an upper bound.
"""
import sys, os, glob, csv, json
d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = os.environ.get("PF_ROUND", "r04")
KERN = {"mpa512": "k_mpa_search", "maaco512": "k_maaco_walk8", "maaco1024": "k_maaco_walk8", "maaco128": "k_maaco_walk(",
        "ga512": "k_decode_batch", "pso512": "k_decode_batch", "astar1024": "k_astar_batch"}
try:
    out = json.load(open(os.path.join(ROOT, "profiles", f"{RND}_traffic.json")))
except Exception:
    out = {}


def match(name, kern):
    return kern in name


def pmc(dirname, counter, kern, last):
    """Sum of `counter` over the LAST `last` dispatches of the kernel (the timed region of the same command) -> (sum, n)."""
    per = {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if match(r["Kernel_Name"], kern) and r["Counter_Name"] == counter:
                per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    ids = sorted(per)[-last:] if last else sorted(per)
    return sum(per[i] for i in ids), len(ids)


def timed_avg_ms(d, w, kern, last):
    """Average duration of the last `last` dispatches of the kernel in the --kernel-trace run (= the timed region)."""
    f = os.path.join(d, w + "_dominant_trace.csv")
    if not os.path.exists(f):
        return None
    rows = [r for r in csv.DictReader(open(f)) if match(r["Kernel_Name"], kern)]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-last:] if last else rows
    return sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / max(len(rows), 1) / 1e6


for w, k in KERN.items():
    st = glob.glob(os.path.join(d, w + "_stats", "**", "*kernel_stats.csv"), recursive=True)
    if not st:
        continue
    rows = [r for r in csv.DictReader(open(st[0])) if match(r["Name"], k)]
    if not rows:
        continue
    calls = sum(int(r["Calls"]) for r in rows); tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    bench = json.loads(open(os.path.join(d, w + "_bench.json")).read().strip().splitlines()[-1])
    roof = bench["roofline"]
    last = int(roof.get("launches", 0))                     # launches of the timed region: the last ones of the process
    fetch, nf = pmc(os.path.join(d, w + "_fetch"), "FETCH_SIZE", k, last)
    write, nw = pmc(os.path.join(d, w + "_write"), "WRITE_SIZE", k, last)
    fb, wb = fetch * 1024 / max(nf, 1), write * 1024 / max(nw, 1)
    tavg = timed_avg_ms(d, w, k, last)
    # the --stats summary covers warm-up and set-up launches too; avg_ms_rocprof is over the timed region's dispatches
    out[w] = {"kernel": roof["kernel"], "calls_all": calls, "avg_ms_rocprof_all_launches": tot_ns / max(calls, 1) / 1e6,
              "timed_launches": last, "avg_ms_rocprof": tavg if tavg is not None else tot_ns / max(calls, 1) / 1e6,
              "bench_avg_launch_ms": roof["avg_launch_ms"], "bench_value": bench["value"], "bench_ms_per_step": bench["ms_per_step"],
              "FETCH_SIZE_bytes_per_launch_raw": fb, "WRITE_SIZE_bytes_per_launch": wb,
              "traffic_bytes_per_launch": 2 * fb + wb,
              "note": "FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); uncalibrated for 8/16-B scattered loads; separate --pmc passes",
              "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"],
              "traffic_over_algorithmic": (2 * fb + wb) / max(roof["algorithmic_bytes_per_launch"], 1)}
    os.system(f"cp {st[0]} {ROOT}/profiles/{RND}_{w}_kernel_stats.csv")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{RND}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
