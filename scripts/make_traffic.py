"""profiles/r02_traffic.json + profiles/r02_<workload>_kernel_stats.csv from the rocprofv3 outputs of one profiling round.

    python scripts/make_traffic.py <gpurun_out/prof_dir>

expects, per workload W: <dir>/W_stats (--kernel-trace --stats), <dir>/W_fetch (--pmc FETCH_SIZE), <dir>/W_write
(--pmc WRITE_SIZE), <dir>/W_bench.json (the bench line of an unprofiled run of the same command).  FETCH_SIZE /
WRITE_SIZE are in KB (MI355X_MICROARCH.md, HBM section); FETCH_SIZE is doubled on gfx950 as that guide says (128-B requests
tallied at 64 B) -- calibrated there for wide streaming reads only, so for 8/16-byte scattered loads it is an upper bound.
"""
import sys, os, glob, csv, json
d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = "r02"
KERN = {"mpa512": "k_mpa_sweep", "maaco512": "k_maaco_walk8", "maaco1024": "k_maaco_walk8", "maaco128": "k_maaco_walk(",
        "ga512": "k_decode_batch", "pso512": "k_decode_batch", "astar1024": "k_astar_batch"}
try:
    out = json.load(open(os.path.join(ROOT, "profiles", f"{RND}_traffic.json")))
except Exception:
    out = {}


def match(name, kern):
    return kern in name


def pmc(dirname, counter, kern):
    tot, calls = 0.0, set()
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if match(r["Kernel_Name"], kern) and r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"]); calls.add(r["Dispatch_Id"])
    return tot, len(calls)


for w, k in KERN.items():
    st = glob.glob(os.path.join(d, w + "_stats", "**", "*kernel_stats.csv"), recursive=True)
    if not st:
        continue
    rows = [r for r in csv.DictReader(open(st[0])) if match(r["Name"], k)]
    if not rows:
        continue
    calls = sum(int(r["Calls"]) for r in rows); tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    fetch, nf = pmc(os.path.join(d, w + "_fetch"), "FETCH_SIZE", k)
    write, nw = pmc(os.path.join(d, w + "_write"), "WRITE_SIZE", k)
    bench = json.loads(open(os.path.join(d, w + "_bench.json")).read().strip().splitlines()[-1])
    fb, wb = fetch * 1024 / max(nf, 1), write * 1024 / max(nw, 1)
    roof = bench["roofline"]
    # the stats run covers warm-up and set-up launches too: its average is over all launches of the kernel
    out[w] = {"kernel": roof["kernel"], "calls": calls, "avg_ms_rocprof": tot_ns / max(calls, 1) / 1e6,
              "bench_avg_launch_ms": roof["avg_launch_ms"], "bench_value": bench["value"], "bench_ms_per_step": bench["ms_per_step"],
              "FETCH_SIZE_bytes_per_launch_raw": fb, "WRITE_SIZE_bytes_per_launch": wb,
              "traffic_bytes_per_launch": 2 * fb + wb,
              "note": "FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); uncalibrated for 8/16-B scattered loads; separate --pmc passes",
              "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"],
              "traffic_over_algorithmic": (2 * fb + wb) / max(roof["algorithmic_bytes_per_launch"], 1)}
    os.system(f"cp {st[0]} {ROOT}/profiles/{RND}_{w}_kernel_stats.csv")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{RND}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
