"""profiles/r01_traffic.json from the rocprofv3 outputs of one round of profiling runs.

    python scripts/make_traffic.py <gpurun_out/prof_dir>

expects, per workload W in (mpa512, maaco512): <dir>/W_stats (--kernel-trace --stats), <dir>/W_fetch (--pmc FETCH_SIZE),
<dir>/W_write (--pmc WRITE_SIZE), <dir>/W_bench.json (the bench line of an unprofiled run of the same command).
FETCH_SIZE / WRITE_SIZE are in KB (MI355X_MICROARCH.md, HBM section); FETCH_SIZE is doubled on gfx950 as that guide says.
"""
import sys, os, glob, csv, json, collections
d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERN = {"mpa512": "k_mpa_sweep", "maaco512": "k_maaco_walk8"}
out = {}
def pmc(dirname, counter, kern):
    tot, calls = 0.0, set()
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(kern) or (" " + kern) in r["Kernel_Name"] or kern + "(" in r["Kernel_Name"]:
                if r["Counter_Name"] == counter:
                    tot += float(r["Counter_Value"]); calls.add(r["Dispatch_Id"])
    return tot, len(calls)
for w, k in KERN.items():
    st = glob.glob(os.path.join(d, w + "_stats", "**", "*kernel_stats.csv"), recursive=True)
    if not st:
        continue
    row = [r for r in csv.DictReader(open(st[0])) if k + "(" in r["Name"]][0]
    fetch, nf = pmc(os.path.join(d, w + "_fetch"), "FETCH_SIZE", k)
    write, nw = pmc(os.path.join(d, w + "_write"), "WRITE_SIZE", k)
    bench = json.loads(open(os.path.join(d, w + "_bench.json")).read().strip().splitlines()[-1])
    fb, wb = fetch * 1024 / max(nf, 1), write * 1024 / max(nw, 1)
    out[w] = {"kernel": k, "calls": int(row["Calls"]), "avg_ms_rocprof": float(row["AverageNs"]) / 1e6,
              "bench_avg_launch_ms": bench["roofline"]["avg_launch_ms"], "bench_value": bench["value"],
              "FETCH_SIZE_bytes_per_launch_raw": fb, "WRITE_SIZE_bytes_per_launch": wb,
              "traffic_bytes_per_launch": 2 * fb + wb,
              "note": "FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); uncalibrated for 16-B scattered loads; separate --pmc passes",
              "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"]}
    os.system(f"cp {st[0]} {ROOT}/profiles/r01_{w}_kernel_stats.csv")
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
