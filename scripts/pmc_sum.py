"""Sum rocprofv3 --pmc counter CSVs per kernel: python scripts/pmc_sum.py <dir> [kernel-substring]"""
import sys, glob, csv, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, v in acc.items():
    print(k[:60], {c: int(x) for c, x in sorted(v.items())})
