import csv, sys, glob, collections
for g in sys.argv[1:]:
    for f in glob.glob(g + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
        for (k, c), (n, v) in sorted(acc.items()):
            if "fk" in k: print(f"{k:62s} {c:28s} n={n:3d} avg={v/n:.4g}")
