import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fused_rqs" not in k: continue
        agg["inv" if "Lb1E" in k else "fwd"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for d, cs in agg.items():
    for c, v in sorted(cs.items()):
        print("%s %-28s mean %.4g  (n=%d)" % (d, c, sum(v) / len(v), len(v)))
