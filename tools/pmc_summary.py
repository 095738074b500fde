import csv, collections, sys, glob
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(k.ljust(48), " ".join("%s=%.4g" % (c, sum(x) / len(x)) for c, x in sorted(v.items())), "n=%d" % len(next(iter(v.values()))))
