"""Summarise rocprofv3 counter_collection CSVs: per kernel (short name), mean counter value per dispatch."""
import csv, sys, collections, re
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"]
        k = re.sub(r"\(.*", "", k).replace("void bazmusic::", "")[:48]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", path)
    for k, cs in agg.items():
        if "bazmusic" not in k and "kernel" not in k: continue
        print(" ", k)
        for cn, vals in cs.items():
            print("     %-32s mean %.4g  (n=%d)" % (cn, sum(vals) / len(vals), len(vals)))
