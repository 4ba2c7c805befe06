#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel launch."""
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", path)
    for k, d in acc.items():
        if not any(s in k for s in ("render", "preprocess", "radix", "duplicate", "scan", "identify", "adam", "ply", "l1_ssim", "knn")):
            continue
        print(k.ljust(42), " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
