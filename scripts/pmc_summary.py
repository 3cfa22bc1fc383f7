"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            if "fk_pass" not in k:
                continue
            k = k.split("(")[0].replace("void d4w::", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-32s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
