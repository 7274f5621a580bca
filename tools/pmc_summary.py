"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch of each counter for a kernel."""
import csv, glob, sys, collections
root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_tb_fused"
acc = collections.defaultdict(list)
for f in sorted(glob.glob(root + "/*/*/*counter_collection.csv")):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kern in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
