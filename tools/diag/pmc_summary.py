"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per (kernel, counter)."""
import csv, sys, glob, collections
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if "fast" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("ctc_fast_")[1].split("(")[0]
            agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print("%-34s %-24s %14.0f  (n=%d)" % (k[0], k[1], sum(agg[k]) / len(agg[k]), len(agg[k])))
