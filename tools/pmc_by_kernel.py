"""Mean of every counter per kernel name from rocprofv3 --pmc output directories.  usage: pmc_by_kernel.py <dir> [name substring]"""
import csv, glob, os, sys
from collections import defaultdict
src, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, " ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())), "n=%d" % max(len(v) for v in acc[k].values()))
