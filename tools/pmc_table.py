"""Table of averaged PMC counters per kernel over several rocprofv3 --pmc passes (development aid).  argv: directory"""
import collections, csv, glob, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", row["Kernel_Name"])).replace("gprx::", "")
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, counters in sorted(acc.items()):
    if name.startswith("__amd"):
        continue
    print(name[:60], "dispatches", max(len(v) for v in counters.values()))
    for c, v in sorted(counters.items()):
        print(f"    {c:32s} {sum(v)/len(v):.4e}")
