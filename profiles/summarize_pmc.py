"""Average the per-dispatch PMC counters of rocprofv3 csv output per kernel.
usage: python profiles/summarize_pmc.py gpurun_out/pmc_<tag>_*"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if not k.startswith("mvf::"):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
