"""Average of one PMC counter per kernel from a rocprofv3 counter_collection CSV (+ average duration when the kernel trace is there)."""
import collections, csv, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{c:12s} n={len(v):4d} avg={sum(v)/len(v):12.1f} KB -> x2 = {2*sum(v)/len(v)*1024/1e6:9.1f} MB, x1 = {sum(v)/len(v)*1024/1e6:9.1f} MB  {k[:100]}")
