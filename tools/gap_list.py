"""Largest idle gaps of the main HIP queue inside one steady-state training step of a rocprofv3 --kernel-trace CSV, with the kernels on
either side: python tools/gap_list.py <kernel_trace.csv> [n]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[adam[-2]:adam[-1] + 1]
qs = collections.Counter(r["Queue_Id"] for r in step)
mainq = qs.most_common(1)[0][0]
main = [r for r in step if r["Queue_Id"] == mainq]
gaps = []
for a, b in zip(main, main[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps.append((g, a["Kernel_Name"][:60], b["Kernel_Name"][:60]))
gaps.sort(reverse=True)
print(f"main queue {mainq}: {len(main)} kernels, total gap {sum(g for g, _, _ in gaps) / 1e6:.2f} ms")
for g, a, b in gaps[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f"{g / 1e3:8.1f} us  after {a}  before {b}")
hist = collections.Counter(min(int(g / 5e3) * 5, 50) for g, _, _ in gaps)
print("gap histogram (us bucket: count):", sorted(hist.items()))
