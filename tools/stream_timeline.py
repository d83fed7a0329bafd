"""Per-queue view of one steady-state training step from a rocprofv3 --kernel-trace CSV: when each HIP stream (queue) starts / ends inside
the step, how busy it is, where the forward ends, and how long the main stream is alone / both streams run.  The step is delimited by
the fused Adam kernel."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a0, a1 = adam[-2], adam[-1]
step = rows[a0 + 1:a1 + 1]
t0 = int(rows[a0]["End_Timestamp"])
qs = collections.defaultdict(list)
for r in step:
    qs[r["Queue_Id"]].append((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"]))
span = max(e for v in qs.values() for _, e, _ in v)
print(f"step span {span/1e6:.2f} ms, {len(step)} kernels, queues: {len(qs)}")
def union(iv):
    iv = sorted(iv); b = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: b += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return b + ce - cs
for q, v in sorted(qs.items(), key=lambda kv: -len(kv[1])):
    print(f"queue {q}: {len(v)} kernels, first start {v[0][0]/1e6:.2f} ms, last end {max(e for _, e, _ in v)/1e6:.2f} ms, busy {union([(s, e) for s, e, _ in v])/1e6:.2f} ms, "
          f"sum of durations {sum(e - s for s, e, _ in v)/1e6:.2f} ms")
main = max(qs.values(), key=len)
side = [v for v in qs.values() if v is not main]
side_iv = sorted((s, e) for v in side for s, e, _ in v)
loss = [s for s, e, n in main if "wta_loss_kernel" in n or "single_loss" in n]
if loss:
    print(f"forward ends (loss kernel) at {loss[0]/1e6:.2f} ms")
# overlap accounting on the main queue: time of each main kernel during which some side kernel runs
def overlap(s, e):
    t = 0
    for a, b in side_iv:
        if b <= s: continue
        if a >= e: break
        t += min(e, b) - max(s, a)
    return t
cls = collections.defaultdict(lambda: [0, 0, 0])
for s, e, n in main:
    k = "gemm_persist" if "persist_kernel" in n else "gemm_tiled" if "gemm_" in n else "ln" if "ln_" in n else "attn" if "attn_" in n else "reduce" if "reduce" in n else "other"
    if loss and s >= loss[0]: k = "bwd:" + k
    c = cls[k]; c[0] += e - s; c[1] += overlap(s, e); c[2] += 1
for k, (d, o, n) in sorted(cls.items()):
    print(f"  main {k:18s} {n:4d} kernels {d/1e6:7.2f} ms, of which {o/1e6:6.2f} ms with a side-stream kernel running")
gaps = []
m = sorted((s, e) for s, e, _ in main)
for (s0, e0), (s1, e1) in zip(m, m[1:]):
    if s1 > e0: gaps.append(s1 - e0)
print(f"main-queue gaps: {sum(gaps)/1e6:.2f} ms in {len(gaps)} gaps, largest (us) {[round(g/1e3,1) for g in sorted(gaps, reverse=True)[:6]]}")
