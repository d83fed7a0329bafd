"""GPU idle time inside the timed steps: union of kernel [start, end) intervals from a rocprofv3 --kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
# last 40 % of the trace = steady-state steps
t0 = iv[0][0] + int(0.6 * (iv[-1][1] - iv[0][0]))
iv = [(s, e) for s, e in iv if s >= t0]
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
gaps.sort(reverse=True)
print(f"span {span/1e6:.1f} ms, busy {busy/1e6:.1f} ms ({100*busy/span:.1f} %), idle {(span-busy)/1e6:.2f} ms in {len(gaps)} gaps; "
      f"largest gaps (us): {[round(g/1e3,1) for g in gaps[:8]]}; gaps > 5 us: {sum(1 for g in gaps if g > 5000)}")
