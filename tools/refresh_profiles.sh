#!/bin/bash
# Regenerates profiles/<tag>_* on a GPU box (run through gpurun from the repo root):
#   <tag>_kernel_stats.csv            rocprofv3 --kernel-trace --stats of the default bench command
#   <tag>_bench_under_rocprof.json    the bench JSON line printed in that run (its roofline.avg_launch_ms must agree with the stats)
#   <tag>_stream_timeline.txt         tools/stream_timeline.py over the kernel trace of that run
#   <tag>_main_queue_gaps.txt         tools/gap_list.py over the same trace (idle gaps of the main queue and the kernels around them)
#   <tag>_pmc_hbm_traffic.csv         per-kernel FETCH_SIZE / WRITE_SIZE averages from two separate --pmc passes
#   pmc_traffic.json                  profiles/pmc_traffic.json with this configuration's entry replaced by the last row of that csv
# The files are written under gpurun_out/ (merged back by gpurun); copy them into profiles/ afterwards.
set -eu
TAG="${1:-r05_bf16x3_B158}"
EXTRA="${2:-}"          # extra bench flags of the configuration profiled, e.g. "--f16f8 1 --f16-backward"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_new; rm -rf "$O"; mkdir -p "$O"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-extra --no-power --no-isolated $EXTRA > "$O/stats.log" 2>&1
find "$O/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/${TAG}_kernel_stats.csv"
grep "^{" "$O/stats.log" > "$O/${TAG}_bench_under_rocprof.json"
find "$O/stats" -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/stream_timeline.py {} > "$O/${TAG}_stream_timeline.txt"
find "$O/stats" -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/gap_list.py {} 40 > "$O/${TAG}_main_queue_gaps.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/pmc_$c" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra --no-power --no-prof $EXTRA > "$O/pmc_$c.log" 2>&1
  find "$O/pmc_$c" -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$O/pmc_$c.csv"
done
python - "$O" "$TAG" <<'PY'
import csv, collections, sys
O, TAG = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{O}/pmc_{c}.csv")):
        agg[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
rows = []
for k, v in agg.items():
    if not v["FETCH_SIZE"] or not v["WRITE_SIZE"]:
        continue
    f, w = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
    rows.append((k, len(v["FETCH_SIZE"]), f, w, (2 * f + w) * 1024 / 1e6))
rows.sort(key=lambda t: -t[1] * t[4])
with open(f"{O}/{TAG}_pmc_hbm_traffic.csv", "w") as fh:
    fh.write("kernel,launches,avg_FETCH_SIZE_KB,avg_WRITE_SIZE_KB,hbm_MB_per_launch(2xFETCH+WRITE)\n")
    for k, n, f, w, mb in rows:
        fh.write(f"\"{k}\",{n},{f:.1f},{w:.1f},{mb:.1f}\n")
    tp = [(n, f, w) for k, n, f, w, mb in rows if "persist_kernel" in k]
    if tp:
        N = sum(n for n, _, _ in tp); F = sum(n * f for n, f, _ in tp) / N; W = sum(n * w for n, _, w in tp) / N
        fh.write(f"\"gemm_bf16_persist_kernel (all instantiations)\",{N},{F:.1f},{W:.1f},{(2 * F + W) * 1024 / 1e6:.1f}\n")
        print("persist kernel HBM MB/launch:", (2 * F + W) * 1024 / 1e6)
        # profiles/pmc_traffic.json is what bench.py reads for roofline.traffic: same number, same run (copy it into profiles/ with the csv)
        import json
        line = json.loads(open(f"{O}/{TAG}_bench_under_rocprof.json").read().strip().splitlines()[-1])
        tj = json.load(open("profiles/pmc_traffic.json"))
        forms = line.get("config", {}).get("traffic_key_suffix", "")
        tj[f"{line['dtype']}{forms}:{line['config']['windows_per_gpu']}"] = {"bytes_per_launch": round((2 * F + W) * 1024, 0), "source": f"profiles/{TAG}_pmc_hbm_traffic.csv"}
        json.dump(tj, open(f"{O}/pmc_traffic.json", "w"), indent=1)
st = list(csv.DictReader(open(f"{O}/{TAG}_kernel_stats.csv")))
tp = [(float(r["TotalDurationNs"]), int(r["Calls"])) for r in st if "persist_kernel" in r["Name"]]
if tp:
    print("rocprof gemm_bf16_persist_kernel: calls", sum(c for _, c in tp), "avg ms", sum(t for t, _ in tp) / sum(c for _, c in tp) / 1e6)
PY
rm -rf "$O/stats" "$O"/pmc_FETCH_SIZE "$O"/pmc_WRITE_SIZE "$O"/pmc_*.csv
ls -la "$O"
