#!/bin/bash
# Matrix-core utilisation per kernel from PMC counters (own pass, kernel trace only): SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix
# pipe is busy, summed over the chip's 1024 SIMDs) against GRBM_GUI_ACTIVE (busy clock cycles, summed over the 8 XCDs) of the same
# dispatches -> utilisation = MFMA_BUSY / (1024 * GUI_ACTIVE / 8).  Writes gpurun_out/profiles_new/<tag>_pmc_mfma_util.csv
set -eu
TAG="${1:-r05_bf16x3_B158}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_new; mkdir -p "$O"; rm -rf "$O/mf"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/mf" -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra --no-power --no-prof > "$O/mfma.log" 2>&1
python - "$(find "$O/mf" -name '*counter_collection.csv' | head -1)" "$O/${TAG}_pmc_mfma_util.csv" <<'PY'
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, v in agg.items():
    m, g = v.get("SQ_VALU_MFMA_BUSY_CYCLES"), v.get("GRBM_GUI_ACTIVE")
    if not m or not g or sum(g) == 0:
        continue
    rows.append((k, len(m), sum(m) / len(m), sum(g) / len(g), sum(m) / (1024.0 * sum(g) / 8.0)))
rows.sort(key=lambda t: -t[1] * t[2])
with open(sys.argv[2], "w") as f:
    f.write("kernel,launches,avg_SQ_VALU_MFMA_BUSY_CYCLES,avg_GRBM_GUI_ACTIVE,mfma_utilisation(busy/(1024*gui/8))\n")
    for k, n, m, g, u in rows:
        f.write(f"\"{k}\",{n},{m:.0f},{g:.0f},{u:.4f}\n")
    tp = [(n, m, g) for k, n, m, g, u in rows if "persist_kernel" in k]
    if tp:
        M = sum(n * m for n, m, g in tp); G = sum(n * g for n, m, g in tp)
        f.write(f"\"gemm_bf16_persist_kernel (all instantiations)\",{sum(n for n, _, _ in tp)},,,{M / (1024.0 * G / 8.0):.4f}\n")
        print("persist kernel MFMA utilisation", M / (1024.0 * G / 8.0))
for k, n, m, g, u in rows[:12]:
    print(f"{u:6.3f}  n={n:4d}  {k[:110]}")
PY
rm -rf "$O/mf"
