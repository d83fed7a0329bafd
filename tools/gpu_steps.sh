#!/bin/bash
# Runs GPU steps one after another on a gpurun box; a step that was killed at its time limit (exit 124 / 137) ends the sequence (no further
# GPU step after a hang), an ordinary failure does not.  Usage: tools/gpu_steps.sh "<secs> <name> <command ...>" ...   (logs: gpurun_out/<name>.log)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
for spec in "$@"; do
  secs="${spec%% *}"; rest="${spec#* }"; name="${rest%% *}"; cmd="${rest#* }"
  echo "=== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name exit $rc"; tail -n 4 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name hit its limit: stopping"; exit 1; fi
done
exit 0
