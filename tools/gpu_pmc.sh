#!/bin/bash
# usage (on the GPU box): tools/gpu_pmc.sh <tag> <spec> "<counters pass 1>" ["<counters pass 2>" ...]
# one rocprofv3 --pmc run per pass (never mixed with tracing), summarised by tools/rocpd_summary.py
tag=$1; spec=$2; shift 2
cd /tmp && export TMPDIR=/tmp
n=0
for pass in "$@"; do
  n=$((n+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$n
  rm -rf $out
  timeout -s KILL 90 rocprofv3 --pmc $pass -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/gpu_time.py pmc $spec > $out.log 2>&1 || { echo "pass $n failed"; tail -5 $out.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $out -name '*.db' | head -1) > $out.txt 2>&1
  cat $out.txt
done
