#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_profile.sh <tag>
# rocprofv3 --kernel-trace --stats of `python3 bench.py`, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (never combined with tracing); everything summarised into gpurun_out/<tag>.txt for profiles/.
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag
rm -rf $out $out.* ; mkdir -p $out
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $out/stats -o s -- python3 $R/bench.py --steps 20 --warmup 3 > $out/bench_stats.log 2>&1 || { echo stats failed; tail -5 $out/bench_stats.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o f -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_fetch.log 2>&1 || { echo fetch failed; tail -5 $out/bench_fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/write -o w -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_write.log 2>&1 || { echo write failed; tail -5 $out/bench_write.log; exit 1; }
{
  python3 $R/tools/kernel_resources.py
  echo
  for d in stats fetch write; do python3 $R/tools/rocpd_summary.py $(find $out/$d -name '*.db' | head -1); done
  grep -h '^{"metric"' $out/bench_stats.log
} > $R/gpurun_out/$tag.txt 2>&1
grep -c . $R/gpurun_out/$tag.txt
