#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_profile.sh <tag> [extra bench.py arguments, e.g. --workload cover]
# rocprofv3 --kernel-trace --stats of `python3 bench.py`, then FETCH_SIZE, WRITE_SIZE and two SQ passes in separate
# --pmc runs (never combined with tracing, at most 8 SQ counters per run); everything summarised into
# gpurun_out/<tag>.txt for profiles/ (--no-extras: every dispatch of the render kernel is then a full frame of the
# workload, so per-dispatch averages are per frame), and the HBM bytes per dispatch into gpurun_out/<tag>_traffic.json.
tag=$1; shift; extra="$@"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag
rm -rf $out $out.* ; mkdir -p $out
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $out/stats -o s -- python3 $R/bench.py --steps 20 --warmup 3 --no-extras $extra > $out/bench_stats.log 2>&1 || { echo stats failed; tail -5 $out/bench_stats.log; exit 1; }
pass() {  # name, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 -d $out/$1 -o p -- python3 $R/bench.py --steps 5 --warmup 1 --no-extras --no-cpu-baseline $extra > $out/bench_$1.log 2>&1 || { echo "$1 failed"; tail -5 $out/bench_$1.log; exit 1; }
}
pass fetch "FETCH_SIZE" && pass write "WRITE_SIZE" &&
pass sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" &&
pass sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY" || exit 1
{
  python3 $R/tools/kernel_resources.py
  echo
  for d in stats fetch write sq1 sq2; do python3 $R/tools/rocpd_summary.py $(find $out/$d -name '*.db' | head -1); done
  grep -h '^{"metric"' $out/bench_stats.log
} > $R/gpurun_out/$tag.txt 2>&1
python3 $R/tools/traffic_json.py $R/gpurun_out/$tag.txt profiles/$tag.txt > $R/gpurun_out/${tag}_traffic.json
grep -c . $R/gpurun_out/$tag.txt
cat $R/gpurun_out/${tag}_traffic.json
