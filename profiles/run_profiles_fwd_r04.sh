# rocprofv3 kernel stats of the two forward-only configurations (eager launches, so every kernel is traced by name) + their
# bench lines with graph replay.  usage (GPU box, repo root): bash profiles/run_profiles_fwd_r04.sh
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04fwd; mkdir -p $O
for C in 2 5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c$C -o p -- python3 $R/bench.py --config $C --no-graphs --steps 10 --warmup 2 > $O/c$C.log 2>&1
  cp $(find $O/c$C -name "*kernel_stats.csv" | head -1) $O/r04_config${C}_kernel_stats.csv
  rm -rf $O/c$C
  python3 $R/bench.py --config $C > $O/r04_bench_config$C.json 2> $O/r04_bench_config$C.log
done
