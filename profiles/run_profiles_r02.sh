# rocprofv3 runs of round 2 (program directly after "--"): kernel stats of the default bench, then the two PMC passes.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02prof}; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-per-pass > $O/stats.log 2>&1
if [ "$2" = pmc ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-per-pass > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-per-pass > $O/write.log 2>&1
fi
tail -1 $O/stats.log | cut -c1-300
find $O -name "*kernel_stats.csv" | head
