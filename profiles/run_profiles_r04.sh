# rocprofv3 runs of round 4 (program directly after "--"): kernel stats of the default bench, then the two PMC passes, parsed
# on the box into the small files that get committed:
#   profiles/r04_kernel_stats.csv, r04_stats.log (bench line of that run), r04_pmc_traffic.txt, pmc_traffic.json
# usage (on the GPU box, from the repo root): bash profiles/run_profiles_r04.sh <label> [commit]
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; L=${1:-r04}; O=$R/gpurun_out/${L}prof; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-per-pass > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-per-pass > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-per-pass > $O/write.log 2>&1
mkdir -p $R/gpurun_out/${L}
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${L}/${L}_kernel_stats.csv
grep -e '"metric"' -e '^\[bench\]' $O/stats.log > $R/gpurun_out/${L}/${L}_stats.log
python3 $R/profiles/parse_pmc.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) \
  $R/gpurun_out/${L}/pmc_traffic.json "profiles/${L} rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 1 --warmup 0 --no-per-pass (commit ${2:-HEAD})" \
  > $R/gpurun_out/${L}/${L}_pmc_traffic.txt
rm -rf $O/fetch $O/write $O/stats   # the raw traces are tens of MB
head -14 $R/gpurun_out/${L}/${L}_pmc_traffic.txt
