mkdir -p gpurun_out
echo "== alone"; python profiles/repro/repro_head_tail_bwd.py 3000 2 1 2>&1 | tail -3
echo "== two at once"
(python profiles/repro/repro_head_tail_bwd.py 6000 2 1 > gpurun_out/ht_a.log 2>&1) &
python profiles/repro/repro_head_tail_bwd.py 6000 2 1 > gpurun_out/ht_b.log 2>&1
wait
tail -n 40 gpurun_out/ht_a.log; echo ====; tail -n 8 gpurun_out/ht_b.log
