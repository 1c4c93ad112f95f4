mkdir -p gpurun_out
unset VKAS_LIB_PATH
(python profiles/repro/repro_step_determinism5.py 200 > gpurun_out/r5_fix_a.log 2>&1) &
python profiles/repro/repro_step_determinism5.py 200 > gpurun_out/r5_fix_b.log 2>&1
wait
tail -n 3 gpurun_out/r5_fix_a.log | cut -c1-300; tail -n 1 gpurun_out/r5_fix_b.log
