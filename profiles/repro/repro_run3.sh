(python profiles/repro/repro_step_determinism3.py 80 > gpurun_out/r3_a.log 2>&1) &
python profiles/repro/repro_step_determinism3.py 80 > gpurun_out/r3_b.log 2>&1
wait
tail -c 3000 gpurun_out/r3_a.log; echo ====; tail -c 3000 gpurun_out/r3_b.log
