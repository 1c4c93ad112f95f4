# two copies of the repeated-step check at once: flat gradient of N identical steps against the first (bound 1e-5)
mkdir -p gpurun_out
(python profiles/repro/repro_step_determinism.py ${1:-300} > gpurun_out/soak_a.log 2>&1) &
python profiles/repro/repro_step_determinism.py ${1:-300} > gpurun_out/soak_b.log 2>&1
wait
grep -c differs gpurun_out/soak_a.log gpurun_out/soak_b.log; tail -n 2 gpurun_out/soak_a.log | cut -c1-300; tail -n 2 gpurun_out/soak_b.log | cut -c1-300
