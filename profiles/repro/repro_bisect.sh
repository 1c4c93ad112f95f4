run2() { echo "--- $1"; (env $1 python profiles/repro/repro_step_determinism.py 80 2>&1 | tail -1) & (env $1 python profiles/repro/repro_step_determinism.py 80 2>&1 | tail -1); wait; }
run2 "VKAS_NT_RING=0 VKAS_NT_NOSLAB=1 VKAS_TN_NOSLAB=1 VKAS_NO_MLP_CHAIN=1"
run2 "VKAS_NT_RING=0 VKAS_NO_MLP_CHAIN=1"
run2 "VKAS_NT_RING=0 VKAS_NT_NOSLAB=1 VKAS_TN_NOSLAB=1"
run2 "VKAS_GEMM=simple VKAS_NO_MLP_CHAIN=1"
