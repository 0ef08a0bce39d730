run() { echo "== $*"; ENVPRE="$*" NOWARM=1 bash tools/gpu_call.sh matrix "tj_hard tj_medium pp_hard" -- "--gate-split 1 --steps 160" 2>&1 | tail -3; }
run IC3_NOP=1
run IC3_PS_ZS=2
run IC3_PS_ZS=3
run IC3_PS_ZS=4
run IC3_PS_ZS=2 IC3_PS_ZEPI=1
run IC3_PS_ZS=2 IC3_PS_ZEPI=2
run IC3_PS_ZS=0 IC3_PS_ZEPI=2
run IC3_PS_ZS=2 IC3_PS_ZF=24
run IC3_PS_ZS=2 IC3_PS_Z0=16
run IC3_PS_ZS=2 IC3_PS_ZH=24
run IC3_PS_ZS=3 IC3_PS_ZC=8
run IC3_PS_ZFRAC=50
run IC3_PS_ZFRAC=85
bash tools/gpu_call.sh tests tests/test_gate_split_gpu.py tests/test_policy_gpu.py tests/test_policy_step_gpu.py tests/test_policy_step_onehop_gpu.py tests/test_trainer_gpu.py tests/test_gates_backward_gpu.py tests/test_multirank_gpu.py
