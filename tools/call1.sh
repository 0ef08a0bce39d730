bash tools/gpu_call.sh probe VALU-free
bash tools/gpu_call.sh matrix "pp_hard tj_hard tj_medium" -- "--prefill-obs 0 --gate-split 0" "--prefill-obs 0 --gate-split 1" "--prefill-obs 1 --gate-split 0" "--prefill-obs 1 --gate-split 1" "--prefill-obs 1 --gate-split 1 --time-kernels 0"
bash tools/gpu_call.sh tests tests/test_policy_step_onehop_gpu.py tests/test_policy_step_gpu.py tests/test_trainer_gpu.py tests/test_auto_reset_gpu.py
