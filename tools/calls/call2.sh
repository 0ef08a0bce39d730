for cfg in "IC3_FILL_WGS=100 IC3_FILL_MODE=1" "IC3_FILL_WGS=100 IC3_FILL_MODE=0" "IC3_FILL_WGS=200 IC3_FILL_MODE=1" "IC3_FILL_WGS=50 IC3_FILL_MODE=1" "IC3_FILL_WGS=400 IC3_FILL_MODE=1"; do
  echo "=== $cfg"
  ENVPRE="$cfg" bash tools/gpu_call.sh matrix "pp_hard" -- "--prefill-obs 1 --gate-split 1" "--prefill-obs 1 --gate-split 0"
done
echo "=== defaults, all workloads"
bash tools/gpu_call.sh matrix "pp_hard tj_hard tj_medium pp_easy" -- "--prefill-obs 0 --gate-split 1" "--prefill-obs 1 --gate-split 1" "--prefill-obs 1 --gate-split 1 --time-kernels 0" "--no-dense-obs --gate-split 1"
bash tools/gpu_call.sh matrix "pp_scaled" -- "--prefill-obs 0 --gate-split 1 --steps 20" "--prefill-obs 1 --gate-split 1 --steps 20"
bash tools/gpu_call.sh tests tests/test_policy_step_onehop_gpu.py tests/test_policy_step_gpu.py tests/test_trainer_gpu.py tests/test_multirank_gpu.py tests/test_abi_cpu.py -m ""
