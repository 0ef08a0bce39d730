for mode in 0 1; do for waves in 100 200 400; do for nap in 0 1 2 3 4 6 8 12; do
  echo "=== mode $mode waves $waves nap $nap"
  ENVPRE="IC3_FILL_MODE=$mode IC3_FILL_WAVES=$waves IC3_FILL_NAP=$nap" bash tools/gpu_call.sh matrix "pp_hard" -- "--prefill-obs 1 --gate-split 1 --steps 80" 2>&1 | tail -1
done; done; done
for nap in 0 2 4 8 12 16 24 32; do
  echo "=== tj nap $nap"
  ENVPRE="IC3_FILL_NAP=$nap" bash tools/gpu_call.sh matrix "tj_hard tj_medium" -- "--prefill-obs 1 --gate-split 1 --steps 80" 2>&1 | tail -2
done
