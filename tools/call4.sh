P="python tools/exp/prefill_probe.py"
$P noobs; $P fused; $P patch_nofill
for w in 25 50 100 400; do for nap in 0 4 16 64; do IC3_FILL_WAVES=$w IC3_FILL_NAP=$nap $P patch_fill; done; done
IC3_FILL_WAVES=100 IC3_FILL_NAP=2 IC3_FILL_MODE=1 $P patch_fill
$P noobs tj_hard; $P fused tj_hard; $P patch_nofill tj_hard; IC3_FILL_WAVES=100 IC3_FILL_NAP=8 $P patch_fill tj_hard
