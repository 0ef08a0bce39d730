P="python tools/exp/prefill_probe.py"
for w in 400 800 1600; do for d in 0 1 2 4 8 16; do IC3_FILL_WAVES=$w IC3_FILL_DEPTH=$d IC3_FILL_NAP=0 $P patch_fill; done; done
IC3_FILL_WAVES=400 IC3_FILL_DEPTH=2 IC3_FILL_NAP=0 IC3_FILL_MODE=1 $P patch_fill
IC3_FILL_WAVES=800 IC3_FILL_DEPTH=1 IC3_FILL_NAP=0 IC3_FILL_MODE=1 $P patch_fill
