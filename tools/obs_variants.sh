#!/bin/bash
# A/B of the obs-assembly kernel variants (experiments; IC3_OBS_VARIANT is not a product knob)
for v in 0 1 2 3; do echo "variant $v"; IC3_OBS_VARIANT=$v python tools/microbench_env.py 2>&1 | grep -E "pp_hard|pp_scaled"; done
