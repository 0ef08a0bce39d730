#!/bin/bash
# A/B builds of policy_step.hip: tools/build_variant.sh NAME "-DFLAG=..."  ->  ic3net_amd/csrc/libic3rollout_NAME.so
# (run with IC3_ROLLOUT_LIB=<that path>)
set -e
cd "$(dirname "$0")/../ic3net_amd/csrc"
make -s -j8 libic3rollout.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result $2 -c policy_step.hip -o policy_step_$1.o
objs=$(ls *.o | grep -v '^policy_step' | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libic3rollout_$1.so $objs policy_step_$1.o -ldl
echo built libic3rollout_$1.so
