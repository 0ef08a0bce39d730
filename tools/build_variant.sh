#!/bin/bash
# A/B builds of one source of libic3rollout: tools/build_variant.sh NAME "-DFLAG=..." [SOURCE (default policy_step)]
#   ->  ic3net_amd/csrc/libic3rollout_NAME.so   (run with IC3_ROLLOUT_LIB=<that path>)
set -e
cd "$(dirname "$0")/../ic3net_amd/csrc"
SRC=${3:-policy_step}
make -s -j8 libic3rollout.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-result $2 -c $SRC.hip -o variant_${SRC}_$1.o
objs=$(ls *.o | grep -v "^variant_" | grep -v "^policy_step_" | grep -v "^$SRC.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libic3rollout_$1.so $objs variant_${SRC}_$1.o -ldl
echo built libic3rollout_$1.so
