// TEST INFRASTRUCTURE — stand-in for <hip/hip_ext.h> (see hip_runtime.h): the launch that stamps two events itself.
#pragma once
#include <hip/hip_runtime.h>

#define hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ev0, ev1, flags, ...) \
    do {                                                                               \
        if (ev0) (void)hipEventRecord(ev0, stream);                                     \
        ic3_host::launch(kernel, grid, block, lds, stream, ##__VA_ARGS__);              \
        if (ev1) (void)hipEventRecord(ev1, stream);                                     \
    } while (0)
