// TEST INFRASTRUCTURE — a stand-in for <hip/hip_runtime.h> that lets the PRODUCT's wave-level device functions
// (ic3net_amd/csrc/env_device.hpp: pp/tj_step_lanes, window tables, obs patches) compile for the host, so that they can be
// driven over the reference's golden trajectories on a CPU and under AddressSanitizer / UndefinedBehaviorSanitizer
// (tests/host/ic3_host_build.cpp, tests/test_host_build_cpu.py, tools/host_asan.sh).  Not a CPU fallback of the product:
// nothing under ic3net_amd/ loads it.
//
// Execution model: one 64-lane wavefront = 64 host threads running the same device function in lockstep where the code
// asks for it — __ballot / __shfl exchange through a shared array between two barriers (the device code calls them from
// all lanes of a wave, uniformly: that is what makes it valid on the GPU too).
#pragma once

#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
inline const char* hipGetErrorString(hipError_t) { return "host build"; }

struct int2 {
    int x, y;
};
inline int2 make_int2(int x, int y) { return int2{ x, y }; }
struct ic3_host_dim3 {
    unsigned x, y, z;
};

namespace ic3_host {
struct Wave {
    std::barrier<> bar{ 64 };
    long long vals[64];
    int active[64];   // a lane that has returned contributes 0 to later ballots (like an exited hardware lane)
};
inline thread_local int tl_lane = 0;
inline thread_local Wave* tl_wave = nullptr;
inline thread_local ic3_host_dim3 tl_tid{ 0, 0, 0 };

// runs fn(lane) on 64 lockstep lanes
inline void run_wave(const std::function<void(int)>& fn)
{
    Wave w;
    for (int l = 0; l < 64; ++l) w.active[l] = 1;
    std::vector<std::thread> th;
    th.reserve(64);
    for (int l = 0; l < 64; ++l)
        th.emplace_back([&, l]() {
            tl_lane = l;
            tl_wave = &w;
            tl_tid = ic3_host_dim3{ (unsigned)l, 0, 0 };
            fn(l);
            w.active[l] = 0;
            w.bar.arrive_and_drop();   // a lane that returns leaves the barrier (all cross-lane ops are behind it)
        });
    for (auto& t : th) t.join();
}
inline long long exchange(long long v, int src)
{
    Wave* w = tl_wave;
    w->vals[tl_lane] = v;
    w->bar.arrive_and_wait();
    const long long r = w->vals[src & 63];
    w->bar.arrive_and_wait();
    return r;
}
}  // namespace ic3_host

#define threadIdx (ic3_host::tl_tid)

inline unsigned long long __ballot(int pred)
{
    ic3_host::Wave* w = ic3_host::tl_wave;
    w->vals[ic3_host::tl_lane] = pred ? 1 : 0;
    w->bar.arrive_and_wait();
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (w->active[l]) m |= (unsigned long long)(w->vals[l] & 1) << l;
    w->bar.arrive_and_wait();
    return m;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __shfl(int v, int src) { return (int)ic3_host::exchange(v, src); }
inline float __shfl(float v, int src)
{
    int i;
    std::memcpy(&i, &v, 4);
    i = (int)ic3_host::exchange(i, src);
    std::memcpy(&v, &i, 4);
    return v;
}
template <class T>
inline T __shfl_xor(T v, int m) { return __shfl(v, ic3_host::tl_lane ^ m); }
// DPP row operations of group_sum<G> (quad_perm / row_half_mirror / row_mirror): the lane each control reads from
inline int __builtin_amdgcn_update_dpp(int, int src, int ctrl, int, int, bool)
{
    const int l = ic3_host::tl_lane;
    int from = l;
    if (ctrl == 0xB1) from = l ^ 1;
    else if (ctrl == 0x4E) from = l ^ 2;
    else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));
    else if (ctrl == 0x140) from = (l & ~15) | (15 - (l & 15));
    return __shfl(src, from);
}

inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T>
inline T min(T a, T b) { return a < b ? a : b; }
template <class T>
inline T max(T a, T b) { return a > b ? a : b; }

// raw buffer loads behind a descriptor (BufRows of the sparse encoder): base pointer + byte range check
struct __amdgpu_buffer_rsrc_t {
    const char* base;
    uint32_t bytes;
};
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int bytes, int)
{
    return __amdgpu_buffer_rsrc_t{ (const char*)p, (uint32_t)bytes };
}
typedef unsigned int ic3_host_u32x4 __attribute__((ext_vector_type(4)));
inline ic3_host_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int)
{
    ic3_host_u32x4 v = { 0, 0, 0, 0 };
    const uint32_t off = (uint32_t)voff + (uint32_t)soff;
    if ((uint64_t)off + 16 <= r.bytes) std::memcpy(&v, r.base + off, 16);
    return v;
}
