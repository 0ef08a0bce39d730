// TEST INFRASTRUCTURE — a stand-in for <hip/hip_runtime.h> that lets the PRODUCT's device code compile for the host: the
// wave-level functions of ic3net_amd/csrc/env_device.hpp behind tests/host/ic3_host_build.cpp, and every .hip source of the
// library — whole kernels, launches, the matrix-core builtins — behind the library's own C ABI (tests/host/Makefile ->
// libic3rollout_host.so), so that the reference's golden trajectories and ic3_policy_step free runs can be driven through
// the real kernel sources on a CPU, deterministically and under AddressSanitizer / UndefinedBehaviorSanitizer
// (tests/test_host_build_cpu.py, test_host_abi_cpu.py, test_host_policy_step_cpu.py, tools/host_asan.sh).  Not a CPU
// fallback of the product: nothing under ic3net_amd/ loads it, and it says nothing about the hardware itself (wait counts,
// hazards, occupancy, timing).
//
// Execution model: a lane is a FIBER (ucontext), all lanes of a wavefront / workgroup are scheduled cooperatively on the
// calling thread, round-robin, each running until it returns or waits in a barrier.  __ballot / __shfl exchange through a
// per-wave array between two barriers over the wave's lanes (the device code calls them from all lanes of a wave,
// uniformly: that is what makes it valid on the GPU too); a lane that returns leaves the barriers like an exited hardware
// lane.  No OS threads, no real atomics: runs are deterministic, a barrier nobody can complete aborts with a message
// instead of hanging, and a launch costs context switches instead of futex traffic.
//
// Whole kernels (tests/host/ic3_host_abi.cpp: the product's .hip sources behind the same C ABI, `device = -1`):
// hipLaunchKernelGGL runs the grid one workgroup at a time, a workgroup = blockDim lanes (wavefronts of 64 as above,
// __syncthreads = a barrier over the workgroup's lanes).  `__shared__` arrays become function-local statics (one workgroup
// runs at a time), the dynamic LDS of a launch is one 160 KB buffer refilled with a poison pattern before every workgroup
// (LDS is not zeroed on the GPU either).  hipMalloc / hipMemcpy / hipMemset work on host memory (allocations poisoned the
// same way), streams are ignored (every call is synchronous).
#pragma once

#include <sys/mman.h>
#include <ucontext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <tuple>
#include <type_traits>
#include <vector>

#ifndef __HIPCC__
#define __HIPCC__ 1   // (ic3_common.hpp guards its device-only helpers with it)
#endif
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101 };
inline const char* hipGetErrorString(hipError_t) { return "host build"; }

struct int2 {
    int x, y;
};
inline int2 make_int2(int x, int y) { return int2{ x, y }; }
struct ic3_host_dim3 {
    unsigned x, y, z;
};
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define IC3_HOST_ASAN_FIBERS 1
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#endif
#endif

namespace ic3_host {
void yield();
// a barrier over `expected` lanes; a lane that drops out lowers the count for this and every later phase
struct Barrier {
    int expected, arrived = 0;
    unsigned phase = 0;
    explicit Barrier(int n) : expected(n) {}
    void arrive_and_wait();
    void arrive_and_drop();
};
struct Wave {
    Barrier bar;
    explicit Wave(int lanes = 64) : bar(lanes) {}
    long long vals[64];
    int active[64];   // a lane that has returned contributes 0 to later ballots (like an exited hardware lane)
    alignas(16) unsigned char wide[64][32];   // operands of an emulated MFMA (up to 2 x 16 bytes per lane)
};
struct Block {
    Barrier bar;
    explicit Block(int n) : bar(n) {}
};
inline int tl_lane = 0;
inline Wave* tl_wave = nullptr;
inline Block* tl_block = nullptr;
inline ic3_host_dim3 tl_tid{ 0, 0, 0 }, tl_bid{ 0, 0, 0 }, tl_bdim{ 64, 1, 1 }, tl_gdim{ 1, 1, 1 };
inline const void* tl_kernarg = nullptr;   // the first kernel parameter of the running launch (__builtin_amdgcn_kernarg_segment_ptr)
constexpr size_t LDS_BYTES = 160 * 1024;
inline void* dynamic_lds()
{
    alignas(64) static unsigned char buf[LDS_BYTES];
    return buf;
}

// The cooperative scheduler.  run(n, fn) plays fn(0) .. fn(n - 1) as n fibers until all have returned.
class Fibers {
public:
#ifdef IC3_HOST_ASAN_FIBERS
    static constexpr size_t STACK = 2048 * 1024;   // (unoptimised, instrumented frames of the big kernels)
#else
    static constexpr size_t STACK = 256 * 1024;
#endif
    struct Fiber {
        ucontext_t ctx;
        char* stack = nullptr;
        void* fake = nullptr;   // (ASan) fake-stack handle while switched out
        bool done = false;
        int lane = 0;           // the per-lane registers of the shim, saved / restored around a switch
        Wave* wave = nullptr;
        ic3_host_dim3 tid{ 0, 0, 0 };
    };
    static Fibers& get()
    {
        static Fibers* f = new Fibers();
        return *f;
    }
    void run(int n, const std::function<void(int)>& fn)
    {
        if (cur_ >= 0) die("nested launch from device code");
        while ((int)fibers_.size() < n) {
            fibers_.emplace_back(new Fiber());
            void* p = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_STACK, -1, 0);
            if (p == MAP_FAILED) die("mmap of a fiber stack failed");
            fibers_.back()->stack = (char*)p;
        }
        job_ = &fn;
        for (int i = 0; i < n; ++i) {
            Fiber& f = *fibers_[i];
            f.done = false;
            f.fake = nullptr;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = STACK;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, (void (*)())trampoline, 0);
        }
        // IC3_HOST_SCHED=reverse walks the lanes from the highest to the lowest: code whose result depends on which lane
        // runs first between two cross-lane operations (a wave-lockstep assumption) gives different results under the two
        // orders (tests/test_host_policy_step_cpu.py::test_results_do_not_depend_on_the_lane_schedule)
        // ("shuffle": a different pseudo-random order in every round)
        static const char* sched = std::getenv("IC3_HOST_SCHED");
        static const bool reverse = sched && !std::strcmp(sched, "reverse"), shuffle = sched && !std::strcmp(sched, "shuffle");
        std::vector<int> order(n);
        for (int k = 0; k < n; ++k) order[k] = reverse ? n - 1 - k : k;
        int left = n;
        while (left > 0) {
            const unsigned long long before = progress_;
            if (shuffle)
                for (int k = n - 1; k > 0; --k) {
                    rng_ = rng_ * 6364136223846793005ull + 1442695040888963407ull;
                    std::swap(order[k], order[(int)((rng_ >> 33) % (unsigned long long)(k + 1))]);
                }
            for (int k = 0; k < n; ++k) {
                const int i = order[k];
                Fiber& f = *fibers_[i];
                if (f.done) continue;
                cur_ = i;
                tl_lane = f.lane;
                tl_wave = f.wave;
                tl_tid = f.tid;
                switch_to(f);
                f.lane = tl_lane;
                f.wave = tl_wave;
                f.tid = tl_tid;
                cur_ = -1;
                if (f.done) { --left; ++progress_; }
            }
            if (left > 0 && progress_ == before)
                die("no lane can make progress: a barrier (__syncthreads / __shfl / __ballot) that not all live lanes reach");
        }
        job_ = nullptr;
    }
    void yield_current()
    {
        if (cur_ < 0) die("barrier outside a launch");
        Fiber& f = *fibers_[cur_];
#ifdef IC3_HOST_ASAN_FIBERS
        __sanitizer_start_switch_fiber(&f.fake, main_bottom_, main_size_);
#endif
        swapcontext(&f.ctx, &main_);
#ifdef IC3_HOST_ASAN_FIBERS
        __sanitizer_finish_switch_fiber(f.fake, nullptr, nullptr);
#endif
    }
    void progressed() { ++progress_; }
    [[noreturn]] static void die(const char* what)
    {
        std::fprintf(stderr, "ic3 host shim: %s\n", what);
        std::abort();
    }

private:
    void switch_to(Fiber& f)
    {
#ifdef IC3_HOST_ASAN_FIBERS
        void* fake = nullptr;
        __sanitizer_start_switch_fiber(&fake, f.stack, STACK);
#endif
        swapcontext(&main_, &f.ctx);
#ifdef IC3_HOST_ASAN_FIBERS
        __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
    }
    static void trampoline()
    {
        Fibers& s = get();
#ifdef IC3_HOST_ASAN_FIBERS
        __sanitizer_finish_switch_fiber(nullptr, &s.main_bottom_, &s.main_size_);
#endif
        const int id = s.cur_;
        (*s.job_)(id);
        Fiber& f = *s.fibers_[id];
        f.done = true;
#ifdef IC3_HOST_ASAN_FIBERS
        __sanitizer_start_switch_fiber(nullptr, s.main_bottom_, s.main_size_);   // (this fiber's fake stack is released)
#endif
        swapcontext(&f.ctx, &s.main_);
        die("a finished fiber was resumed");
    }
    std::vector<std::unique_ptr<Fiber>> fibers_;
    const std::function<void(int)>* job_ = nullptr;
    ucontext_t main_;
    [[maybe_unused]] const void* main_bottom_ = nullptr;   // (ASan) bounds of the scheduler's own stack
    [[maybe_unused]] size_t main_size_ = 0;
    int cur_ = -1;
    unsigned long long progress_ = 0, rng_ = 0x9E3779B97F4A7C15ull;
};
inline void yield() { Fibers::get().yield_current(); }
inline void Barrier::arrive_and_wait()
{
    const unsigned mine = phase;
    if (++arrived >= expected) {
        arrived = 0;
        ++phase;
        Fibers::get().progressed();
        return;
    }
    while (phase == mine) yield();
}
inline void Barrier::arrive_and_drop()
{
    --expected;
    if (expected > 0 && arrived >= expected) {   // the lanes already waiting were only waiting for this one
        arrived = 0;
        ++phase;
        Fibers::get().progressed();
    }
}

// runs fn(lane) on 64 lockstep lanes
inline void run_wave(const std::function<void(int)>& fn)
{
    Wave w;
    for (int l = 0; l < 64; ++l) w.active[l] = 1;
    Fibers::get().run(64, [&](int l) {
        tl_lane = l;
        tl_wave = &w;
        tl_tid = ic3_host_dim3{ (unsigned)l, 0, 0 };
        fn(l);
        w.active[l] = 0;
        w.bar.arrive_and_drop();   // a lane that returns leaves the barrier (all cross-lane ops are behind it)
    });
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
#define blockIdx (ic3_host::tl_bid)
#define blockDim (ic3_host::tl_bdim)
#define gridDim (ic3_host::tl_gdim)
#define __shared__ static
// (ic3_common.hpp's spelling of `extern __shared__ T name[]`)
#define IC3_DYNAMIC_LDS(T, name) T* const name = reinterpret_cast<T*>(ic3_host::dynamic_lds())
// (ic3_common.hpp's compiler fences of the hand-scheduled kernels)
#define IC3_OPAQUE_SGPR(x) asm volatile("" : "+r"(x))
#define IC3_OPAQUE_VGPR(x) asm volatile("" : "+m"(x))
#define IC3_WAIT_VMEM() asm volatile("" ::: "memory")
#define IC3_WAIT_VMEM_N(n) asm volatile("" ::: "memory")
inline void __syncthreads() { ic3_host::tl_block->bar.arrive_and_wait(); }

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
// the lanes of a wavefront reach this point together (on the GPU they always do; here they are fibers): what one lane does
// behind it on behalf of the wave — a release of data all lanes wrote — sees every lane's part
inline void __builtin_amdgcn_wave_barrier() { ic3_host::tl_wave->bar.arrive_and_wait(); }
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
inline double __shfl(double v, int src);          // (defined below: visible to the templates' overload resolution)
inline unsigned __shfl(unsigned v, int src);
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
inline void ic3_host_buf_read(__amdgpu_buffer_rsrc_t r, int voff, int soff, void* out, int dwords);
inline ic3_host_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int)
{
    ic3_host_u32x4 v;
    ic3_host_buf_read(r, voff, soff, &v, 4);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// whole kernels: launches, the HIP runtime calls of the C ABI, the remaining device intrinsics
// ---------------------------------------------------------------------------------------------------------------------
namespace ic3_host {
#ifdef IC3_HOST_ASAN_FIBERS
extern "C" void __asan_poison_memory_region(void const volatile* addr, size_t size);
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
inline void lds_guard(void* p, size_t n) { __asan_poison_memory_region(p, n); }
inline void lds_unguard(void* p, size_t n) { __asan_unpoison_memory_region(p, n); }
#else
inline void lds_guard(void*, size_t) {}
inline void lds_unguard(void*, size_t) {}
#endif
template <class... P, class... A>
inline void launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds_bytes, hipStream_t, A&&... args)
{
    std::tuple<std::decay_t<P>...> params{ static_cast<std::decay_t<P>>(std::forward<A>(args))... };
    if constexpr (sizeof...(P) > 0) tl_kernarg = &std::get<0>(params);
    const int nt = (int)(block.x * block.y * block.z);
    if (nt <= 0 || lds_bytes > LDS_BYTES) std::abort();
    const int nwaves = (nt + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                // poison: what a kernel reads it must have written; and what lies behind the bytes the launch asked for must
                // stay untouched (checked behind the workgroup; under ASan every access there is reported at once)
                unsigned char* const lds = static_cast<unsigned char*>(dynamic_lds());
                lds_unguard(lds + lds_bytes, LDS_BYTES - lds_bytes);
                std::memset(lds, 0xCD, LDS_BYTES);
                lds_guard(lds + lds_bytes, LDS_BYTES - lds_bytes);
                Block blk(nt);
                std::vector<std::unique_ptr<Wave>> waves;
                for (int w = 0; w < nwaves; ++w) {
                    const int lanes = std::min(64, nt - 64 * w);
                    waves.emplace_back(new Wave(lanes));
                    for (int l = 0; l < 64; ++l) waves[w]->active[l] = l < lanes;
                }
                Fibers::get().run(nt, [&](int t) {
                        Wave* w = waves[t >> 6].get();
                        tl_lane = t & 63;
                        tl_wave = w;
                        tl_block = &blk;
                        tl_tid = ic3_host_dim3{ (unsigned)t % block.x, ((unsigned)t / block.x) % block.y,
                                                (unsigned)t / (block.x * block.y) };
                        tl_bid = ic3_host_dim3{ bx, by, bz };
                        tl_bdim = ic3_host_dim3{ block.x, block.y, block.z };
                        tl_gdim = ic3_host_dim3{ grid.x, grid.y, grid.z };
                        std::apply(kernel, params);
                        w->active[t & 63] = 0;
                        w->bar.arrive_and_drop();
                        blk.bar.arrive_and_drop();
                    });
                lds_unguard(lds + lds_bytes, LDS_BYTES - lds_bytes);
                for (size_t b = lds_bytes; b < LDS_BYTES; ++b)
                    if (lds[b] != 0xCD) Fibers::die("a workgroup wrote past the dynamic LDS its launch asked for");
            }
}
}  // namespace ic3_host
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    ic3_host::launch(kernel, grid, block, lds, stream, ##__VA_ARGS__)

enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
template <class T>
inline hipError_t hipMalloc(T** p, size_t bytes)
{
    *p = (T*)std::malloc(bytes ? bytes : 1);   // (ASan / UBSan watch its bounds)
    if (*p) std::memset(*p, 0xCD, bytes);      // poison: device memory is not zeroed either; what is read must have been written
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int device) { return device == -1 ? hipSuccess : hipErrorInvalidDevice; }   // the host "device"
inline hipError_t hipGetDevice(int* device) { *device = -1; return hipSuccess; }
template <class F>
inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
struct ic3_host_event {
    std::chrono::steady_clock::time_point t;
};
typedef ic3_host_event* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new ic3_host_event{ std::chrono::steady_clock::now() }; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
// (the host "device" runs every launch to completion inside the call: streams and waits are no-ops)
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}

template <class T>
inline T ic3_host_shfl_any(T v, int src)
{
    static_assert(sizeof(T) <= 8, "shuffle of up to 8 bytes");
    long long raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    raw = ic3_host::exchange(raw, src);
    std::memcpy(&v, &raw, sizeof(T));
    return v;
}
inline double __shfl(double v, int src) { return ic3_host_shfl_any(v, src); }
inline unsigned __shfl(unsigned v, int src) { return ic3_host_shfl_any(v, src); }
template <class T>
inline T __shfl_down(T v, int d) { return __shfl(v, ic3_host::tl_lane + d < 64 ? ic3_host::tl_lane + d : ic3_host::tl_lane); }
template <class T>
inline T __shfl_up(T v, int d) { return __shfl(v, ic3_host::tl_lane - d >= 0 ? ic3_host::tl_lane - d : ic3_host::tl_lane); }

inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class F, class I>
inline F ic3_host_atomic_add_fp(F* p, F v)
{
    I old, want;
    F cur;
    __atomic_load((I*)p, &old, __ATOMIC_RELAXED);
    do {
        std::memcpy(&cur, &old, sizeof(F));
        const F next = cur + v;
        std::memcpy(&want, &next, sizeof(F));
    } while (!__atomic_compare_exchange((I*)p, &old, &want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return cur;
}
inline float atomicAdd(float* p, float v) { return ic3_host_atomic_add_fp<float, uint32_t>(p, v); }
inline double atomicAdd(double* p, double v) { return ic3_host_atomic_add_fp<double, uint64_t>(p, v); }
inline int atomicMax(int* p, int v)
{
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __builtin_amdgcn_exp2f(float x) { return std::exp2(x); }
inline float __builtin_amdgcn_logf(float x) { return std::log2(x); }   // v_log_f32: base 2

// ---------------------------------------------------------------------------------------------------------------------
// the matrix-core kernels (policy_step.hip, gates_bwd.hip, commnet_fwd.hip): MFMA as a cross-lane operation with the
// hardware's operand / result layouts, raw buffer loads / stores with the descriptor's range check (per dword, VGPR +
// SGPR offset, unsigned — tools/exp/buf_probe.hip), the scalar helpers
// ---------------------------------------------------------------------------------------------------------------------
typedef float ic3_host_f32x16 __attribute__((ext_vector_type(16)));
typedef float ic3_host_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 ic3_host_bf16x8 __attribute__((ext_vector_type(8)));

namespace ic3_host {
// every live lane of the wave deposits `bytes` at wide[lane], then all read; two barriers like exchange()
template <class F>
inline void wave_gather(const void* mine, size_t bytes, F&& use)
{
    Wave* w = tl_wave;
    std::memcpy(w->wide[tl_lane], mine, bytes);
    w->bar.arrive_and_wait();
    use(w->wide);
    w->bar.arrive_and_wait();
}
}  // namespace ic3_host

// v_mfma_f32_32x32x2_f32: A[i][k] in lane 32 k + i, B[k][j] in lane 32 k + j; D[i][j]: lane 32 ((i / 4) % 2) + j, register 4 (i / 8) + i % 4
inline ic3_host_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, ic3_host_f32x16 c, int, int, int)
{
    const float ab[2] = { a, b };
    const int lane = ic3_host::tl_lane, j = lane & 31, hi = lane >> 5;
    ic3_host::wave_gather(ab, sizeof(ab), [&](unsigned char (*wide)[32]) {
        for (int r = 0; r < 16; ++r) {
            const int i = 8 * (r >> 2) + 4 * hi + (r & 3);
            float acc = c[r];
            for (int k = 0; k < 2; ++k) {
                float av, bv;
                std::memcpy(&av, wide[32 * k + i], 4);
                std::memcpy(&bv, wide[32 * k + j] + 4, 4);
                acc += av * bv;
            }
            c[r] = acc;
        }
    });
    return c;
}
// v_mfma_f32_16x16x4_f32: A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j; D[i][j]: lane 16 (i / 4) + j, register i % 4
inline ic3_host_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, ic3_host_f32x4 c, int, int, int)
{
    const float ab[2] = { a, b };
    const int lane = ic3_host::tl_lane, j = lane & 15, q = lane >> 4;
    ic3_host::wave_gather(ab, sizeof(ab), [&](unsigned char (*wide)[32]) {
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            float acc = c[r];
            for (int k = 0; k < 4; ++k) {
                float av, bv;
                std::memcpy(&av, wide[16 * k + i], 4);
                std::memcpy(&bv, wide[16 * k + j] + 4, 4);
                acc += av * bv;
            }
            c[r] = acc;
        }
    });
    return c;
}
// v_mfma_f32_32x32x16_bf16: A[i][8 h + e] = element e of lane 32 h + i, B[8 h + e][j] = element e of lane 32 h + j; D as 32x32x2
inline ic3_host_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(ic3_host_bf16x8 a, ic3_host_bf16x8 b, ic3_host_f32x16 c, int, int, int)
{
    unsigned char ab[32];
    std::memcpy(ab, &a, 16);
    std::memcpy(ab + 16, &b, 16);
    const int lane = ic3_host::tl_lane, j = lane & 31, hi = lane >> 5;
    auto bf = [](const unsigned char* p) {
        uint16_t u;
        std::memcpy(&u, p, 2);
        const uint32_t w = (uint32_t)u << 16;
        float f;
        std::memcpy(&f, &w, 4);
        return f;
    };
    ic3_host::wave_gather(ab, sizeof(ab), [&](unsigned char (*wide)[32]) {
        for (int r = 0; r < 16; ++r) {
            const int i = 8 * (r >> 2) + 4 * hi + (r & 3);
            float acc = c[r];
            for (int h = 0; h < 2; ++h)
                for (int e = 0; e < 8; ++e) acc += bf(wide[32 * h + i] + 2 * e) * bf(wide[32 * h + j] + 16 + 2 * e);
            c[r] = acc;
        }
    });
    return c;
}

inline void ic3_host_buf_read(__amdgpu_buffer_rsrc_t r, int voff, int soff, void* out, int dwords)
{
    std::memset(out, 0, 4 * (size_t)dwords);
    const uint32_t off = (uint32_t)voff + (uint32_t)soff;
    for (int d = 0; d < dwords; ++d)
        if ((uint64_t)off + 4 * d + 4 <= r.bytes) std::memcpy((char*)out + 4 * d, r.base + off + 4 * d, 4);
}
inline void ic3_host_buf_write(__amdgpu_buffer_rsrc_t r, int voff, int soff, const void* in, int dwords)
{
    const uint32_t off = (uint32_t)voff + (uint32_t)soff;
    for (int d = 0; d < dwords; ++d)   // out-of-range dwords are dropped, like the hardware does
        if ((uint64_t)off + 4 * d + 4 <= r.bytes) std::memcpy(const_cast<char*>(r.base) + off + 4 * d, (const char*)in + 4 * d, 4);
}
inline int __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voff, int soff, int)
{
    int v;
    ic3_host_buf_read(r, voff, soff, &v, 1);
    return v;
}
inline void __builtin_amdgcn_raw_buffer_store_b32(int v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int)
{
    ic3_host_buf_write(r, voff, soff, &v, 1);
}
inline void __builtin_amdgcn_raw_buffer_store_b128(ic3_host_u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int)
{
    ic3_host_buf_write(r, voff, soff, &v, 4);
}
inline int __builtin_amdgcn_readfirstlane(int v)
{
    ic3_host::Wave* w = ic3_host::tl_wave;
    int first = 0;
    while (first < 63 && !w->active[first]) ++first;
    return __shfl(v, first);
}
inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_s_setprio(int) {}
// a sleeping wave is a wave that SPINS on something another lane will write (policy_step_ws.hpp's role hand-offs through LDS
// counters): the emulated lane yields to the others; the atomics those hand-offs go through count as progress for the
// scheduler's deadlock detector (a round in which nobody finishes, completes a barrier or signals is a hang)
inline void __builtin_amdgcn_s_sleep(int) { ic3_host::yield(); }
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
template <class T> inline T ic3_host_atomic_load(T* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
template <class T, class V> inline T ic3_host_fetch_add(T* p, V v)
{
    ic3_host::Fibers::get().progressed();
    return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST);
}
#define __hip_atomic_load(p, order, scope) ic3_host_atomic_load(p)
#define __hip_atomic_fetch_add(p, v, order, scope) ic3_host_fetch_add(p, v)
inline void* __builtin_amdgcn_kernarg_segment_ptr() { return const_cast<void*>(ic3_host::tl_kernarg); }
inline unsigned long long __builtin_amdgcn_s_memrealtime() { return 0; }
inline unsigned __builtin_amdgcn_s_getreg(int) { return 0; }

// what the launch helpers of the matrix-core kernels ask the runtime
struct hipDeviceProp_t {
    int multiProcessorCount;
};
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int)
{
    prop->multiProcessorCount = 2;   // two "CUs": small tile plans still mix full and half tiles
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone, hipStreamCaptureStatusActive };
// every stream reports "being captured": the product then skips its on-device timing calibration (calibrate_tile_costs)
// and plans tiles with its documented fallback costs — timing a host emulation would mean nothing
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st)
{
    *st = hipStreamCaptureStatusActive;
    return hipSuccess;
}
