// buf_probe.hip — what does the gfx950 buffer range check cover?  policy_step_kernel issues its obs zero stores
// unconditionally and relies on the hardware dropping the ones past the tile's slice (num_records); this checks
// (1) that an out-of-range buffer_store is dropped, (2) whether the SGPR offset takes part in the check.
//   hipcc --offload-arch=gfx950 -O3 buf_probe.hip -o buf_probe && ./buf_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* buf, unsigned in_range_bytes, int use_soffset)
{
    const unsigned long long b = (unsigned long long)buf;
    i4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)in_range_bytes);
    r[3] = 0x00020000;
    f4 v = { 1.f, 1.f, 1.f, 1.f };
    const int lane16 = threadIdx.x * 16;
    // 8 wave stores of 1 KiB at offsets 0, 1 KiB, ...; only the first in_range_bytes may land
    for (int i = 0; i < 8; ++i) {
        if (use_soffset) {
            const int so = __builtin_amdgcn_readfirstlane(i * 1024);
            asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt" ::"v"(v), "v"(lane16), "s"(r), "s"(so) : "memory");
        } else {
            const int vo = lane16 + i * 1024;
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt" ::"v"(v), "v"(vo), "s"(r) : "memory");
        }
    }
}

int main()
{
    float* d;
    const size_t n = 8 * 256;
    hipMalloc(&d, n * 4);
    for (int mode = 0; mode < 2; ++mode) {
        for (unsigned lim : { 3072u, 3072u + 512u }) {
            hipMemset(d, 0, n * 4);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, lim, mode);
            hipDeviceSynchronize();
            std::vector<float> h(n);
            hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
            size_t written = 0, last = 0;
            for (size_t i = 0; i < n; ++i)
                if (h[i] != 0.f) {
                    ++written;
                    last = i;
                }
            printf("%s offset, num_records %u B: %zu bytes written, last byte %zu -> %s\n", mode ? "SGPR" : "VGPR", lim,
                   written * 4, last * 4 + 3, written * 4 == lim ? "range check covers it" : "NOT range-checked");
        }
    }
    return 0;
}
