// mall_split.hip - does a per-store cache hint keep PART of a repeatedly rewritten buffer resident in the 256 MiB Infinity
// Cache (MALL)?  A buffer of S bytes is rewritten by every launch (as a rollout buffer is): the first C bytes with plain
// stores, the rest with stores carrying a hint (nt / sc1 / sc0 sc1).  If the hinted stores bypass the MALL, the plain part
// (C < 256 MiB) stays resident, its rewrites cost no HBM bandwidth and the launch gets faster as C grows towards the
// cache size; if not, time does not depend on C.   hipcc --offload-arch=gfx950 -O3 -o mall_split mall_split.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int HINT>
__device__ __forceinline__ void st(f32x4* p, f32x4 v) {
    if (HINT == 0) *p = v;
    else if (HINT == 1) __builtin_nontemporal_store(v, p);
    else if (HINT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if (HINT == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}

// persistent grid, 1 KiB pieces dealt round-robin over all waves (the dense-fill pattern)
template <int HINT>
__global__ void k_fill(f32x4* out, size_t n16, size_t plain16) {
    const size_t nw = (size_t)gridDim.x * (blockDim.x >> 6);
    const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (size_t q = w * 64 + lane; q < n16; q += nw * 64) {
        if (q < plain16) st<0>(out + q, v); else st<HINT>(out + q, v);
    }
}

int main() {
    const size_t MB = 1000 * 1000;
    const size_t sizes[] = {331 * MB, 78 * MB * 20};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (size_t S : sizes) {
        f32x4* buf;
        CHECK(hipMalloc(&buf, S + 4096));
        CHECK(hipMemset(buf, 0, S));
        const size_t n16 = S / 16;
        for (int pass = 0; pass < 2; ++pass)
            for (int hint = 1; hint <= 4; ++hint)
                for (size_t C : {(size_t)0, 64 * MB, 128 * MB, 192 * MB, 224 * MB, 256 * MB, S}) {
                    if (C > S) continue;
                    auto launch = [&]() {
                        switch (hint) {
                            case 1: hipLaunchKernelGGL(k_fill<1>, dim3(256), dim3(256), 0, 0, buf, n16, C / 16); break;
                            case 2: hipLaunchKernelGGL(k_fill<2>, dim3(256), dim3(256), 0, 0, buf, n16, C / 16); break;
                            case 3: hipLaunchKernelGGL(k_fill<3>, dim3(256), dim3(256), 0, 0, buf, n16, C / 16); break;
                            default: hipLaunchKernelGGL(k_fill<4>, dim3(256), dim3(256), 0, 0, buf, n16, C / 16); break;
                        }
                    };
                    for (int i = 0; i < 5; ++i) launch();
                    CHECK(hipEventRecord(e0));
                    for (int i = 0; i < 30; ++i) launch();
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    printf("pass %d  S %5zu MB  hint %s  plain part %5zu MB  %.1f us  %.2f TB/s\n", pass, S / MB,
                           hint == 1 ? "nt       " : hint == 2 ? "sc1      " : hint == 3 ? "sc0 sc1  " : "sc0 sc1 nt", C / MB, ms / 30 * 1e3,
                           (double)S / (ms / 30 * 1e-3) / 1e12);
                    fflush(stdout);
                }
        CHECK(hipFree(buf));
    }
    return 0;
}
