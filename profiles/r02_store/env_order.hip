// r02_store_env_order.hip - pure-store microbenchmark (no library code): in which ORDER should whole observation
// blocks ("envs", 17 496 B at N = 27, or padded to 17 536 B = 137 lines) be written by 256 workgroups x NW waves?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/env_order profiles/r02_store_env_order.hip && /tmp/env_order
// Every pattern writes K slabs of B envs (the rollout buffer [K][B][env]); a wave always writes ONE whole env at a time
// with 1 KiB store instructions (16 B per lane, 8-byte head / tail where the env starts on an odd 8-byte unit):
//   blocked   workgroup g owns envs g*E .. g*E+E-1, wave w takes e = w, w+NW, ...          (the rollout kernel today)
//   wg_rr     workgroup g owns envs j*G+g (j < E), wave w takes j = w, w+NW, ...            (envs dealt over workgroups)
//   wave_rr   env index = (j*G + g)*NW + w                                                  (envs dealt over all waves)
//   piece_rr  the slab as 1 KiB pieces dealt over all waves (no env structure; what a dense fill does)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void write_env(char* p, int n, int lane) {
    if (((size_t)p & 8) && n >= 8) { if (lane == 0) *reinterpret_cast<f32x2*>(p) = f32x2{1.f, 2.f}; p += 8; n -= 8; }
    f32x4* dst = reinterpret_cast<f32x4*>(p);
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (int q = lane; q < n / 16; q += 64) dst[q] = v;
    if ((n & 8) && lane == 63) *reinterpret_cast<f32x2*>(p + (n & ~15)) = f32x2{1.f, 2.f};
}

__global__ void k_env(char* out, int K, int B, int E, size_t pitch, int env_bytes, int mode) {
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int G = gridDim.x, g = blockIdx.x;
    for (int k = 0; k < K; ++k) {
        char* slab = out + (size_t)k * B * pitch;
        if (mode == 3) {
            const size_t pieces = (size_t)B * pitch / 1024;
            f32x4* dst = reinterpret_cast<f32x4*>(slab);
            const f32x4 v = {1.f, 2.f, 3.f, 4.f};
            for (size_t pc = (size_t)g * nw + w; pc < pieces; pc += (size_t)G * nw) dst[pc * 64 + lane] = v;
            continue;
        }
        for (int j = w; j < E; j += nw) {
            long b;
            if (mode == 0) b = (long)g * E + j;
            else if (mode == 1) b = (long)j * G + g;
            else b = ((long)(j / nw) * G + g) * nw + w;
            if (b < B) write_env(slab + (size_t)b * pitch, env_bytes, lane);
        }
    }
}

int main() {
    const int K = 20, B = 4096, G = 256, E = 16;
    const size_t max_bytes = (size_t)K * B * 17536 + 4096;
    char* buf;
    CHECK(hipMalloc(&buf, max_bytes));
    CHECK(hipMemset(buf, 0, max_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[4] = {"blocked", "wg_rr", "wave_rr", "piece_rr"};
    for (int pass = 0; pass < 2; ++pass)
        for (int nw : {1, 2, 4})
            for (int pitch : {17496, 17536})
                for (int fill_pad = 0; fill_pad < (pitch == 17496 ? 1 : 2); ++fill_pad)      // 17 536: also write the 40 pad bytes (whole lines)
                for (int mode = 0; mode < 4; ++mode) {
                    const int env_bytes = fill_pad ? pitch : 17496;
                    const double bytes = (double)K * B * 17496;                                  // useful bytes only
                    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_env, dim3(G), dim3(nw * 64), 0, 0, buf, K, B, E, (size_t)pitch, env_bytes, mode);
                    CHECK(hipEventRecord(e0));
                    const int REP = 20;
                    for (int i = 0; i < REP; ++i) hipLaunchKernelGGL(k_env, dim3(G), dim3(nw * 64), 0, 0, buf, K, B, E, (size_t)pitch, env_bytes, mode);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    printf("pass %d  waves/wg %d  pitch %d  pad %s  %-9s  %.1f us/launch  %.2f TB/s (useful bytes)\n", pass, nw, pitch, fill_pad ? "written" : "skipped", names[mode], ms / REP * 1e3, bytes / (ms / REP * 1e-3) / 1e12);
                }
    return 0;
}
