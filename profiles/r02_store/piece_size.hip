// r02_store_piece_size.hip - pure-store microbenchmark: the buffer as consecutive pieces of P bytes dealt round-robin
// over ALL waves of a persistent grid (256 workgroups x NW waves); a wave writes its piece sequentially with 1 KiB
// store instructions.  P = 1 KiB is the dense fill; P = 17 496 is "one whole N = 27 observation block per wave".
// Which piece sizes / alignments keep the dense-fill rate?      hipcc --offload-arch=gfx950 -O3 -o build/piece_size ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_piece(char* out, size_t bytes, int P) {
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t npieces = bytes / P;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t pc = (size_t)blockIdx.x * nw + w; pc < npieces; pc += (size_t)gridDim.x * nw) {
        char* p = out + pc * P;
        int n = P;
        if (((size_t)p & 8) && n >= 8) { if (lane == 0) *reinterpret_cast<f32x2*>(p) = f32x2{1.f, 2.f}; p += 8; n -= 8; }
        f32x4* dst = reinterpret_cast<f32x4*>(p);
        for (int q = lane; q < n / 16; q += 64) dst[q] = v;
        if ((n & 8) && lane == 63) *reinterpret_cast<f32x2*>(p + (n & ~15)) = f32x2{1.f, 2.f};
    }
}

int main() {
    const size_t bytes = (size_t)20 * 4096 * 17496;
    char* buf;
    CHECK(hipMalloc(&buf, bytes + 65536));
    CHECK(hipMemset(buf, 0, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int sizes[] = {1024, 2048, 4096, 5832, 6144, 8192, 16384, 17496, 17536, 18432, 32768, 65536, 279936, 262144};
    for (int pass = 0; pass < 2; ++pass)
        for (int nw : {4, 8})
            for (int P : sizes) {
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_piece, dim3(256), dim3(nw * 64), 0, 0, buf, bytes, P);
                CHECK(hipEventRecord(e0));
                for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_piece, dim3(256), dim3(nw * 64), 0, 0, buf, bytes, P);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("pass %d  waves/wg %d  piece %7d B  %.1f us  %.2f TB/s\n", pass, nw, P, ms / 20 * 1e3, (double)(bytes / P * P) / (ms / 20 * 1e-3) / 1e12);
            }
    return 0;
}
