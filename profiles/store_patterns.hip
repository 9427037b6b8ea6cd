// store_patterns.hip - which pure store stream does the MI355X memory system take fastest?
// Standalone microbenchmark (no library code): hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip
// Every kernel writes the same `bytes` once per launch; GB/s = bytes / average launch time (HIP events,
// REP launches back to back).  Patterns:
//   simple      one 16-byte store per thread, 256-thread workgroups, grid = bytes / 4096 (what a fill does)
//   gridstride  persistent grid (WGPC workgroups per CU x T threads), 16 B per lane, chip-wide moving window
//   span        persistent grid, one workgroup per CU owns a contiguous span per "step" (the rollout writers'
//               pattern): NW writer waves, wave w takes tiles w, w+NW, ... of TILE bytes, 1 KiB per wave store;
//               K steps per launch, span of step k at k * step_bytes + wg * span_bytes; optional 8-byte
//               misalignment (odd agent counts) and 8-byte-per-lane stores (the row writer)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_simple(f32x4* out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    if (i < n4) out[i] = v;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void k_simple_unroll(f32x4* out, size_t n4) {
    // each workgroup writes UNROLL consecutive 4 KiB pieces
    const size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const size_t i = base + (size_t)u * 256;
        if (i < n4) out[i] = v;
    }
}

__global__ void k_gridstride(f32x4* out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) out[i] = v;
}

// span pattern.  All sizes in bytes; tile and span sizes are multiples of 8.
template <int LANE_BYTES>
__global__ void k_span(char* out, int K, size_t step_bytes, size_t span_bytes, int tile_bytes, int misalign, int nt) {
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tiles = (int)(span_bytes / tile_bytes);
    for (int k = 0; k < K; ++k) {
        char* span = out + (size_t)k * step_bytes + (size_t)blockIdx.x * span_bytes + misalign;
        for (int t = w; t < tiles; t += nw) {
            char* tile = span + (size_t)t * tile_bytes;
            if (LANE_BYTES == 16) {
                // 16-byte aligned body + 8-byte head/tail, like the LDS-tile writer
                const size_t addr = (size_t)tile;
                const int head = (addr & 8) ? 8 : 0;
                if (head && lane == 0) *reinterpret_cast<f32x2*>(tile) = f32x2{1.f, 2.f};
                const int body = (tile_bytes - head) & ~15;
                f32x4* dst = reinterpret_cast<f32x4*>(tile + head);
                const f32x4 v = {1.f, 2.f, 3.f, 4.f};
                for (int q = lane; q < body / 16; q += 64) {
                    if (nt) __builtin_nontemporal_store(v, &dst[q]); else dst[q] = v;
                }
                if (((tile_bytes - head) & 15) && lane == 63) *reinterpret_cast<f32x2*>(tile + head + body) = f32x2{1.f, 2.f};
            } else {
                f32x2* dst = reinterpret_cast<f32x2*>(tile);
                const f32x2 v = {1.f, 2.f};
                for (int q = lane; q < tile_bytes / 8; q += 64) dst[q] = v;
            }
        }
    }
}


// generalised span pattern: workgroup g owns `gpw` groups of `group_bytes`; group j sits at
// ((j * wgs + g) if strided else (g * gpw + j)) * group_bytes inside the step slot.  A group is cut into
// tiles of `tile` bytes, measured from the group start (absalign = 0: row-structured tiles) or on absolute
// multiples of `tile` (absalign = 1: line-aligned pieces, ragged at the group's ends); wave w takes tiles
// w, w + nw, ... of the group (order = 0) or a contiguous run of tiles (order = 1).
__global__ void k_groups(char* out, int K, size_t step_bytes, size_t group_bytes, int gpw, int strided, int tile,
                         int absalign, int order) {
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wgs = gridDim.x, g = blockIdx.x;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < gpw; ++j) {
            const size_t a = (size_t)k * step_bytes + (strided ? ((size_t)j * wgs + g) : ((size_t)g * gpw + j)) * group_bytes;
            const size_t b = a + group_bytes;
            const size_t org = absalign ? 0 : a;
            const long first = (long)((a - org) / tile), last = (long)((b - 1 - org) / tile);
            const long nt = last - first + 1;
            long m0 = first + w, m1 = last, ms = nw;
            if (order == 1) { const long per = (nt + nw - 1) / nw; m0 = first + w * per; m1 = m0 + per - 1 < last ? m0 + per - 1 : last; ms = 1; }
            for (long m = m0; m <= m1; m += ms) {
                size_t lo = org + (size_t)m * tile, hi = lo + tile;
                if (lo < a) lo = a;
                if (hi > b) hi = b;
                char* p = out + lo;
                int n = (int)(hi - lo);
                if (((size_t)p & 8) && n >= 8) { if (lane == 0) *reinterpret_cast<f32x2*>(p) = f32x2{1.f, 2.f}; p += 8; n -= 8; }
                f32x4* dst = reinterpret_cast<f32x4*>(p);
                for (int q = lane; q < n / 16; q += 64) dst[q] = v;
                if ((n & 8) && lane == 63) *reinterpret_cast<f32x2*>(p + (n & ~15)) = f32x2{1.f, 2.f};
            }
        }
}


// chunked dense window: the buffer is a sequence of chunks of `chunk` bytes dealt round-robin over the
// workgroups (chunk c -> workgroup c % wgs), so the chip writes a moving window of wgs * chunk bytes; inside
// a chunk the workgroup's waves interleave pieces of `piece` bytes (16-byte lane stores, 8-byte head/tail).
__global__ void k_chunks(char* out, size_t bytes, size_t chunk, int piece) {
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    const size_t nchunks = bytes / chunk;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t a = c * chunk, b = a + chunk;
        for (size_t lo = a + (size_t)w * piece; lo < b; lo += (size_t)nw * piece) {
            const size_t hi = lo + piece < b ? lo + piece : b;
            char* p = out + lo;
            int n = (int)(hi - lo);
            if (((size_t)p & 8) && n >= 8) { if (lane == 0) *reinterpret_cast<f32x2*>(p) = f32x2{1.f, 2.f}; p += 8; n -= 8; }
            f32x4* dst = reinterpret_cast<f32x4*>(p);
            for (int q = lane; q < n / 16; q += 64) dst[q] = v;
            if ((n & 8) && lane == 63) *reinterpret_cast<f32x2*>(p + (n & ~15)) = f32x2{1.f, 2.f};
        }
    }
}


// lean emulations of the rollout writers (27 x 4096, 16 envs per workgroup, NW writer waves), 16-byte lane stores:
//   mode 0: today - wave w streams whole envs w, w+NW, ... (17496 B each, 16-byte aligned, not line-aligned)
//   mode 1: each wave streams ONE contiguous, 1 KiB-aligned share of the workgroup's 279936-byte span
//   mode 2: waves interleave aligned 4 KiB runs of the span
//   mode 3: waves interleave aligned 1 KiB pieces
__global__ void k_roll(char* out, int K, int mode) {
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    constexpr unsigned SPAN = 16 * 17496, SPAN16 = SPAN / 16, ENV16 = 17496 / 16;   // 17496 = 16 * 1093.5: env starts alternate 0 / 8 mod 16
    for (int k = 0; k < K; ++k) {
        f32x4* span = reinterpret_cast<f32x4*>(out + ((size_t)k * gridDim.x + blockIdx.x) * SPAN);
        if (mode == 0) {
            for (int e = w; e < 16; e += nw) {
                // env e covers bytes [17496 e, 17496 (e+1)): odd e starts 8 bytes off a 16-byte boundary
                char* eb = reinterpret_cast<char*>(span) + (size_t)e * 17496;
                const int head = (e & 1) ? 8 : 0;
                if (head && lane == 0) *reinterpret_cast<f32x2*>(eb) = f32x2{1.f, 2.f};
                f32x4* d = reinterpret_cast<f32x4*>(eb + head);
                const unsigned n16 = (17496 - head) / 16;
                for (unsigned q = lane; q < n16; q += 64) d[q] = v;
                if (((17496 - head) & 8) && lane == 63) *reinterpret_cast<f32x2*>(eb + head + n16 * 16) = f32x2{1.f, 2.f};
            }
        } else if (mode == 1) {
            const unsigned pieces = (SPAN16 + 63) / 64;                   // 1 KiB pieces (last one short)
            const unsigned per = (pieces + nw - 1) / nw;
            const unsigned q0 = w * per * 64, q1 = min((w + 1) * per * 64, SPAN16);
            for (unsigned q = q0 + lane; q < q1; q += 64) span[q] = v;
        } else {
            const unsigned run = (mode == 2) ? 256 : 64;                  // 16-byte units per run
            for (unsigned r0 = w * run; r0 < SPAN16; r0 += nw * run)
                for (unsigned q = r0 + lane; q < min(r0 + run, SPAN16); q += 64) span[q] = v;
        }
    }
}


// cache-policy variants of the aligned contiguous-share pattern (mode 1 of k_roll) and of today's pattern
template <int POL>
__device__ __forceinline__ void st16(f32x4* p, f32x4 v) {
    if (POL == 0) *p = v;
    else if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}
template <int POL>
__global__ void k_roll_pol(char* out, int K) {
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    constexpr unsigned SPAN = 16 * 17496, SPAN16 = SPAN / 16;
    for (int k = 0; k < K; ++k) {
        f32x4* span = reinterpret_cast<f32x4*>(out + ((size_t)k * gridDim.x + blockIdx.x) * SPAN);
        const unsigned pieces = (SPAN16 + 63) / 64;
        const unsigned per = (pieces + nw - 1) / nw;
        const unsigned q0 = w * per * 64, q1 = min((w + 1) * per * 64, SPAN16);
        for (unsigned q = q0 + lane; q < q1; q += 64) st16<POL>(span + q, v);
    }
}


// "expander": a stateless kernel that turns per-env-step tables T[4N] = pos[N] | zeros[N-1] | shape[N] | ideal_vel |
// vel[N] (float2) into observations, one workgroup per aligned 16 x T bytes of the output, launched in address order
// like the `simple` fill.  Shows what the compose work costs on top of the best store stream.
template <int NC, int T>
__global__ __launch_bounds__(T) void k_expand(const f32x2* __restrict__ tables, f32x4* __restrict__ obs, size_t total_units) {
    constexpr unsigned ROWU = 3 * NC, ENVU = ROWU * NC, TBL = 4 * NC;
    constexpr unsigned NE = (2u * T + ENVU - 1) / ENVU + 1;                 // envs a workgroup's piece can touch
    __shared__ f32x2 tb[NE * TBL];
    const size_t U0 = (size_t)blockIdx.x * (2 * T);
    const size_t es0 = U0 / ENVU;
    const unsigned r0 = (unsigned)(U0 - es0 * ENVU);
    const size_t n_es = total_units / ENVU;
    for (unsigned t = threadIdx.x; t < NE * TBL; t += T)
        if (es0 + t / TBL < n_es) tb[t] = tables[es0 * TBL + t];
    __syncthreads();
    auto unit = [&](unsigned rel) -> f32x2 {
        const unsigned e = rel / ENVU, rem = rel - e * ENVU;
        const unsigned row = rem / ROWU, u = rem - row * ROWU;
        const f32x2* A = tb + e * TBL;
        const unsigned j = u - 1u;
        const bool is_delta = j < (unsigned)(NC - 1);
        unsigned idx = is_delta ? j + (j >= row ? 1u : 0u) : u;
        idx = (u == 0u) ? 3 * NC + row : idx;
        f32x2 val = A[idx];
        const f32x2 pi = A[row];
        if (is_delta) val -= pi;
        return val;
    };
    const unsigned rel = r0 + 2 * threadIdx.x;
    if (U0 + 2 * threadIdx.x + 1 < total_units) {
        const f32x2 a = unit(rel), b = unit(rel + 1);
        obs[(U0 >> 1) + threadIdx.x] = f32x4{a.x, a.y, b.x, b.y};
    }
}


// paced variants: does spacing a wave's store instructions (s_sleep) change what the memory system delivers?
template <int SLEEP>
__global__ void k_roll_paced(char* out, int K, int mode) {
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    constexpr unsigned SPAN = 16 * 17496, SPAN16 = SPAN / 16;
    for (int k = 0; k < K; ++k) {
        f32x4* span = reinterpret_cast<f32x4*>(out + ((size_t)k * gridDim.x + blockIdx.x) * SPAN);
        if (mode == 0) {
            for (int e = w; e < 16; e += nw) {
                char* eb = reinterpret_cast<char*>(span) + (size_t)e * 17496;
                const int head = (e & 1) ? 8 : 0;
                if (head && lane == 0) *reinterpret_cast<f32x2*>(eb) = f32x2{1.f, 2.f};
                f32x4* d = reinterpret_cast<f32x4*>(eb + head);
                const unsigned n16 = (17496 - head) / 16;
                for (unsigned q = lane; q < n16; q += 64) { d[q] = v; if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
                if (((17496 - head) & 8) && lane == 63) *reinterpret_cast<f32x2*>(eb + head + n16 * 16) = f32x2{1.f, 2.f};
            }
        } else if (mode == 1) {
            const unsigned pieces = (SPAN16 + 63) / 64;
            const unsigned per = (pieces + nw - 1) / nw;
            const unsigned q0 = w * per * 64, q1 = min((w + 1) * per * 64, SPAN16);
            for (unsigned q = q0 + lane; q < q1; q += 64) { span[q] = v; if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
        } else {
            const unsigned run = 256;
            for (unsigned r0 = w * run; r0 < SPAN16; r0 += nw * run)
                for (unsigned q = r0 + lane; q < min(r0 + run, SPAN16); q += 64) { span[q] = v; if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP); }
        }
    }
}

static float time_it(std::function<void()> fn, int rep) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) fn();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < rep; ++i) fn();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / rep;
}

int main(int argc, char** argv) {
    // the 27 x 4096 rollout: 20 steps x 256 workgroups x 16 envs x 17496 B
    const size_t env_bytes = 17496, span_bytes = 16 * env_bytes, step_bytes = 256 * span_bytes;
    const int K = 20;
    const size_t bytes = K * step_bytes;                      // 1.43 GB
    char* buf;
    CHECK(hipMalloc(&buf, bytes + 4096));
    CHECK(hipMemset(buf, 0, bytes + 4096));
    const int rep = 10;
    auto report = [&](const char* name, float ms, size_t b) {
        printf("%-64s %9.1f us  %7.0f GB/s\n", name, ms * 1e3, b / (ms * 1e-3) / 1e9);
        fflush(stdout);
    };
    const size_t n4 = bytes / 16;
    report("simple (1 x 16 B per thread, 256-thread WGs)", time_it([&] { hipLaunchKernelGGL(k_simple, dim3((n4 + 255) / 256), dim3(256), 0, 0, (f32x4*)buf, n4); }, rep), bytes);
    report("simple x4 (4 x 4 KiB per WG)", time_it([&] { hipLaunchKernelGGL(k_simple_unroll<4>, dim3((n4 + 1023) / 1024), dim3(256), 0, 0, (f32x4*)buf, n4); }, rep), bytes);
    report("simple x16", time_it([&] { hipLaunchKernelGGL(k_simple_unroll<16>, dim3((n4 + 4095) / 4096), dim3(256), 0, 0, (f32x4*)buf, n4); }, rep), bytes);
    CHECK(hipMemsetAsync(buf, 0, bytes, 0));
    report("hipMemsetAsync", time_it([&] { CHECK(hipMemsetAsync(buf, 1, bytes, 0)); }, rep), bytes);
    for (int wgpc : {1, 2, 4, 8})
        for (int T : {256, 512, 1024}) {
            if (wgpc * T > 2048) continue;
            char nm[128]; snprintf(nm, sizeof nm, "gridstride %d WG/CU x %d threads", wgpc, T);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_gridstride, dim3(256 * wgpc), dim3(T), 0, 0, (f32x4*)buf, n4); }, rep), bytes);
        }
    // span patterns (one WG per CU unless noted)
    struct Cfg { int nw, tile, mis, lane_bytes, nt, wgs; };
    std::vector<Cfg> cfgs;
    for (int nw : {2, 4, 8, 12, 16}) cfgs.push_back({nw, 5832, 0, 16, 0, 256});
    for (int nw : {4, 8}) cfgs.push_back({nw, 5832, 0, 8, 0, 256});
    for (int nw : {4, 8}) cfgs.push_back({nw, 17496, 0, 16, 0, 256});         // whole env per wave visit
    for (int nw : {4, 8}) cfgs.push_back({nw, 1944, 0, 16, 0, 256});          // 3 rows
    for (int nw : {4, 8}) cfgs.push_back({nw, 5832, 0, 16, 1, 256});          // nontemporal
    for (int nw : {4, 8}) cfgs.push_back({nw, 4096, 0, 16, 0, 256});          // line-aligned tiles (span padded)
    for (int nw : {4, 8}) cfgs.push_back({nw, 5832, 0, 16, 0, 512});          // two WGs per CU, 8 envs each
    for (int nw : {2, 4}) cfgs.push_back({nw, 5832, 0, 16, 0, 1024});         // four WGs per CU, 4 envs each
    for (const Cfg& c : cfgs) {
        const size_t span = (c.tile == 4096) ? ((span_bytes / 4096) * 4096) : span_bytes * 256 / c.wgs;
        const size_t stepb = span * c.wgs;
        const size_t total = (size_t)K * stepb;
        char nm[160];
        snprintf(nm, sizeof nm, "span: %4d WGs x %2d waves, tile %5d B, %2d B/lane%s%s", c.wgs, c.nw, c.tile, c.lane_bytes,
                 c.nt ? ", nontemporal" : "", c.mis ? ", +8 B" : "");
        auto fn = [&] {
            if (c.lane_bytes == 16) hipLaunchKernelGGL(k_span<16>, dim3(c.wgs), dim3(c.nw * 64), 0, 0, buf, K, stepb, span, c.tile, c.mis, c.nt);
            else hipLaunchKernelGGL(k_span<8>, dim3(c.wgs), dim3(c.nw * 64), 0, 0, buf, K, stepb, span, c.tile, c.mis, c.nt);
        };
        report(nm, time_it(fn, rep), total);
    }

    printf("-- generalised groups (27 x 4096 x 20 steps: 4096 envs of 17496 B per step)\n");
    struct G { int wgs, nw, envs_per_group, strided, tile, absalign, order; };
    std::vector<G> gs;
    for (int nw : {4, 8}) {
        gs.push_back({256, nw, 16, 0, 17496, 0, 0});     // the rollout writers today
        gs.push_back({256, nw, 16, 0, 5888, 1, 0});      // 46-line pieces
        gs.push_back({256, nw, 16, 0, 4096, 1, 0});
        gs.push_back({256, nw, 16, 0, 2048, 1, 0});
        gs.push_back({256, nw, 16, 0, 1024, 1, 0});
        gs.push_back({256, nw, 16, 0, 8192, 1, 0});
        gs.push_back({256, nw, 16, 0, 16384, 1, 0});
        gs.push_back({256, nw, 16, 0, 4096, 1, 1});      // each wave a contiguous quarter of the span
        gs.push_back({256, nw, 4, 1, 17496, 0, 0});      // 4-env groups dealt round-robin over the WGs
        gs.push_back({256, nw, 4, 1, 4096, 1, 0});
        gs.push_back({256, nw, 1, 1, 17496, 0, 0});      // single envs dealt round-robin
        gs.push_back({256, nw, 1, 1, 4096, 1, 0});
        gs.push_back({256, nw, 8, 1, 4096, 1, 0});
        gs.push_back({512, nw, 8, 0, 4096, 1, 0});
        gs.push_back({1024, nw, 4, 0, 4096, 1, 0});
    }
    for (const G& c : gs) {
        const size_t group_bytes = (size_t)c.envs_per_group * env_bytes;
        const int gpw = 4096 / c.envs_per_group / c.wgs;
        char nm[200];
        snprintf(nm, sizeof nm, "groups: %4d WGs x %d waves, %2d-env groups x %2d %s, tile %5d %s%s", c.wgs, c.nw, c.envs_per_group, gpw,
                 c.strided ? "strided" : "blocked", c.tile, c.absalign ? "abs-aligned" : "from group start", c.order ? ", contiguous per wave" : "");
        report(nm, time_it([&] { hipLaunchKernelGGL(k_groups, dim3(c.wgs), dim3(c.nw * 64), 0, 0, buf, K, step_bytes, group_bytes, gpw,
                                                    c.strided, c.tile, c.absalign, c.order); }, rep), bytes);
    }

    if (argc > 1)
    printf("-- chunks dealt round-robin over 256 workgroups (moving window = 256 x chunk)\n");
    for (int pass = 0; pass < (argc > 1 ? 2 : 0); ++pass)
        for (int nw : {4, 8})
            for (size_t chunk : {(size_t)4096, (size_t)16384, (size_t)17496, (size_t)17536, (size_t)34992, (size_t)65536, (size_t)69984, (size_t)139968, (size_t)279936})
                for (int piece : {1024, 4096}) {
                    if (piece == 4096 && chunk == 4096 && nw == 8) continue;
                    char nm[200];
                    snprintf(nm, sizeof nm, "chunks: 256 WGs x %d waves, chunk %6zu B, pieces of %4d B (pass %d)", nw, chunk, piece, pass);
                    report(nm, time_it([&] { hipLaunchKernelGGL(k_chunks, dim3(256), dim3(nw * 64), 0, 0, buf, bytes, chunk, piece); }, 30), bytes / chunk * chunk);
                }

    printf("-- lean emulation of the rollout writers (256 WGs, 20 steps)\n");
    for (int pass = 0; pass < 2; ++pass)
        for (int nw : {4, 6, 8})
            for (int mode : {0, 1, 2, 3}) {
                static const char* mn[] = {"whole envs per wave (today)", "one contiguous aligned share per wave", "aligned 4 KiB runs interleaved", "aligned 1 KiB pieces interleaved"};
                char nm[200];
                snprintf(nm, sizeof nm, "roll: %d writer waves, %s (pass %d)", nw, mn[mode], pass);
                report(nm, time_it([&] { hipLaunchKernelGGL(k_roll, dim3(256), dim3(nw * 64), 0, 0, buf, K, mode); }, 30), bytes);
            }

    printf("-- store cache policies on the aligned contiguous-share pattern (256 WGs x 4 waves, 20 steps)\n");
    for (int pass = 0; pass < 2; ++pass) {
        report("policy: plain", time_it([&] { hipLaunchKernelGGL(k_roll_pol<0>, dim3(256), dim3(256), 0, 0, buf, K); }, 30), bytes);
        report("policy: nt", time_it([&] { hipLaunchKernelGGL(k_roll_pol<1>, dim3(256), dim3(256), 0, 0, buf, K); }, 30), bytes);
        report("policy: sc1", time_it([&] { hipLaunchKernelGGL(k_roll_pol<2>, dim3(256), dim3(256), 0, 0, buf, K); }, 30), bytes);
        report("policy: sc0 sc1", time_it([&] { hipLaunchKernelGGL(k_roll_pol<3>, dim3(256), dim3(256), 0, 0, buf, K); }, 30), bytes);
        report("policy: sc0", time_it([&] { hipLaunchKernelGGL(k_roll_pol<4>, dim3(256), dim3(256), 0, 0, buf, K); }, 30), bytes);
        report("policy: sc0 sc1 nt", time_it([&] { hipLaunchKernelGGL(k_roll_pol<5>, dim3(256), dim3(256), 0, 0, buf, K); }, 30), bytes);
    }

    printf("-- expander (tables -> observations), 27 x 4096 x 20\n");
    {
        const size_t n_es = (size_t)K * 4096, tbl_bytes = n_es * 108 * 8;
        f32x2* tables;
        CHECK(hipMalloc(&tables, tbl_bytes + 8192));
        CHECK(hipMemset(tables, 0, tbl_bytes + 8192));
        const size_t total_units = bytes / 8;
        for (int pass = 0; pass < 2; ++pass) {
            report("expand<27>, 256-thread WGs (4 KiB each)", time_it([&] { hipLaunchKernelGGL((k_expand<27, 256>), dim3((total_units + 511) / 512), dim3(256), 0, 0, tables, (f32x4*)buf, total_units); }, 30), bytes);
            report("expand<27>, 64-thread WGs (1 KiB each)", time_it([&] { hipLaunchKernelGGL((k_expand<27, 64>), dim3((total_units + 127) / 128), dim3(64), 0, 0, tables, (f32x4*)buf, total_units); }, 30), bytes);
            report("expand<27>, 128-thread WGs (2 KiB each)", time_it([&] { hipLaunchKernelGGL((k_expand<27, 128>), dim3((total_units + 255) / 256), dim3(128), 0, 0, tables, (f32x4*)buf, total_units); }, 30), bytes);
            report("expand<27>, 512-thread WGs (8 KiB each)", time_it([&] { hipLaunchKernelGGL((k_expand<27, 512>), dim3((total_units + 1023) / 1024), dim3(512), 0, 0, tables, (f32x4*)buf, total_units); }, 30), bytes);
            report("expand<27>, 1024-thread WGs (16 KiB each)", time_it([&] { hipLaunchKernelGGL((k_expand<27, 1024>), dim3((total_units + 2047) / 2048), dim3(1024), 0, 0, tables, (f32x4*)buf, total_units); }, 30), bytes);
            report("simple again", time_it([&] { hipLaunchKernelGGL(k_simple, dim3((n4 + 255) / 256), dim3(256), 0, 0, (f32x4*)buf, n4); }, 30), bytes);
        }
        CHECK(hipFree(tables));
    }

    printf("-- paced store issue (s_sleep n = 64 n cycles after every 1 KiB store instruction), 256 WGs, 20 steps\n");
    for (int pass = 0; pass < 2; ++pass)
        for (int nw : {4, 8})
            for (int mode : {0, 1, 2}) {
                static const char* mn[] = {"whole envs per wave (today)", "one contiguous aligned share per wave", "aligned 4 KiB runs interleaved"};
                char nm[200];
#define PACED(SL) snprintf(nm, sizeof nm, "paced: %d waves, %s, s_sleep %d (pass %d)", nw, mn[mode], SL, pass); \
                report(nm, time_it([&] { hipLaunchKernelGGL(k_roll_paced<SL>, dim3(256), dim3(nw * 64), 0, 0, buf, K, mode); }, 30), bytes);
                PACED(0) PACED(1) PACED(2) PACED(3) PACED(4) PACED(6)
#undef PACED
            }

    printf("-- head to head: the same addresses from two kernels (k_chunks vs k_roll mode 2), alternating\n");
    for (int pass = 0; pass < 3; ++pass)
        for (int nw : {4, 8}) {
            char nm[200];
            snprintf(nm, sizeof nm, "h2h k_chunks chunk 279936 pieces 4096, %d waves (pass %d)", nw, pass);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_chunks, dim3(256), dim3(nw * 64), 0, 0, buf, bytes, (size_t)279936, 4096); }, 30), bytes);
            snprintf(nm, sizeof nm, "h2h k_roll mode 2 (aligned 4 KiB runs), %d waves (pass %d)", nw, pass);
            report(nm, time_it([&] { hipLaunchKernelGGL(k_roll, dim3(256), dim3(nw * 64), 0, 0, buf, K, 2); }, 30), bytes);
        }
    CHECK(hipDeviceSynchronize());
    CHECK(hipFree(buf));
    return 0;
}
