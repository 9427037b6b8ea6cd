// c_abi_rollout.cpp - libformation_hip.so driven from plain C++ through the C ABI alone (no Python, no
// PyTorch): device buffers from hipMalloc, the caller's own stream, status codes.  This is what a binding in
// any host language does; it doubles as a Python-free throughput check.
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/c_abi_rollout.cpp \
//         -L gym-formation_amd/lib -lformation_hip -Wl,-rpath,$PWD/gym-formation_amd/lib -o build/c_abi_rollout
//   ./build/c_abi_rollout [agents=27] [envs=4096] [steps=400]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "formation_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define FG_CHECK(x) do { int rc_ = (x); if (rc_ != FG_OK) { \
    fprintf(stderr, "libformation_hip: status %d: %s (%s:%d)\n", rc_, fg_last_error(), __FILE__, __LINE__); return 3; } } while (0)

template <typename T> static T* dmalloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) { fprintf(stderr, "hipMalloc of %zu bytes failed\n", n * sizeof(T)); exit(2); }
    return static_cast<T*>(p);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 27;
    const int B = argc > 2 ? atoi(argv[2]) : 4096;
    const int steps = argc > 3 ? atoi(argv[3]) : 400;
    const int K = 20;                                     // steps per rollout launch
    if (fg_abi_version() != FG_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    // formation_hd_env constants (core.py:119-139, formation_hd_env.py:13-33, environment.py:218-221)
    FgParams P = {};
    P.dt = 0.1f; P.damping = 0.25f; P.contact_force = 100.f; P.contact_margin = 1e-3f;
    P.sensitivity = 5.0f; P.mass = 1.0f; P.dist_min = 0.06f; P.collide_thresh = 0.03f;
    P.world_length = 100; P.auto_reset = 1; P.seed = 1; P.rng_offset = 0;

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    const size_t bn = (size_t)B * N, obs_env = (size_t)N * 6 * N;
    float *px = dmalloc<float>(bn), *py = dmalloc<float>(bn), *vx = dmalloc<float>(bn), *vy = dmalloc<float>(bn);
    float *shape = dmalloc<float>(bn * 2), *ivel = dmalloc<float>((size_t)B * 2);
    int32_t* step = dmalloc<int32_t>(B);
    float *act = dmalloc<float>((size_t)K * bn * 2);
    float *obs = dmalloc<float>((size_t)K * B * obs_env), *rew = dmalloc<float>((size_t)K * bn), *ind = dmalloc<float>((size_t)K * bn);
    uint8_t* done = dmalloc<uint8_t>((size_t)K * bn);

    // random policy: a pool of K steps of U(-1,1) actions from a host LCG, staged once
    std::vector<float> h_act((size_t)K * bn * 2);
    uint32_t lcg = 12345u;
    for (float& a : h_act) { lcg = lcg * 1664525u + 1013904223u; a = (float)(lcg >> 8) * (2.0f / 16777216.0f) - 1.0f; }
    HIP_OK(hipMemcpyAsync(act, h_act.data(), h_act.size() * sizeof(float), hipMemcpyHostToDevice, st));

    // Scenario.reset_world on the device, then the observation env.reset() returns
    FG_CHECK(fg_reset_hd(&P, B, N, nullptr, px, py, vx, vy, shape, ivel, step, st));
    FG_CHECK(fg_observe_hd(&P, B, N, px, py, vx, vy, shape, ivel, step, obs, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st));

    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    auto run = [&](bool rollout, int n) -> int {
        for (int t = 0; t < n; t += rollout ? K : 1) {
            P.rng_offset = (uint64_t)t + 1;
            if (rollout) FG_CHECK(fg_rollout_hd(&P, B, N, K, px, py, vx, vy, act, shape, ivel, step, obs, rew, ind, done, 1, st));
            else FG_CHECK(fg_step_hd(&P, B, N, px, py, vx, vy, act + (size_t)(t % K) * bn * 2, shape, ivel, step,
                                     obs, rew, ind, done, nullptr, nullptr, nullptr, st));
        }
        return 0;
    };
    const double bytes_step = (double)fg_step_hd_bytes(N) * B;
    for (int mode = 0; mode < 2; ++mode) {
        const bool rollout = mode == 0;
        const int n = rollout ? (steps / K) * K : steps;
        if (int rc = run(rollout, rollout ? 2 * K : 40)) return rc;             // warm-up
        HIP_OK(hipEventRecord(e0, st));
        if (int rc = run(rollout, n)) return rc;
        HIP_OK(hipEventRecord(e1, st));
        HIP_OK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %d agents x %d envs: %8.3f us/step  %.3e env-steps/s  %6.0f GB/s algorithmic\n",
               rollout ? "fg_rollout_hd (20 per launch)" : "fg_step_hd (1 per launch)", N, B, ms * 1e3 / n,
               (double)B * n / (ms * 1e-3), bytes_step * n / (ms * 1e-3) / 1e9);
    }

    // The same rollout into a PLACED observation buffer: address space backed by separately created
    // physical chunks, the buffer composed of chunks spread over the whole arena in shuffled order (fg_arena_map),
    // everything else handed back (fg_arena_trim).  (formation_gym/placement.py additionally times several such selections and keeps the best.)
    {
        size_t free_b = 0, total_b = 0;
        HIP_OK(hipMemGetInfo(&free_b, &total_b));
        const uint64_t need = (uint64_t)K * B * obs_env * sizeof(float);
        uint64_t chunk = 32ull << 20;
        while (chunk < (1ull << 30) && need / chunk > 16) chunk <<= 1;
        uint64_t arena_bytes = (uint64_t)(0.6 * (double)free_b);
        if (arena_bytes > (192ull << 30)) arena_bytes = 192ull << 30;
        void *arena = nullptr, *placed = nullptr;
        uint32_t n = 0;
        if (need >= (256ull << 20) && arena_bytes >= 2 * need &&
            fg_arena_create(0, arena_bytes, chunk, &arena, &chunk, &n) == FG_OK) {
            const uint32_t W = (uint32_t)((need + chunk - 1) / chunk);
            std::vector<uint32_t> idx(W);
            for (uint32_t j = 0; j < W; ++j) idx[j] = (uint32_t)(((uint64_t)j * n + n / 2) / W);   // one chunk per stratum
            for (uint32_t j = W - 1; j > 0; --j) { lcg = lcg * 1664525u + 1013904223u; std::swap(idx[j], idx[(lcg >> 8) % (j + 1)]); }
            FG_CHECK(fg_arena_map(arena, idx.data(), W, &placed));
            FG_CHECK(fg_arena_trim(arena));                               // every other chunk back to the driver
            float* obs_placed = static_cast<float*>(placed);
            FgParams Q = P;
            Q.obs_placed = 1;
            auto run_placed = [&](int n_steps) -> int {
                for (int t = 0; t < n_steps; t += K) {
                    Q.rng_offset = (uint64_t)t + 1;
                    FG_CHECK(fg_rollout_hd(&Q, B, N, K, px, py, vx, vy, act, shape, ivel, step, obs_placed, rew, ind, done, 1, st));
                }
                return 0;
            };
            const int n_steps = (steps / K) * K;
            if (int rc = run_placed(2 * K)) return rc;
            HIP_OK(hipEventRecord(e0, st));
            if (int rc = run_placed(n_steps)) return rc;
            HIP_OK(hipEventRecord(e1, st));
            HIP_OK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-28s %d agents x %d envs: %8.3f us/step  %.3e env-steps/s  %6.0f GB/s algorithmic  (%u chunks of %llu MiB out of %u)\n",
                   "fg_rollout_hd, placed buffer", N, B, ms * 1e3 / n_steps, (double)B * n_steps / (ms * 1e-3),
                   bytes_step * n_steps / (ms * 1e-3) / 1e9, W, (unsigned long long)(chunk >> 20), n);
            HIP_OK(hipStreamSynchronize(st));
            FG_CHECK(fg_arena_destroy(arena));
        } else {
            printf("placed-buffer run skipped (buffer below the Infinity Cache size, or no room for an arena)\n");
        }
    }

    // sanity on the last observation of env 0: finite, relative positions antisymmetric, zero block zero
    std::vector<float> h_obs(obs_env), h_rew(N);
    HIP_OK(hipMemcpy(h_obs.data(), obs, obs_env * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_rew.data(), rew, N * sizeof(float), hipMemcpyDeviceToHost));
    int bad = 0;
    for (float x : h_obs) bad += !std::isfinite(x);
    auto rel = [&](int i, int j, int c) { return h_obs[(size_t)i * 6 * N + 2 + 2 * (j < i ? j : j - 1) + c]; };   // p_j - p_i
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
            if (i != j) for (int c = 0; c < 2; ++c) bad += std::fabs(rel(i, j, c) + rel(j, i, c)) > 2e-6f;
    for (int i = 0; i < N; ++i)
        for (int u = 2 * N; u < 4 * N - 2; ++u) bad += h_obs[(size_t)i * 6 * N + u] != 0.0f;
    for (int i = 1; i < N; ++i) bad += h_rew[i] != h_rew[0];                   // shared reward, broadcast
    bad += !(h_rew[0] < 0.0f);
    printf("%s\n", bad ? "sanity FAILED" : "sanity ok");
    for (void* p : {(void*)px, (void*)py, (void*)vx, (void*)vy, (void*)shape, (void*)ivel, (void*)step, (void*)act,
                    (void*)obs, (void*)rew, (void*)ind, (void*)done}) (void)hipFree(p);
    (void)hipStreamDestroy(st);
    return bad ? 4 : 0;
}
