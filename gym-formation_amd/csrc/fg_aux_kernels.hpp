// fg_aux_kernels.hpp - Reset kernels (counter RNG, bit-exact MT19937) and the small landmark-scenario kernel.
// Part of libformation_hip (gfx950); included by formation_hip.hip, one translation unit.
#ifndef FG_AUX_KERNELS_HPP_
#define FG_AUX_KERNELS_HPP_

#include "fg_common.hpp"
#include "fg_pair_loops.hpp"

namespace fg {

// ---------------------------------------------------------------------------
// standalone masked reset (Scenario.reset_world, formation_hd_env.py:77-95)
// ---------------------------------------------------------------------------
template <int G, int T>
__global__ __launch_bounds__(T) void reset_kernel(const Args a, const uint8_t* mask) {
    constexpr int E = T / G;
    __shared__ float scratch[64];
    const int N = a.N;
    const int tid = threadIdx.x;
    const int e = (G >= T) ? 0 : tid / G;
    const int i = (G >= T) ? tid : tid % G;
    const int b = blockIdx.x * E + e;
    const bool valid = (b < a.B) && (i < N);
    const bool mine = valid && (mask == nullptr || mask[b] != 0);
    uint32_t c[4] = {(uint32_t)(b + a.p.env_index_base), (uint32_t)i, (uint32_t)rng_base(a.p), (uint32_t)(rng_base(a.p) >> 32)};
    philox4x32(c, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
    float raw[2] = {valid ? u_pm1(c[2]) : 0.f, valid ? u_pm1(c[3]) : 0.f};
    const float rx = raw[0], ry = raw[1];
    env_reduce<G, T, 2, R_SUM, R_SUM, R_SUM, R_SUM>(raw, scratch);
    if (mine) {
        const size_t sidx = (size_t)b * N + i;
        const float invN = a.inv_n;
        a.px[sidx] = u_pm1(c[0]); a.py[sidx] = u_pm1(c[1]);
        a.vx[sidx] = 0.f; a.vy[sidx] = 0.f;
        reinterpret_cast<float2*>(a.shape)[sidx] = make_float2(__builtin_fmaf(-raw[0], invN, rx), __builtin_fmaf(-raw[1], invN, ry));
        if (i == 0) {
            uint32_t c2[4] = {(uint32_t)(b + a.p.env_index_base), 0xFFFFFFFFu, (uint32_t)rng_base(a.p), (uint32_t)(rng_base(a.p) >> 32)};
            philox4x32(c2, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
            reinterpret_cast<float2*>(a.ivel)[b] = make_float2(u_pm1(c2[0]), u_pm1(c2[1]));
            if (a.step) a.step[b] = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// World.update_agent_state (core.py:279-286): state.c = action.c + c_noise * N(0,1), zeros for a silent agent
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void update_comm_kernel(const FgParams p, int B, int N, const float2* __restrict__ action_c,
                                                          float2* __restrict__ comm) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * N) return;
    const int b = (int)(t / N), i = (int)(t - (long long)b * N);
    const float c_noise = p.agent_props ? p.agent_props[(size_t)i * FG_AGENT_PROPS + 5] : 0.0f;
    float2 c = make_float2(0.f, 0.f);
    if (c_noise >= 0.0f) {
        c = action_c[t];
        if (c_noise > 0.0f) {                                  // its own counter stream: agent index | 0x40000000
            const real2 n = motor_noise(p.seed, (uint32_t)(b + p.env_index_base), (uint32_t)i | 0x40000000u, rng_base(p));
            c.x += c_noise * n.x; c.y += c_noise * n.y;
        }
    }
    comm[t] = c;
}

// The same for any dim_c (core.py:279-286 takes whatever World.dim_c is; every scenario file of the reference sets 2): one
// thread per (env, agent, pair of components); pair 0 draws the very noise update_comm_kernel draws, pair q > 0 its own
// counter stream (agent index | 0x40000000 | q << 12).
__global__ __launch_bounds__(256) void update_comm_dim_kernel(const FgParams p, int B, int N, int dim_c,
                                                              const float* __restrict__ action_c, float* __restrict__ comm) {
    const int pairs = (dim_c + 1) >> 1;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * N * pairs) return;
    const long long bi = t / pairs;
    const int q = (int)(t - bi * pairs);
    const int b = (int)(bi / N), i = (int)(bi - (long long)b * N);
    const float c_noise = p.agent_props ? p.agent_props[(size_t)i * FG_AGENT_PROPS + 5] : 0.0f;
    const size_t o = (size_t)bi * dim_c + 2 * q;
    const bool two = 2 * q + 1 < dim_c;
    float c0 = 0.f, c1 = 0.f;
    if (c_noise >= 0.0f) {
        c0 = action_c[o];
        if (two) c1 = action_c[o + 1];
        if (c_noise > 0.0f) {
            const real2 n = motor_noise(p.seed, (uint32_t)(b + p.env_index_base), (uint32_t)i | 0x40000000u | ((uint32_t)q << 12),
                                        rng_base(p));
            c0 += c_noise * n.x; c1 += c_noise * n.y;
        }
    }
    comm[o] = c0;
    if (two) comm[o + 1] = c1;
}

// ---------------------------------------------------------------------------
// Bit-exact reset on device: Scenario.reset_world (formation_hd_env.py:77-95) drawing from the
// env's own legacy NumPy MT19937 stream (environment.py:106-110 seeds it), so that multi-episode
// rollouts keep matching the reference without a host round trip.  One workgroup per env; the
// 624-word state lives in LDS, is tempered / twisted in parallel and written back.
//   draw order: N agent positions, N landmark positions, ideal velocity, two doubles each;
//   double = ((a >> 5) * 2^26 + (b >> 6)) / 2^53 from two 32-bit outputs; U(-1,1) = -1 + 2 d.
// mt_state: uint32 [B][626] = key[624], pos, unused.
// ---------------------------------------------------------------------------
// The next `count` tempered 32-bit outputs of the env's MT19937 stream -> outs[0 .. count) (LDS), by the 256 threads of the
// workgroup: `mt` (LDS, 624 words) is loaded from gstate[0 .. 624), regenerated as often as needed; returns the new position
// (the caller writes `mt` and the position back).  NumPy's legacy generator word for word (randomkit.c: genrand).
__device__ __forceinline__ int mt_generate_words(uint32_t* mt, uint32_t* outs, const uint32_t* gstate, int count, int tid) {
    for (int q = tid; q < 624; q += 256) mt[q] = gstate[q];
    int pos = (int)gstate[624];
    __syncthreads();
    int produced = 0;
    auto mix = [](uint32_t a, uint32_t b2) -> uint32_t {
        const uint32_t y = (a & 0x80000000u) | (b2 & 0x7fffffffu);
        return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    };
    while (produced < count) {
        if (pos >= 624) {                              // regenerate the 624 words (three dependent thirds)
            uint32_t nv = 0;
            if (tid < 227) nv = mt[tid + 397] ^ mix(mt[tid], mt[tid + 1]);
            __syncthreads();
            if (tid < 227) mt[tid] = nv;
            __syncthreads();
            if (tid < 227) nv = mt[tid] ^ mix(mt[tid + 227], mt[tid + 228]);            // kk = tid + 227
            __syncthreads();
            if (tid < 227) mt[tid + 227] = nv;
            __syncthreads();
            if (tid < 169) nv = mt[tid + 227] ^ mix(mt[tid + 454], mt[tid + 455]);      // kk = tid + 454 .. 622
            const uint32_t old623 = mt[623];
            __syncthreads();
            if (tid < 169) mt[tid + 454] = nv;
            __syncthreads();
            if (tid == 0) mt[623] = mt[396] ^ mix(old623, mt[0]);
            __syncthreads();
            pos = 0;
        }
        const int take = min(624 - pos, count - produced);
        for (int q = tid; q < take; q += 256) {
            uint32_t y = mt[pos + q];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            outs[produced + q] = y;
        }
        produced += take; pos += take;
        __syncthreads();
    }
    return pos;
}

// Which envs reset: those whose mask byte is set (mask != NULL), every env (mask NULL, world_length <= 0), or - the
// vec-env worker's rule, decided on the device without a mask upload or a host read-back - those whose episode is over,
// step[b] >= world_length (mask NULL, world_length > 0; env_wrappers.py:14-18).  With `obs` the workgroup also writes the
// RESET observation of its env (formation_hd_env.py:52-59 on the fresh state: what the worker returns), the same bits
// fg_observe_hd gives, so that a vec-env step needs no second pass over the whole batch.
__global__ __launch_bounds__(256) void mt_reset_kernel(int B, int N, const uint8_t* __restrict__ mask,
                                                       uint32_t* __restrict__ mt_state,
                                                       float* px, float* py, float* vx, float* vy,
                                                       float* shape, float* ivel, float* lm_pos, int32_t* step,
                                                       int world_length, float* obs, long long obs_pitch2) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u32[];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= B || (mask && !mask[b])) return;
    if (!mask && world_length > 0 && step[b] < world_length) return;
    uint32_t* const mt = lds_u32;                      // [624]
    uint32_t* const outs = lds_u32 + 624;              // [8N + 4] tempered outputs
    double* const dsum = reinterpret_cast<double*>(lds_u32 + 624 + ((8 * N + 4 + 1) & ~1));   // [2] mean of raw
    float2* const P = reinterpret_cast<float2*>(dsum + 2);   // [N] fresh positions, [N] ideal shape, [1] ideal velocity
    float2* const S = P + N;
    float2* const IV = S + N;
    uint32_t* const gstate = mt_state + (size_t)b * 626;
    const int M = 8 * N + 4;
    int pos = mt_generate_words(mt, outs, gstate, M, tid);
    auto draw = [&](int m) -> double {                 // m-th double of this reset
        const double a = (double)(outs[2 * m] >> 5), c = (double)(outs[2 * m + 1] >> 6);
        return -1.0 + 2.0 * ((a * 67108864.0 + c) / 9007199254740992.0);
    };
    if (tid == 0) {                                    // np.mean over axis 0: rows added in order
        double sx = 0.0, sy = 0.0;
        for (int i = 0; i < N; ++i) { sx += draw(2 * N + 2 * i); sy += draw(2 * N + 2 * i + 1); }
        dsum[0] = sx / (double)N; dsum[1] = sy / (double)N;
    }
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        const size_t o = (size_t)b * N + i;
        const float2 pp = make_float2((float)draw(2 * i), (float)draw(2 * i + 1));
        px[o] = pp.x; py[o] = pp.y;
        vx[o] = 0.f; vy[o] = 0.f;
        const double rx = draw(2 * N + 2 * i), ry = draw(2 * N + 2 * i + 1);
        const float2 ss = make_float2((float)(rx - dsum[0]), (float)(ry - dsum[1]));
        shape[2 * o] = ss.x; shape[2 * o + 1] = ss.y;
        if (lm_pos) { lm_pos[2 * o] = (float)rx; lm_pos[2 * o + 1] = (float)ry; }
        P[i] = pp; S[i] = ss;
    }
    if (tid == 0) {
        const float2 iv = make_float2((float)draw(4 * N), (float)draw(4 * N + 1));
        ivel[2 * b] = iv.x; ivel[2 * b + 1] = iv.y;
        IV[0] = iv;
        if (step) step[b] = 0;
        gstate[624] = (uint32_t)pos;
    }
    for (int q = tid; q < 624; q += 256) gstate[q] = mt[q];
    if (obs) {                                         // the reset observation: [0 | p_j - p_i (j != i) | 0 .. | shape | ideal_vel]
        __syncthreads();
        float2* const out = reinterpret_cast<float2*>(obs) + (size_t)b * (size_t)obs_pitch2;
        const unsigned n3 = 3u * (unsigned)N, total = n3 * (unsigned)N;
        for (unsigned q = tid; q < total; q += 256) {
            const unsigned row = q / n3, u = q - row * n3;
            float2 val = make_float2(0.f, 0.f);
            if (u >= 1u && u < (unsigned)N) {
                const unsigned j = u - 1u, idx = j + (j >= row ? 1u : 0u);
                const float2 a2 = P[idx], c2 = P[row];
                val = make_float2(a2.x - c2.x, a2.y - c2.y);
            } else if (u >= 2u * N - 1u && u < n3 - 1u) {
                val = S[u - (2u * N - 1u)];
            } else if (u == n3 - 1u) {
                val = IV[0];
            }
            out[q] = val;
        }
    }
}

// ---------------------------------------------------------------------------
// Landmark scenarios (N + M <= 1024 movable entities): basic_formation_env (BASELINE config 1),
// formation_hd_partial_env, formation_hd_partial_range_env, formation_hd_obs_env.
// One lane per movable entity (N agents, then M obstacles), one env per aligned group of G
// lanes of a wave (N + M <= 64) or per workgroup of G threads (beyond).  Reference lines under formation_gym/envs/:
//   basic     observation basic_formation_env.py:29-41, reward :43-52 (self "collision" included)
//   partial   observation formation_hd_partial_env.py:38-57 (ring neighbours), reward :59-72
//   range     observation formation_hd_partial_range_env.py:38-52 (clipped), reward as partial
//   obstacle  observation formation_hd_obs_env.py:44-58, reward :60-99 incl. the obstacle
//             velocity override (:84-89); obstacles are movable colliders of World.step
// ---------------------------------------------------------------------------
struct ScnArgs {
    FgParams p;
    FgScenario sc;
    int B, N, do_phys;
    float* px; float* py; float* vx; float* vy;
    const float* act; float* lm; float* opos; float* ovel; int32_t* step;
    float* obs; float* rew; float* indiv; uint8_t* done; int32_t* near_ag;
    int stage;     // compose the workgroup's observation rows in LDS and stream them out as ONE contiguous span
    int K;         // steps per launch (fg_rollout_scenario; 1 otherwise): act / reward / indiv / done / near_ag [K][B]...,
    int obs_every; // obs [K / obs_every][B][N][D]
    float coll_scale;     // per-agent tables (FgParams.agent_props): penalty distance of a pair = coll_scale * (size_a + size_b)
    float inv_n, inv_l;   // 1 / N, 1 / L, correctly rounded on the host: the run-time-count kernel and the one-env-per-lane
                          // kernels (compile-time counts) must multiply by the very same values (cf. Args.inv_n)
};

// Scenario.reset_world of these scenarios from the device counter RNG (basic_formation_env.py:54-65,
// formation_hd_partial_env.py:88-99, formation_hd_partial_range_env.py:76-87, formation_hd_obs_env.py:101-114): agents and landmarks U(-1,1)^2, velocities zero,
// obstacle k from U([s_k, 2.0], [s_k+1, 2.5]) with s = linspace(-1.8, 1.8, M + 1), falling at the scenario's velocity.
// One Philox block per entity, counter (global env index, entity code, per-launch offset); entity code = agent index,
// 0x10000000 | landmark index, 0x20000000 | obstacle index (formation_hd_env's reset uses the agent indices and
// 0xFFFFFFFF the same way).  Distributional parity with the reference's MT19937 draws, as for formation_hd_env.
constexpr uint32_t SCN_LANDMARK_CODE = 0x10000000u, SCN_OBSTACLE_CODE = 0x20000000u;
__device__ __forceinline__ float2 scn_fresh_pm1(const FgParams& P, int b, uint32_t code, uint64_t off) {
    uint32_t c[4] = {(uint32_t)(b + P.env_index_base), code, (uint32_t)off, (uint32_t)(off >> 32)};
    philox4x32(c, (uint32_t)P.seed, (uint32_t)(P.seed >> 32));
    return make_float2(u_pm1(c[0]), u_pm1(c[1]));
}
__device__ __forceinline__ float2 scn_fresh_obstacle(const FgParams& P, int b, int k, int M, uint64_t off) {
    const float2 r = scn_fresh_pm1(P, b, SCN_OBSTACLE_CODE | (uint32_t)k, off);
    const float lo = -1.8f + 3.6f * (float)k / (float)M, hi = -1.8f + 3.6f * (float)(k + 1) / (float)M;
    return make_float2(lo + (hi - lo) * (0.5f * r.x + 0.5f), 2.0f + 0.5f * (0.5f * r.y + 0.5f));
}

// standalone masked reset of the landmark scenarios (mask NULL = every env): the draws scn_kernel's fused auto-reset makes
__global__ __launch_bounds__(256) void scn_reset_kernel(const FgParams P, const FgScenario sc, int B, int N,
                                                        const uint8_t* __restrict__ mask,
                                                        float* px, float* py, float* vx, float* vy,
                                                        float2* lm, float2* opos, float2* ovel, int32_t* step) {
    const int L = sc.num_landmarks, M = sc.num_obstacles, per = N + L + M;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * per) return;
    const int b = (int)(t / per), r = (int)(t - (long long)b * per);
    if (mask && !mask[b]) return;
    const uint64_t off = rng_base(P);
    if (r < N) {
        const float2 q = scn_fresh_pm1(P, b, (uint32_t)r, off);
        const size_t o = (size_t)b * N + r;
        px[o] = q.x; py[o] = q.y; vx[o] = 0.f; vy[o] = 0.f;
        if (r == 0 && step) step[b] = 0;
    } else if (r < N + L) {
        lm[(size_t)b * L + (r - N)] = scn_fresh_pm1(P, b, SCN_LANDMARK_CODE | (uint32_t)(r - N), off);
    } else {
        const int k = r - N - L;
        opos[(size_t)b * M + k] = scn_fresh_obstacle(P, b, k, M, off);
        ovel[(size_t)b * M + k] = make_float2(sc.obstacle_vx, sc.obstacle_vy);
    }
}

// Scenario.reset_world of the landmark scenarios from the env's own legacy MT19937 stream, bit-exact with the host path
// (basic_formation_env.py:54-65, formation_hd_partial_env.py:88-99, formation_hd_partial_range_env.py:76-87,
// formation_hd_obs_env.py:101-120): N agent positions, L landmark positions, each -1 + 2 u; then M obstacles from
// np.random.uniform([s_k, 2.0], [s_k+1, 2.5]) = low + (high - low) u with s = np.linspace(-1.8, 1.8, M + 1), velocity (ovx, ovy);
// agent velocities zero, step counter zero.  Which envs: mt_reset_kernel's rule (mask / everybody / step >= world_length).
__global__ __launch_bounds__(256) void mt_reset_scn_kernel(int B, int N, int L, int M, const uint8_t* __restrict__ mask,
                                                           uint32_t* __restrict__ mt_state,
                                                           float* px, float* py, float* vx, float* vy,
                                                           float2* lm, float2* opos, float2* ovel, int32_t* step,
                                                           int world_length, float ovx, float ovy) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u32[];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= B || (mask && !mask[b])) return;
    if (!mask && world_length > 0 && step[b] < world_length) return;
    uint32_t* const mt = lds_u32;                      // [624]
    uint32_t* const outs = lds_u32 + 624;              // [4 (N + L + M)] tempered outputs
    uint32_t* const gstate = mt_state + (size_t)b * 626;
    const int words = 4 * (N + L + M);
    const int pos = mt_generate_words(mt, outs, gstate, words, tid);
    auto unit = [&](int m) -> double {                 // m-th double of this reset, in [0, 1)
        const double a = (double)(outs[2 * m] >> 5), c = (double)(outs[2 * m + 1] >> 6);
        return (a * 67108864.0 + c) / 9007199254740992.0;
    };
    for (int i = tid; i < N; i += 256) {
        const size_t o = (size_t)b * N + i;
        px[o] = (float)(-1.0 + 2.0 * unit(2 * i)); py[o] = (float)(-1.0 + 2.0 * unit(2 * i + 1));
        vx[o] = 0.f; vy[o] = 0.f;
    }
    for (int l = tid; l < L; l += 256)
        lm[(size_t)b * L + l] = make_float2((float)(-1.0 + 2.0 * unit(2 * (N + l))), (float)(-1.0 + 2.0 * unit(2 * (N + l) + 1)));
    for (int k = tid; k < M; k += 256) {
        // np.linspace(-1.8, 1.8, M + 1): arange(M + 1) * (3.6 / M) + (-1.8), the last point set to the stop value
        // (every product rounded before its sum, as NumPy's C code does: no fused multiply-add here)
        const double dstep = (1.8 - (-1.8)) / (double)M;
        const double lo = __dadd_rn(__dmul_rn((double)k, dstep), -1.8);
        const double hi = (k + 1 == M) ? 1.8 : __dadd_rn(__dmul_rn((double)(k + 1), dstep), -1.8);
        const double x = __dadd_rn(lo, __dmul_rn(hi - lo, unit(2 * (N + L + k))));
        const double y = __dadd_rn(2.0, __dmul_rn(2.5 - 2.0, unit(2 * (N + L + k) + 1)));
        opos[(size_t)b * M + k] = make_float2((float)x, (float)y);
        ovel[(size_t)b * M + k] = make_float2(ovx, ovy);
    }
    if (tid == 0) {
        if (step) step[b] = 0;
        gstate[624] = (uint32_t)pos;
    }
    for (int q = tid; q < 624; q += 256) gstate[q] = mt[q];
}

template <int G, int T>
__global__ __launch_bounds__(T) void scn_kernel(const ScnArgs a) {
    constexpr int E = T / G;
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int N = a.N, L = a.sc.num_landmarks, M = a.sc.num_obstacles, NE = N + M;
    const int kind = a.sc.kind;
    const int tid = threadIdx.x;
    const int e = tid / G, i = tid % G;
    const int b = blockIdx.x * E + e;
    const bool live = b < a.B;
    const bool is_agent = live && i < N;
    const bool is_obst = live && i >= N && i < NE;
    constexpr int SCR = (G > 64) ? 32 : 0;            // G > 64 (one env per workgroup): cross-wave partials of env_reduce
    float* const scratch = reinterpret_cast<float*>(smem);
    float2* const tables = smem + SCR;
    float2* const PRE = tables + e * (2 * NE + L);
    float2* const POST = PRE + NE;
    float2* const LM = POST + NE;
    float2 p = make_float2(0.f, 0.f), v = p;
    const size_t sidx = (size_t)b * N + i;
    const size_t oidx = (size_t)b * M + (i - N);
    if (is_agent) {
        p = make_float2(a.px[sidx], a.py[sidx]);
        v = make_float2(a.vx[sidx], a.vy[sidx]);
    } else if (is_obst) {
        p = reinterpret_cast<const float2*>(a.opos)[oidx];
        v = reinterpret_cast<const float2*>(a.ovel)[oidx];
    }
    if (is_agent || is_obst) { PRE[i] = p; POST[i] = p; }
    for (int l = i; live && l < L; l += G) LM[l] = reinterpret_cast<const float2*>(a.lm)[(size_t)b * L + l];
    int t_step = (live && a.step) ? a.step[b] : 0;
    __syncthreads();
    // agents of different mass / size / accel / max_speed / u_noise (FgParams.agent_props; core.py:45-109): the lane's own row;
    // its partners' mass and size are read from the table in the pair loops (a cold path: no reference scenario has them).
    // The obstacles keep the scenario's size and Entity's default mass 1 (formation_hd_obs_env.py:36-42).
    // Column 6 of the table = the agent's flags (core.py:54-58), honoured as step_kernel's option path does: a pair needs both
    // to collide (:292-293); against an immovable partner the force is taken as it is, not scaled by the mass ratio (:319-321);
    // an immovable agent is not integrated (:266-267); a ghost passes through soft walls (:326-327); the penalties of an agent
    // that does not collide are not counted (`if agent.collide:` in every reward callback).  The obstacles are ordinary colliders.
    const bool het = a.p.agent_props != nullptr;
    const AgentProps me = agent_props_of(a.p, i, het && i < N);
    const int my_flags = (het && i < N) ? me.flags : 0;
    const float my_size = i < N ? (het ? me.size : 0.5f * a.p.dist_min) : 0.5f * (2.0f * a.sc.obstacle_size);
    const float my_mass = het ? (i < N ? me.mass : 1.0f) : a.p.mass;
    const int KS = a.K > 1 ? a.K : 1;
    float2 u_next = make_float2(0.f, 0.f);              // the action of step ks + 1 is fetched while step ks runs
    if (a.do_phys && is_agent) u_next = reinterpret_cast<const float2*>(a.act)[sidx];
    // K steps in one launch (fg_rollout_scenario): the state stays in registers / LDS, every step's reward, done and (every
    // obs_every-th) observation go to their slab - the same arithmetic as K single-step launches, bit for bit
    const uint64_t rbase = rng_base(a.p);               // read once: no load from the device counter inside the step loop
    for (int ks = 0; ks < KS; ++ks) {
    const uint64_t off = rbase + (uint64_t)ks;
    const size_t kb = (size_t)ks * a.B;                 // slab of step ks in the [K][B]... outputs
    const float2 u_now = u_next;
    if (a.do_phys && is_agent && ks + 1 < KS) u_next = reinterpret_cast<const float2*>(a.act)[(kb + a.B) * N + sidx];
    if (a.do_phys) {
        if (is_agent || is_obst) {
            // World.step: all pairs of movable colliders, contact distance size_i + size_j
            float fx = 0.f, fy = 0.f;
            const float k = a.p.contact_margin;
            // (the loops of this kernel run over a handful of entities with run-time counts: four LDS reads are issued
            // ahead of their use, index clamped, so that a wave waits once per four partners instead of once per partner;
            // the order of the sums is the ascending-j order of core.py:240-262 either way)
            for (int j0 = 0; j0 < NE; j0 += 4) {
                float2 qq[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) qq[t] = PRE[min(j0 + t, NE - 1)];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int j = j0 + t;
                    const float2 q = qq[t];
                    float size_j = 0.5f * (j < N ? a.p.dist_min : 2.0f * a.sc.obstacle_size);
                    int fj = 0;
                    if (het && j < N) {
                        size_j = a.p.agent_props[(size_t)j * FG_AGENT_PROPS + 1];
                        fj = (int)a.p.agent_props[(size_t)j * FG_AGENT_PROPS + 6];
                    }
                    const float dmin = my_size + size_j;
                    const float cut = dmin + 18.0f * k;
                    const float dx = p.x - q.x, dy = p.y - q.y;
                    const float d2 = dx * dx + dy * dy;
                    if (j < NE && j != i && d2 < cut * cut && !((fj | my_flags) & FG_AGENT_NO_COLLIDE)) {
                        const float d = __builtin_amdgcn_sqrtf(d2);
                        const float x = (dmin - d) / k;
                        const float pen = k * (fmaxf(x, 0.0f) + __logf(1.0f + __expf(-fabsf(x))));
                        float c = a.p.contact_force * pen * __builtin_amdgcn_rcpf(d);
                        if (het && !(fj & FG_AGENT_IMMOVABLE))
                            c = ((j < N ? a.p.agent_props[(size_t)j * FG_AGENT_PROPS] : 1.0f) / my_mass) * c;   // core.py:314-317
                        fx += dx * c; fy += dy * c;
                    }
                }
            }
            if (is_agent) {
                const float2 u = u_now;
                const float2 fa = action_force(a.p, me, u, (uint32_t)(b + a.p.env_index_base), (uint32_t)i, off);
                fx += fa.x; fy += fa.y;
            }
            if (a.p.num_walls > 0) wall_forces(a.p, p, my_size, fx, fy, (my_flags & FG_AGENT_GHOST) != 0);
            if (!(my_flags & FG_AGENT_IMMOVABLE)) {
                v.x = v.x * (1.0f - a.p.damping) + (fx / my_mass) * a.p.dt;
                v.y = v.y * (1.0f - a.p.damping) + (fy / my_mass) * a.p.dt;
                if (is_agent) v = clamp_speed(me.max_speed, v);
                p.x += v.x * a.p.dt; p.y += v.y * a.p.dt;
            }
            POST[i] = p;
            if (is_agent) {
                a.px[sidx] = p.x; a.py[sidx] = p.y; a.vx[sidx] = v.x; a.vy[sidx] = v.y;
            } else {
                // the reward callback re-arms the obstacle velocity every step (:84-89)
                const bool falling = p.y > a.sc.obstacle_floor;
                v = make_float2(falling ? a.sc.obstacle_vx : 0.f, falling ? a.sc.obstacle_vy : 0.f);   // what the next step loads
                reinterpret_cast<float2*>(a.opos)[oidx] = p;
                reinterpret_cast<float2*>(a.ovel)[oidx] = v;
            }
        }
        t_step += 1;
        __syncthreads();
    }
    // ---- formation term ----
    float form = 0.f;      // basic: sum_l min_a |p_a - l| ; others: Hausdorff(centred agents, centred landmarks)
    if (kind == FG_SCN_BASIC) {
        float cover = 0.f;
        for (int l0 = 0; l0 < L; l0 += G) {
            const int l = l0 + i;
            if (live && l < L) {
                const float2 m = LM[l];
                float best = INFINITY; int barg = 0;
                for (int j0 = 0; j0 < N; j0 += 4) {
                    float2 qq[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) qq[t] = POST[min(j0 + t, N - 1)];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float dx = qq[t].x - m.x, dy = qq[t].y - m.y, d2 = dx * dx + dy * dy;
                        if (j0 + t < N && d2 < best) { best = d2; barg = j0 + t; }
                    }
                }
                cover += sqrtf(best);
                if (a.near_ag) a.near_ag[(kb + b) * L + l] = barg;
            }
        }
        float red[1] = {cover};
        env_reduce<G, T, 1, R_SUM, R_SUM, R_SUM, R_SUM>(red, scratch);
        form = red[0];
    } else {
        float s4[4] = {is_agent ? p.x : 0.f, is_agent ? p.y : 0.f, 0.f, 0.f};
        for (int l = i; live && l < L; l += G) { s4[2] += LM[l].x; s4[3] += LM[l].y; }
        env_reduce<G, T, 4, R_SUM, R_SUM, R_SUM, R_SUM>(s4, scratch);
        const float mx = s4[0] * a.inv_n, my = s4[1] * a.inv_n;
        const float lx = s4[2] * a.inv_l, ly = s4[3] * a.inv_l;
        float rowmin = -INFINITY, colmax = -INFINITY;
        if (is_agent) {                                         // min over landmarks for my agent
            rowmin = INFINITY;
            for (int l0 = 0; l0 < L; l0 += 4) {
                float2 mm[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) mm[t] = LM[min(l0 + t, L - 1)];
#pragma unroll
                for (int t = 0; t < 4; ++t) {                  // a clamped repeat of the last landmark does not change a minimum
                    const float dx = (p.x - mx) - (mm[t].x - lx), dy = (p.y - my) - (mm[t].y - ly);
                    rowmin = fminf(rowmin, dx * dx + dy * dy);
                }
            }
        }
        for (int l = i; live && l < L; l += G) {                // min over agents for my landmark(s)
            float cm = INFINITY;
            const float2 ml = LM[l];
            for (int j0 = 0; j0 < N; j0 += 4) {
                float2 qq[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) qq[t] = POST[min(j0 + t, N - 1)];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float dx = (qq[t].x - mx) - (ml.x - lx), dy = (qq[t].y - my) - (ml.y - ly);
                    cm = fminf(cm, dx * dx + dy * dy);
                }
            }
            colmax = fmaxf(colmax, cm);
        }
        float red[2] = {rowmin, colmax};
        env_reduce<G, T, 2, R_MAX, R_MAX, R_MAX, R_MAX>(red, scratch);
        form = sqrtf(fmaxf(red[0], red[1]));
    }
    // ---- collision counts ----
    int cnt = 0;
    if (is_agent) {
        const float thr = a.p.collide_thresh, thr2 = (float)((double)thr * (double)thr);
        for (int j0 = 0; j0 < N; j0 += 4) {
            float2 qq[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) qq[t] = POST[min(j0 + t, N - 1)];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = j0 + t;
                const float dx = qq[t].x - p.x, dy = qq[t].y - p.y;
                float t2 = thr2;
                if (het && j < N) {                               // is_collision per pair: dist < size_a + size_b
                    const float tj = a.coll_scale * (my_size + a.p.agent_props[(size_t)j * FG_AGENT_PROPS + 1]);
                    t2 = tj * tj;
                }
                cnt += (j < N && (kind == FG_SCN_BASIC || j != i) && dx * dx + dy * dy < t2) ? 1 : 0;
            }
        }
        const float ot = (het ? my_size : 0.5f * a.p.dist_min) + a.sc.obstacle_size, ot2 = (float)((double)ot * (double)ot);
        for (int j = N; j < NE; ++j) {
            const float dx = POST[j].x - p.x, dy = POST[j].y - p.y;
            cnt += (dx * dx + dy * dy < ot2) ? 1 : 0;
        }
    }
    if (my_flags & FG_AGENT_NO_COLLIDE) cnt = 0;
    float cs[1] = {(float)cnt};
    env_reduce<G, T, 1, R_SUM, R_SUM, R_SUM, R_SUM>(cs, scratch);
    const bool is_done = t_step >= a.p.world_length;
    // ---- outputs ----
    const int nbr = (kind == FG_SCN_PARTIAL) ? a.sc.num_obs : (N - 1);
    const int D = 2 + (kind == FG_SCN_BASIC ? 2 : 0) + 2 * L + 2 * M + 2 * nbr + 2 * (N - 1);
    if (is_agent) {
        if (a.rew) a.rew[kb * N + sidx] = (float)(-(double)N * (double)form - (double)a.sc.penalty * (double)cs[0]);
        if (a.indiv) a.indiv[kb * N + sidx] = -form - a.sc.penalty * (float)cnt;
        if (a.done) a.done[kb * N + sidx] = is_done ? 1 : 0;
    }
    if (a.p.auto_reset && a.do_phys) {                  // uniform over the launch
        // the vec-env worker's rule (env_wrappers.py:14-18): an env whose episode is over restarts at once, and the
        // observation returned with the finished step's reward / done is the RESET observation
        __syncthreads();                                // every lane has finished reading POST / LM of the finished step
        if (live && is_done) {
            if (is_agent) {
                p = scn_fresh_pm1(a.p, b, (uint32_t)i, off); v = make_float2(0.f, 0.f);
                POST[i] = p;
                a.px[sidx] = p.x; a.py[sidx] = p.y; a.vx[sidx] = 0.f; a.vy[sidx] = 0.f;
            } else if (is_obst) {
                p = scn_fresh_obstacle(a.p, b, i - N, M, off);
                v = make_float2(a.sc.obstacle_vx, a.sc.obstacle_vy);
                POST[i] = p;
                reinterpret_cast<float2*>(a.opos)[oidx] = p;
                reinterpret_cast<float2*>(a.ovel)[oidx] = v;
            }
            for (int l = i; l < L; l += G) {
                const float2 m = scn_fresh_pm1(a.p, b, SCN_LANDMARK_CODE | (uint32_t)l, off);
                LM[l] = m;
                reinterpret_cast<float2*>(a.lm)[(size_t)b * L + l] = m;
            }
            t_step = 0;
        }
        __syncthreads();
    }
    const bool want_obs = a.obs_every <= 1 || (ks + 1) % a.obs_every == 0;
    const size_t ob = (size_t)(a.obs_every > 1 ? ks / a.obs_every : ks) * a.B;   // slab of this step's observation
    if (is_agent && want_obs) {
        // every lane composes its own row: straight to global memory (rows D floats apart: one 8-byte piece per lane
        // and instruction), or into the workgroup's LDS image of its [E][N][D] block, which all lanes then copy out
        // with consecutive 8-byte stores (a.stage; 16 x 65536 obstacle envs: 203 -> see profiles/r02_aux_kernels.md)
        float2* const stage0 = tables + E * (2 * NE + L);                              // behind the last env's tables
        float2* o = a.stage ? stage0 + (size_t)(e * N + i) * (D / 2) : reinterpret_cast<float2*>(a.obs + (ob * N + sidx) * D);
        int w = 0;
        o[w++] = v;
        if (kind == FG_SCN_BASIC) o[w++] = p;
        // a segment of `count` units, unit t = get(t): four sources are read before the four stores (the staged row lives in
        // LDS like the tables, so a read behind a store would have to wait for it)
        auto emit = [&](int count, auto&& get) {
            for (int t0 = 0; t0 < count; t0 += 4) {
                float2 r[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) r[t] = get(min(t0 + t, count - 1));
#pragma unroll
                for (int t = 0; t < 4; ++t) if (t0 + t < count) o[w + t] = r[t];
                w += min(4, count - t0);
            }
        };
        const bool basic = kind == FG_SCN_BASIC;
        emit(L, [&](int l) { const float2 m = LM[l]; return basic ? make_float2(m.x - p.x, m.y - p.y) : m; });
        emit(M, [&](int t) { const float2 q = POST[N + t]; return make_float2(q.x - p.x, q.y - p.y); });
        if (kind == FG_SCN_PARTIAL) {
            emit(nbr, [&](int kk) {
                int j = i + 1 + kk;                            // (i + 1 + kk) mod N
                while (j >= N) j -= N;
                const float2 q = POST[j];
                return make_float2(q.x - p.x, q.y - p.y);
            });
        } else {
            const float r = (kind == FG_SCN_RANGE) ? a.sc.obs_range : INFINITY;
            emit(N - 1, [&](int t) {
                const float2 q = POST[t < i ? t : t + 1];     // the t-th OTHER agent, index order
                return make_float2(fminf(fmaxf(q.x - p.x, -r), r), fminf(fmaxf(q.y - p.y, -r), r));
            });
        }
        for (int j = 0; j < N - 1; ++j) o[w++] = make_float2(0.f, 0.f);
    }
    if (a.stage && want_obs) {                          // want_obs is uniform over the launch
        __syncthreads();
        const float2* const img = tables + E * (2 * NE + L);
        const int b0 = blockIdx.x * E;
        const int El = min(E, a.B - b0);
        const int units = El * N * (D / 2);
        float2* const out = reinterpret_cast<float2*>(a.obs + (ob + (size_t)b0) * N * D);
        for (int q0 = tid; q0 < units; q0 += 4 * T) {
            float2 r[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) r[t] = img[min(q0 + t * T, units - 1)];
#pragma unroll
            for (int t = 0; t < 4; ++t) if (q0 + t * T < units) out[q0 + t * T] = r[t];
        }
    }
    if (ks + 1 < KS) {                                  // the next step starts from this one's end state
        __syncthreads();                                // POST read by everyone, the staged image copied out
        if (is_agent || is_obst) PRE[i] = p;
        __syncthreads();
    }
    }   // steps
    if (a.do_phys && a.step && live && i == 0) a.step[b] = t_step;
}

// MultiAgentEnv._set_action for the non-default action modes (environment.py:187-215): one lane
// per agent, raw u out (the step kernels scale by the sensitivity).
__global__ __launch_bounds__(256) void decode_actions_kernel(int mode, int64_t count, void* action, float2* u_out) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= count) return;
    float2 u = make_float2(0.f, 0.f);
    if (mode == FG_ACT_ONEHOT5) {
        const float* a = reinterpret_cast<const float*>(action) + g * 5;
        u = make_float2(a[1] - a[2], a[3] - a[4]);
    } else if (mode == FG_ACT_INDEX) {
        const int k = reinterpret_cast<const int32_t*>(action)[g];
        u.x = (k == 1) ? -1.0f : (k == 2) ? 1.0f : 0.0f;
        u.y = (k == 3) ? -1.0f : (k == 4) ? 1.0f : 0.0f;
    } else {
        float2* a = reinterpret_cast<float2*>(action) + g;
        const float2 v = *a;
        u = (v.y > v.x) ? make_float2(0.f, 1.f) : make_float2(1.f, 0.f);   // np.argmax: first maximum; NaN first wins
        if (v.x != v.x) u = make_float2(1.f, 0.f);
        else if (v.y != v.y) u = make_float2(0.f, 1.f);
        *a = u;
    }
    u_out[g] = u;
}

}  // namespace fg

#endif  // FG_AUX_KERNELS_HPP_
