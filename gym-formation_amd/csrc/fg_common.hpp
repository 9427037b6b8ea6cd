// fg_common.hpp - Shared device-side definitions: kernel argument block, LDS layout of one environment,
// in-register butterfly reductions, counter RNG, World options.
// Part of libformation_hip (gfx950); included by formation_hip.hip, one translation unit.
#ifndef FG_COMMON_HPP_
#define FG_COMMON_HPP_


#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "formation_hip.h"


namespace fg {

#define FG_DEV __device__ __forceinline__
// s_setprio level of the rollout producer waves: their dependent chain bounds small-N rollouts (9 x 4096: 1.68 -> 1.58 us/step;
// levels 1-3 alike; store-bound shapes unaffected)
constexpr int FG_PRODUCER_PRIO = 2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The arithmetic type of the fused step kernel (fg_step_kernel.hpp, fg_pair_loops.hpp, the reductions below).
// The product library is fp32 (north_star).  -DFG_F64=1 builds the SAME kernel source in fp64 with run-time N only
// (formation_hip_f64.hip -> libformation_hip_f64.so, tests only): the "parity mode" of SURVEY 7.3 H1, in which the
// kernel's algorithm free-runs against the reference's fp64 trajectories over whole fixtures.
#ifndef FG_F64
#define FG_F64 0
#endif
#if FG_F64
typedef double real;
typedef double2 real2;
#define make_real2 make_double2
#else
typedef float real;
typedef float2 real2;
#define make_real2 make_float2
#endif
typedef real realx2 __attribute__((ext_vector_type(2)));
typedef real realx4 __attribute__((ext_vector_type(4)));

// fp32: hardware transcendentals (v_sqrt / v_exp / v_log / v_rcp, ~1 ulp); fp64: the math library
FG_DEV float rmin(float a, float b) { return fminf(a, b); }
FG_DEV float rmax(float a, float b) { return fmaxf(a, b); }
FG_DEV float rabs(float a) { return fabsf(a); }
FG_DEV float rsqrt_(float a) { return sqrtf(a); }
FG_DEV float rfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
FG_DEV float hw_sqrt(float a) { return __builtin_amdgcn_sqrtf(a); }
FG_DEV float hw_rcp(float a) { return __builtin_amdgcn_rcpf(a); }
FG_DEV float hw_log(float a) { return __logf(a); }
FG_DEV float hw_exp(float a) { return __expf(a); }
FG_DEV double rmin(double a, double b) { return fmin(a, b); }
FG_DEV double rmax(double a, double b) { return fmax(a, b); }
FG_DEV double rabs(double a) { return fabs(a); }
FG_DEV double rsqrt_(double a) { return sqrt(a); }
FG_DEV double rfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
FG_DEV double hw_sqrt(double a) { return sqrt(a); }
FG_DEV double hw_rcp(double a) { return 1.0 / a; }
FG_DEV double hw_log(double a) { return log(a); }
FG_DEV double hw_exp(double a) { return exp(a); }

// World / scenario constants as the kernels read them: the C ABI's FgParams (fp32 fields), or the same names in fp64
// for the parity build (0.1f is not the reference's 0.1)
#if FG_F64
struct KParams {
    double dt, damping, contact_force, contact_margin, sensitivity, mass, dist_min, collide_thresh;
    int32_t world_length, auto_reset;
    uint64_t seed, rng_offset;
    double accel, max_speed, u_noise;
    int32_t num_walls;
    FgWall walls[FG_MAX_WALLS];
    int32_t obs_env_pitch, env_index_base;
    const uint64_t* rng_offset_dev;
    const double* agent_props;
    const double* comm_state;
    int32_t obs_placed, reserved0;
};
#else
typedef FgParams KParams;
#endif
// per-launch offset of the counter RNG: by-value part + the caller's optional device counter (FgParams.rng_offset_dev)
FG_DEV uint64_t rng_base(const KParams& p) { return p.rng_offset + (p.rng_offset_dev ? *p.rng_offset_dev : 0ull); }

// LDS block of one environment, in floats:
//   float2 tables  A[3N] = post pos[N] | zeros[N-1] | ideal_shape[N] | ideal_vel[1],  V[N],  NV[N] = -V
//                  (what the observation writers read: unit u >= N of any row is A[u])
//   float arrays   QX QY (pre-step pos)  PX PY (post-step pos)  SX SY (ideal shape), each padded
//                  to NP = N rounded up to 4 with a far-away sentinel, read two agents at a time
//                  (ds_read_b64) by the packed-math pair loops
__host__ __device__ constexpr int npad(int n) { return (n + 3) & ~3; }
__host__ __device__ constexpr int env_block_floats(int n) { return 10 * n + 6 * npad(n); }
FG_DEV real2* env_tables(real2* smem, int ee, int n) {
    return reinterpret_cast<real2*>(reinterpret_cast<real*>(smem) + ee * env_block_floats(n));
}
FG_DEV const real2* env_tables(const real2* smem, int ee, int n) {
    return reinterpret_cast<const real2*>(reinterpret_cast<const real*>(smem) + ee * env_block_floats(n));
}
constexpr real FAR_AWAY = 1.0e18f;   // sentinel coordinate: squared distances stay finite (2e36)

// Constants of the demo controller's hierarchy (fg_policy_kernels.hpp), rounded ONCE on the host so that kernels
// with a compile-time and with a run-time agent count multiply by the very same values (-fapprox-func turns a
// device-side division into v_rcp_f32, a compile-time one is folded exactly).
constexpr int FG_POLICY_MAX_LEVELS = 10;       // 2^10 = FG_MAX_AGENTS
struct FgPolicyLevels {
    int per, L;                                // agents per group of a level, number of levels: N = per^L
    float inv_per;                             // (float)(1.0 / per)
    float inv_sub[FG_POLICY_MAX_LEVELS];       // inv_sub[l] = (float)(1.0 / per^l): sub-group sums -> centroids
};

struct Args {
    KParams p;
    int B, N, K, obs_every;
    real inv_n;               // 1 / N, correctly rounded on the host: run-time-N kernels must use the very value that
                               // compile-time-N kernels constant-fold (-fapprox-func turns a device-side division into v_rcp_f32)
    int do_phys, do_post;
    int groups;                // wide pipelined kernel, K == 1: env batches per workgroup
    int obs_only;              // step_kernel: only the observation stream (second launch of the split step)
    int split;                 // step_kernel, one env per workgroup: workgroups per env in that launch (0 / 1 = one)
    real* px; real* py; real* vx; real* vy;
    const real* act;          // [K][B][N][2]
    real* shape;              // [B][N][2]
    real* ivel;               // [B][2]
    int32_t* step;             // [B]
    real* obs;                // [slots][B] env blocks of [N][6N], obs_pitch float2 units apart
    long long obs_pitch;       // float2 units between consecutive env blocks: 3 N^2 (contiguous) or a padded pitch
    real* rew;                // [K][B][N]
    real* indiv;              // [K][B][N] or NULL
    uint8_t* done;             // [K][B][N] or NULL
    int32_t* near_lm; int32_t* near_ag; int32_t* hd_idx;
    // closed-loop rollouts (fg_rollout_hd_policy): the actions come from the demo controller (fg_policy_kernels.hpp)
    // evaluated on the workgroup's own state instead of from `act`
    FgPolicyLevels pl;
    real* act_out;            // [K][B][N][2] the actions taken
    real coll_scale;          // heterogeneous agents: collision-penalty distance of a pair = coll_scale * (size_a + size_b)
                               // (= collide_thresh / dist_min of the uniform description, divided on the host)
};

// ---------------------------------------------------------------------------
// reductions over the lanes of one environment
// ---------------------------------------------------------------------------
enum { R_SUM = 0, R_MAX = 1, R_MIN = 2 };

template <int OP> FG_DEV real combine(real a, real b) {
    if (OP == R_SUM) return a + b;
    if (OP == R_MAX) return rmax(a, b);
    return rmin(a, b);
}

// Cross-lane partner fetch for a butterfly reduction step, without going through the LDS
// crossbar (ds_bpermute costs ~100 cycles of latency per step): DPP quad permutes and row
// mirrors inside a 16-lane row, v_permlane16/32_swap across rows (gfx950).  STEP 4 and 8 use
// mirrors instead of xor: any pairing of disjoint halves that already hold their own totals
// gives the same reduction.
template <int STEP, int OP> FG_DEV real bfly(real v) {
#if FG_F64
    return combine<OP>(v, __shfl_xor(v, STEP));        // 64-bit values: through ds_bpermute (parity build, not tuned)
#else
    const int iv = __builtin_bit_cast(int, v);
    if constexpr (STEP >= 16) {
        // v_permlane{16,32}_swap(v, v) returns the two row / half sets side by side:
        // {rows 0,0,2,2 | rows 1,1,3,3} resp. {low,low | high,high}; combining them IS the step
        const auto sw = (STEP == 16) ? __builtin_amdgcn_permlane16_swap(iv, iv, false, false)
                                     : __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
        return combine<OP>(__builtin_bit_cast(real, (int)sw[0]), __builtin_bit_cast(real, (int)sw[1]));
    } else {
        constexpr int CTRL = (STEP == 1) ? 0xB1      // quad_perm [1,0,3,2]
                           : (STEP == 2) ? 0x4E      // quad_perm [2,3,0,1]
                           : (STEP == 4) ? 0x141     // row_half_mirror
                                         : 0x140;    // row_mirror
        const int r = __builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xF, 0xF, false);
        return combine<OP>(v, __builtin_bit_cast(real, r));
    }
#endif
}

// G <= 64: the env occupies an aligned group of G lanes of one wave -> in-register butterfly.
// G  > 64: the env is the whole workgroup (E == 1) -> wave butterfly + LDS partials.
template <int G, int T, int NV, int OP0, int OP1, int OP2, int OP3>
FG_DEV void env_reduce(real (&v)[NV], real* scratch) {
    constexpr int W = (G <= 64) ? G : 64;
#define FG_STEP(S)                                                                   \
    if constexpr (W > S) {                                                           \
        if constexpr (NV > 0) v[0] = bfly<S, OP0>(v[0]);           \
        if constexpr (NV > 1) v[1] = bfly<S, OP1>(v[1]);           \
        if constexpr (NV > 2) v[2] = bfly<S, OP2>(v[2]);           \
        if constexpr (NV > 3) v[3] = bfly<S, OP3>(v[3]);           \
    }
    FG_STEP(1) FG_STEP(2) FG_STEP(4) FG_STEP(8) FG_STEP(16) FG_STEP(32)
#undef FG_STEP
    if (G > 64) {
        constexpr int NW = T / 64;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < NV; ++q) scratch[wave * 4 + q] = v[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = scratch[q];
        for (int w = 1; w < NW; ++w) {
            if constexpr (NV > 0) v[0] = combine<OP0>(v[0], scratch[w * 4 + 0]);
            if constexpr (NV > 1) v[1] = combine<OP1>(v[1], scratch[w * 4 + 1]);
            if constexpr (NV > 2) v[2] = combine<OP2>(v[2], scratch[w * 4 + 2]);
            if constexpr (NV > 3) v[3] = combine<OP3>(v[3], scratch[w * 4 + 3]);
        }
        __syncthreads();   // scratch is reused by the next reduction
    }
}

// ---------------------------------------------------------------------------
// counter-based RNG for the device-side reset (Philox4x32-10, Salmon et al. 2011)
// ---------------------------------------------------------------------------
FG_DEV void philox4x32(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    // The keys are wave-uniform launch constants, so the compiler would hoist the ten round keys (20 SGPRs) out of
    // whatever loop the call sits in - here the rarely taken episode-reset branch of the step loop - and pay for
    // them with SGPR spills on the hot path.  Opaque keys keep the schedule inside the branch.
    asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t m0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t m1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(m1 >> 32) ^ c[1] ^ k0;
        const uint32_t n2 = (uint32_t)(m0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)m1; c[3] = (uint32_t)m0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
FG_DEV float u_pm1(uint32_t x) {           // uniform in [-1, 1)
    return (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f;
}

// ---------------------------------------------------------------------------
// World options no reference scenario enables: walls, motor noise, speed clamp
// ---------------------------------------------------------------------------
// core.py:325-362 get_wall_collision_force, summed over the walls; a ghost entity passes through soft walls (:326-327)
FG_DEV void wall_forces(const KParams& P, real2 p, real size, real& fx, real& fy, bool ghost = false) {
#pragma unroll 1
    for (int w = 0; w < FG_MAX_WALLS; ++w) {                                   // static indices: no scratch copy
        if (w >= P.num_walls) break;
        const FgWall wl = P.walls[w];
        if (ghost && wl.soft) continue;
        const real prll = wl.vertical ? p.y : p.x;
        const real perp = wl.vertical ? p.x : p.y;
        if (prll < wl.end0 - size || prll > wl.end1 + size) continue;      // beyond the endpoints
        real ct = 1.0f, st = 0.0f;
        if (prll < wl.end0 || prll > wl.end1) {                            // rounding the corner
            const real past = (prll < wl.end0) ? prll - wl.end0 : prll - wl.end1;
            st = past / size;                                              // sin(theta)
            ct = hw_sqrt(rmax(1.0f - st * st, real(0)));
        }
        const real dmin = ct * size + 0.5f * wl.width;
        const real delta = perp - wl.axis_pos;
        const real dist = rabs(delta);
        const real x = (dmin - dist) / P.contact_margin;
        const real pen = P.contact_margin * (rmax(x, real(0)) + hw_log(1.0f + hw_exp(-rabs(x))));
        const real mag = P.contact_force * delta * hw_rcp(dist) * pen;   // dist == 0 -> NaN, as the reference
        const real f_perp = ct * mag, f_prll = st * rabs(mag);
        if (wl.vertical) { fx += f_perp; fy += f_prll; } else { fy += f_perp; fx += f_prll; }
    }
}

FG_DEV real2 motor_noise(uint64_t seed, uint32_t b, uint32_t i, uint64_t offset) {
    uint32_t c[4] = {b, i ^ 0x80000000u, (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float r = sqrtf(-2.0f * __logf(((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f)));   // Box-Muller (fp32 draw)
    const float a = 6.2831853f * ((float)(c[1] >> 8) * (1.0f / 16777216.0f));
    return make_real2(r * __cosf(a), r * __sinf(a));
}

// One agent's own properties (core.py:45-109): the World-wide scalars of FgParams, or its row of FgParams.agent_props
struct AgentProps {
    real mass, size, accel, max_speed, u_noise, sens;
    int flags;                 // FG_AGENT_IMMOVABLE | FG_AGENT_NO_COLLIDE | FG_AGENT_GHOST (core.py:54-58) | FG_AGENT_SCRIPTED; 0 = an ordinary agent
};
FG_DEV AgentProps agent_props_of(const KParams& P, int i, bool in_range) {
    AgentProps q = {P.mass, real(0.5f) * P.dist_min, P.accel, P.max_speed, P.u_noise, P.sensitivity, 0};
    if (P.agent_props && in_range) {
        const auto* r = P.agent_props + (size_t)i * FG_AGENT_PROPS;
        q.mass = r[0]; q.size = r[1]; q.accel = r[2]; q.max_speed = r[3]; q.u_noise = r[4];
        q.flags = (int)r[6];
        q.sens = (q.accel > 0.0f) ? q.accel : P.sensitivity;               // environment.py:218-220
        if (q.flags & FG_AGENT_SCRIPTED) q.sens = 1.0f;                    // core.py:210-211: a scripted agent's action.u is used as it is
    }
    return q;
}

// action force incl. accel (core.py:236, environment.py:219-220) and motor noise (core.py:232-233)
FG_DEV real2 action_force(const KParams& P, const AgentProps& q, real2 u, uint32_t b, uint32_t i, uint64_t offset) {
    const real gain = (q.accel > 0.0f) ? q.mass * q.accel : q.mass;
    real2 f = make_real2(gain * (q.sens * u.x), gain * (q.sens * u.y));
    if (q.u_noise > 0.0f) {
        const real2 n = motor_noise(P.seed, b, i, offset);
        f.x += q.u_noise * n.x;
        f.y += q.u_noise * n.y;
    }
    return f;
}

FG_DEV real2 clamp_speed(real max_speed, real2 v) {                      // core.py:271-276
    if (max_speed > 0.0f) {
        const real speed = rsqrt_(v.x * v.x + v.y * v.y);
        if (speed > max_speed) { v.x = v.x / speed * max_speed; v.y = v.y / speed * max_speed; }
    }
    return v;
}

}  // namespace fg

#endif  // FG_COMMON_HPP_
