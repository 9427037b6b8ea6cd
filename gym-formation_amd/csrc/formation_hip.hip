// formation_hip.hip - hand-written HIP kernels (gfx950 / CDNA4, wave64) for the
// formation_gym hot path, and the C ABI declared in include/formation_hip.h.
//
// What one launch of `step_kernel` does for a group of E environments held by
// one workgroup (reference lines in /root/reference/formation_gym/):
//   phase 1  load SoA state + actions + ideal shape of the group into LDS
//   phase 2  World.step: action force (core.py:228-237, environment.py:216-221),
//            all-pairs soft-contact force on PRE-step positions (core.py:240-262,
//            :289-322), damped Euler integration (core.py:264-277)
//   phase 3  Scenario.reward (formation_hd_env.py:61-75): centroid and mean
//            velocity by wave shuffles, one pass over post-step pairs giving the
//            Hausdorff row/column minima and the collision counts, env-wide
//            max/sum by shuffles (one env never spans a wave unless N > 64)
//   phase 4  optional vec-env auto-reset (env_wrappers.py:14-18)
//   phase 5  Scenario.observation (formation_hd_env.py:52-59) for all N agents:
//            the group's [E][N][6N] block is one contiguous span of global
//            memory; every lane composes two (x,y) units from LDS and issues one
//            16-byte store, lanes consecutive -> 1 KiB per wave-instruction.
// Observation bytes (24 N^2 per env) dominate traffic; everything else is 53 N + 16.
// There is no dense contraction here, hence no MFMA: the kernel is HBM-store bound.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "formation_hip.h"

#ifndef FG_PROBES
#define FG_PROBES 0        // 1: honour the FG_PROBE timing experiments (results are then NOT valid)
#endif
#ifndef FG_WRITER_PRIO
#define FG_WRITER_PRIO 0   // tuning: s_setprio level of the rollout writer waves
#endif
#ifndef FG_TILE_NT
#define FG_TILE_NT 0     // tuning: non-temporal stores in the LDS-tiled writer (measured: -6 %, profiles/README.md)
#endif

namespace fg {

#define FG_DEV __device__ __forceinline__
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS block of one environment, in floats:
//   float2 tables  A[3N] = post pos[N] | zeros[N-1] | ideal_shape[N] | ideal_vel[1],  V[N],  NV[N] = -V
//                  (what the observation writers read: unit u >= N of any row is A[u])
//   float arrays   QX QY (pre-step pos)  PX PY (post-step pos)  SX SY (ideal shape), each padded
//                  to NP = N rounded up to 4 with a far-away sentinel, read two agents at a time
//                  (ds_read_b64) by the packed-math pair loops
__host__ __device__ constexpr int npad(int n) { return (n + 3) & ~3; }
__host__ __device__ constexpr int env_block_floats(int n) { return 10 * n + 6 * npad(n); }
FG_DEV float2* env_tables(float2* smem, int ee, int n) {
    return reinterpret_cast<float2*>(reinterpret_cast<float*>(smem) + ee * env_block_floats(n));
}
FG_DEV const float2* env_tables(const float2* smem, int ee, int n) {
    return reinterpret_cast<const float2*>(reinterpret_cast<const float*>(smem) + ee * env_block_floats(n));
}
constexpr float FAR_AWAY = 1.0e18f;   // sentinel coordinate: squared distances stay finite (2e36)

struct Args {
    FgParams p;
    int B, N, K, obs_every;
    int do_phys, do_post;
    int groups;                // wide pipelined kernel, K == 1: env batches per workgroup
    int probe;                 // timing probes, only honoured in -DFG_PROBES=1 builds (profiles/README.md)
    float* px; float* py; float* vx; float* vy;
    const float* act;          // [K][B][N][2]
    float* shape;              // [B][N][2]
    float* ivel;               // [B][2]
    int32_t* step;             // [B]
    float* obs;                // [slots][B][N][6N]
    float* rew;                // [K][B][N]
    float* indiv;              // [K][B][N] or NULL
    uint8_t* done;             // [K][B][N] or NULL
    int32_t* near_lm; int32_t* near_ag; int32_t* hd_idx;
};

// ---------------------------------------------------------------------------
// reductions over the lanes of one environment
// ---------------------------------------------------------------------------
enum { R_SUM = 0, R_MAX = 1, R_MIN = 2 };

template <int OP> FG_DEV float combine(float a, float b) {
    if (OP == R_SUM) return a + b;
    if (OP == R_MAX) return fmaxf(a, b);
    return fminf(a, b);
}

// Cross-lane partner fetch for a butterfly reduction step, without going through the LDS
// crossbar (ds_bpermute costs ~100 cycles of latency per step): DPP quad permutes and row
// mirrors inside a 16-lane row, v_permlane16/32_swap across rows (gfx950).  STEP 4 and 8 use
// mirrors instead of xor: any pairing of disjoint halves that already hold their own totals
// gives the same reduction.
template <int STEP, int OP> FG_DEV float bfly(float v) {
    const int iv = __builtin_bit_cast(int, v);
    if constexpr (STEP >= 16) {
        // v_permlane{16,32}_swap(v, v) returns the two row / half sets side by side:
        // {rows 0,0,2,2 | rows 1,1,3,3} resp. {low,low | high,high}; combining them IS the step
        const auto sw = (STEP == 16) ? __builtin_amdgcn_permlane16_swap(iv, iv, false, false)
                                     : __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
        return combine<OP>(__builtin_bit_cast(float, (int)sw[0]), __builtin_bit_cast(float, (int)sw[1]));
    } else {
        constexpr int CTRL = (STEP == 1) ? 0xB1      // quad_perm [1,0,3,2]
                           : (STEP == 2) ? 0x4E      // quad_perm [2,3,0,1]
                           : (STEP == 4) ? 0x141     // row_half_mirror
                                         : 0x140;    // row_mirror
        const int r = __builtin_amdgcn_update_dpp(iv, iv, CTRL, 0xF, 0xF, false);
        return combine<OP>(v, __builtin_bit_cast(float, r));
    }
}

// G <= 64: the env occupies an aligned group of G lanes of one wave -> in-register butterfly.
// G  > 64: the env is the whole workgroup (E == 1) -> wave butterfly + LDS partials.
template <int G, int T, int NV, int OP0, int OP1, int OP2, int OP3>
FG_DEV void env_reduce(float (&v)[NV], float* scratch) {
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
// core.py:325-362 get_wall_collision_force, summed over the walls (hard walls, no ghosts)
FG_DEV void wall_forces(const FgParams& P, float2 p, float size, float& fx, float& fy) {
#pragma unroll
    for (int w = 0; w < FG_MAX_WALLS; ++w) {                                   // static indices: no scratch copy
        if (w >= P.num_walls) break;
        const FgWall wl = P.walls[w];
        const float prll = wl.vertical ? p.y : p.x;
        const float perp = wl.vertical ? p.x : p.y;
        if (prll < wl.end0 - size || prll > wl.end1 + size) continue;      // beyond the endpoints
        float ct = 1.0f, st = 0.0f;
        if (prll < wl.end0 || prll > wl.end1) {                            // rounding the corner
            const float past = (prll < wl.end0) ? prll - wl.end0 : prll - wl.end1;
            st = past / size;                                              // sin(theta)
            ct = __builtin_amdgcn_sqrtf(fmaxf(1.0f - st * st, 0.0f));
        }
        const float dmin = ct * size + 0.5f * wl.width;
        const float delta = perp - wl.axis_pos;
        const float dist = fabsf(delta);
        const float x = (dmin - dist) / P.contact_margin;
        const float pen = P.contact_margin * (fmaxf(x, 0.0f) + __logf(1.0f + __expf(-fabsf(x))));
        const float mag = P.contact_force * delta * __builtin_amdgcn_rcpf(dist) * pen;   // dist == 0 -> NaN, as the reference
        const float f_perp = ct * mag, f_prll = st * fabsf(mag);
        if (wl.vertical) { fx += f_perp; fy += f_prll; } else { fy += f_perp; fx += f_prll; }
    }
}

FG_DEV float2 motor_noise(uint64_t seed, uint32_t b, uint32_t i, uint64_t offset) {
    uint32_t c[4] = {b, i ^ 0x80000000u, (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float r = sqrtf(-2.0f * __logf(((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f)));   // Box-Muller
    const float a = 6.2831853f * ((float)(c[1] >> 8) * (1.0f / 16777216.0f));
    return make_float2(r * __cosf(a), r * __sinf(a));
}

// action force incl. accel (core.py:236, environment.py:219-220) and motor noise (core.py:232-233)
FG_DEV float2 action_force(const FgParams& P, float2 u, uint32_t b, uint32_t i, uint64_t offset) {
    const float gain = (P.accel > 0.0f) ? P.mass * P.accel : P.mass;
    float2 f = make_float2(gain * (P.sensitivity * u.x), gain * (P.sensitivity * u.y));
    if (P.u_noise > 0.0f) {
        const float2 n = motor_noise(P.seed, b, i, offset);
        f.x += P.u_noise * n.x;
        f.y += P.u_noise * n.y;
    }
    return f;
}

FG_DEV float2 clamp_speed(const FgParams& P, float2 v) {                    // core.py:271-276
    if (P.max_speed > 0.0f) {
        const float speed = sqrtf(v.x * v.x + v.y * v.y);
        if (speed > P.max_speed) { v.x = v.x / speed * P.max_speed; v.y = v.y / speed * P.max_speed; }
    }
    return v;
}

// ---------------------------------------------------------------------------
// pair loops, two partner agents per iteration with packed fp32 math (v_pk_*_f32)
// ---------------------------------------------------------------------------
// World.step contact force on agent i (core.py:289-322): sum over j != i on PRE-step positions.
FG_DEV float2 contact_force_packed(const float* __restrict__ QX, const float* __restrict__ QY, int NP,
                                   int i, float2 p, float cf, float kmargin, float dmin, float cutoff2) {
    float fx = 0.0f, fy = 0.0f;
    const f32x2 px = {p.x, p.x}, py = {p.y, p.y};
    const float inv_k = 1.0f / kmargin;
    auto add = [&](float dx, float dy, float d2) {
        // Hardware transcendentals (v_sqrt/v_exp/v_log/v_rcp, ~1 ulp): the contact branch is
        // taken by about half of all wave iterations at uniform-random density, so its length
        // sets the physics time.  Relative force error ~3e-7 (|f| <= 6) -> < 2e-8 on positions.
        // d2 == 0 for two distinct agents is kept: 0 * inf -> NaN as in core.py:312.
        const float d = __builtin_amdgcn_sqrtf(d2);
        const float x = (dmin - d) * inv_k;
        const float pen = kmargin * (fmaxf(x, 0.0f) + __logf(1.0f + __expf(-fabsf(x))));
        const float c = cf * pen * __builtin_amdgcn_rcpf(d);
        fx += dx * c;
        fy += dy * c;
    };
#pragma unroll 2
    for (int j = 0; j < NP; j += 2) {
        const f32x2 qx = *reinterpret_cast<const f32x2*>(QX + j);
        const f32x2 qy = *reinterpret_cast<const f32x2*>(QY + j);
        const f32x2 dx = px - qx, dy = py - qy;
        const f32x2 d2 = dx * dx + dy * dy;
        // beyond the cutoff the softplus penetration is below fp32 resolution of the force: skipped
        const bool n0 = (d2.x < cutoff2) && (j != i);
        const bool n1 = (d2.y < cutoff2) && (j + 1 != i);
        if (n0 || n1) {
            if (n0) add(dx.x, dy.x, d2.x);
            if (n1) add(dx.y, dy.y, d2.y);
        }
    }
    return make_float2(fx, fy);
}

// Scenario.reward inner pass for agent i / ideal point i (formation_hd_env.py:61-75):
//   rowmin = min_j |p~_i - s_j|^2,  colmin = min_j |p~_j - s_i|^2,  cnt = #{j != i : |p_j - p_i| < thr}
template <bool IDX>
FG_DEV void reward_pass_packed(const float* __restrict__ PX, const float* __restrict__ PY,
                               const float* __restrict__ SX, const float* __restrict__ SY, int NP,
                               float2 p, float ptx, float pty, float tx, float ty, float thr2,
                               float& rowmin, float& colmin, int& cnt, int& arg_lm, int& arg_ag) {
    const f32x2 px = {p.x, p.x}, py = {p.y, p.y};
    const f32x2 ptx2 = {ptx, ptx}, pty2 = {pty, pty}, tx2 = {tx, tx}, ty2 = {ty, ty};
    int c = -1;                                    // the self pair (distance 0) is counted below
#pragma unroll 2
    for (int j = 0; j < NP; j += 2) {
        const f32x2 qx = *reinterpret_cast<const f32x2*>(PX + j);
        const f32x2 qy = *reinterpret_cast<const f32x2*>(PY + j);
        const f32x2 sx = *reinterpret_cast<const f32x2*>(SX + j);
        const f32x2 sy = *reinterpret_cast<const f32x2*>(SY + j);
        const f32x2 cx = qx - px, cy = qy - py;
        const f32x2 dc = cx * cx + cy * cy;
        c += (dc.x < thr2 ? 1 : 0) + (dc.y < thr2 ? 1 : 0);
        const f32x2 rx = ptx2 - sx, ry = pty2 - sy;
        const f32x2 dr = rx * rx + ry * ry;
        const f32x2 ux = qx - tx2, uy = qy - ty2;
        const f32x2 dq = ux * ux + uy * uy;
        if (IDX) {
            if (dr.x < rowmin) { rowmin = dr.x; arg_lm = j; }
            if (dr.y < rowmin) { rowmin = dr.y; arg_lm = j + 1; }
            if (dq.x < colmin) { colmin = dq.x; arg_ag = j; }
            if (dq.y < colmin) { colmin = dq.y; arg_ag = j + 1; }
        } else {
            rowmin = fminf(fminf(rowmin, dr.x), dr.y);
            colmin = fminf(fminf(colmin, dq.x), dq.y);
        }
    }
    cnt = c + (thr2 > 0.0f ? 0 : 1);               // thr == 0: not even the self pair was counted
}

// ---------------------------------------------------------------------------
// observation row writer (specialised N): every wave streams whole rows.
// A row is [v_i | p_j - p_i (j != i) | zeros | ideal_shape | ideal_vel] = 3N (x,y) units.
//  * units N..3N-1 are identical for every row of an env: each lane loads its share ONCE
//    into registers and then only stores (one 8-byte store per 64 units per row);
//  * units 0..N-1: lane u keeps p_{u-1} and p_u in registers; per row it reads p_row (or
//    -v_row on lane u = 0) from LDS, selects by (u-1 >= row), subtracts, stores.
// ~3 vector instructions per 512-byte wave store instead of ~25 for a flat decode.
// Waves of the workgroup split the E*N rows: whole envs per wave when E >= #waves, else
// rows of one env round-robin over the waves that share it.
// ---------------------------------------------------------------------------
// value select (never a pointer select: that would go through scratch + flat loads)
FG_DEV float2 lds_if(bool c, const float2* __restrict__ p, int idx_if_true) {
    const float2 t = p[c ? idx_if_true : 0];
    return make_float2(c ? t.x : 0.0f, c ? t.y : 0.0f);
}

template <int NC, int NW, int E>
FG_DEV void write_obs_rows(const float2* __restrict__ tables0, int env_stride, int w,
                           float2* __restrict__ out_env0, int El, int parts) {
    // tables0 / env_stride: A-table of env 0 and the distance (float2) to the next env's;
    // w: index of this wave among the NW waves that share the job
    // parts: bit 0 = relative-position units, bit 1 = static units (zeros | shape | ideal_vel)
    constexpr int N = NC;
    constexpr int WPE = (E >= NW) ? 1 : NW / E;             // waves sharing one env
    constexpr int ESTEP = (E >= NW) ? NW : 1;               // env stride of one wave
    static_assert((E >= NW) ? (E % NW == 0) : (NW % E == 0), "waves and envs must tile");
    const int lane = threadIdx.x & 63;
    const int row0 = (E >= NW) ? 0 : w % WPE;
    constexpr unsigned ROWU = 3u * N;                       // units per row
    for (int ee = (E >= NW) ? w : w / WPE; ee < El; ee += (E >= NW ? ESTEP : E)) {
        const float2* __restrict__ AA = tables0 + (size_t)ee * env_stride;
        float2* __restrict__ out = out_env0 + (size_t)ee * (ROWU * N);
        if constexpr (N <= 64) {
            // Blocks of RW = 64/N rows: one wave store covers the relative-position part of the
            // whole block, then the static part (zeros | ideal_shape | ideal_vel, the same for every
            // row: register-resident) of the same rows follows at once, so that the cache lines a
            // row shares with its neighbours are completed back to back.
            constexpr int RW = 64 / N;
            const int rsub = lane / N, u = lane - rsub * N;
            const bool act = rsub < RW;
            const float2 zero = make_float2(0.f, 0.f);
            const float2 Pm = lds_if(act && u >= 1, AA, u - 1);
            const float2 Pu = lds_if(act && u >= 1, AA, u);
            const int xoff = (u == 0) ? 4 * N : 0;          // lane u = 0 reads -v_row (NV = A + 4N)
            constexpr int CS = (2 * N + 63) / 64;           // 64-unit chunks of the static part
            constexpr int RS = (2 * N <= 64) ? 64 / (2 * N) : 1;   // rows per static store
            const int ssub = (2 * N <= 64) ? lane / (2 * N) : 0;
            const int sidx = (2 * N <= 64) ? lane - ssub * 2 * N : lane;
            float2 sv[CS];
#pragma unroll
            for (int c = 0; c < CS; ++c)
                sv[c] = lds_if(ssub < RS && sidx + 64 * c < 2 * N, AA, N + sidx + 64 * c);
#pragma unroll 2
            for (int rb = row0; rb < N; rb += RW * WPE) {
                const int r = rb + rsub * WPE;
                if ((parts & 1) && act && r < N) {
                    const float2 x = AA[xoff + r];
                    const float2 c = (u - 1 >= r) ? Pu : Pm;
                    out[(unsigned)r * ROWU + (unsigned)u] = make_float2(c.x - x.x, c.y - x.y);
                }
#pragma unroll
                for (int k0 = 0; k0 < RW; k0 += RS) {
                    const int rs = rb + (k0 + ssub) * WPE;
                    if ((parts & 2) && ssub < RS && k0 + ssub < RW && rs < N) {
#pragma unroll
                        for (int c = 0; c < CS; ++c)
                            if (sidx + 64 * c < 2 * N) out[(unsigned)rs * ROWU + (unsigned)(N + sidx + 64 * c)] = sv[c];
                    }
                }
            }
        } else {
            // ---- N > 64: one row per iteration, register-cached chunks of 64 units ----
            constexpr int CD = (N + 63) / 64, CS = (2 * N + 63) / 64;
            const float2 zero = make_float2(0.f, 0.f);
            float2 Pm[CD], Pu[CD], sv[CS];
#pragma unroll
            for (int c = 0; c < CD; ++c) {
                const int u = lane + 64 * c;
                Pm[c] = lds_if(u >= 1 && u < N, AA, u - 1);
                Pu[c] = lds_if(u >= 1 && u < N, AA, u);
            }
#pragma unroll
            for (int c = 0; c < CS; ++c) sv[c] = lds_if(lane + 64 * c < 2 * N, AA, N + lane + 64 * c);
#pragma unroll 2
            for (int r = row0; r < N; r += WPE) {
                const float2 xp = AA[r];                    // p_row, wave-uniform broadcast
                const float2 x0 = AA[(lane == 0 ? 4 * N : 0) + r];
                float2* __restrict__ orow = out + (unsigned)r * ROWU;
#pragma unroll
                for (int c = 0; c < CD; ++c) {
                    const int u = lane + 64 * c;
                    const float2 x = (c == 0) ? x0 : xp;
                    const float2 cc = (u - 1 >= r) ? Pu[c] : Pm[c];
                    if ((parts & 1) && u < N) orow[u] = make_float2(cc.x - x.x, cc.y - x.y);
                }
#pragma unroll
                for (int c = 0; c < CS; ++c)
                    if ((parts & 2) && lane + 64 * c < 2 * N) orow[N + lane + 64 * c] = sv[c];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// LDS-tiled observation writer (N <= 32): a wave composes RT consecutive rows of one env in
// its own LDS tile with the register-cached scheme of write_obs_rows (ds_write_b64), then
// streams the tile out as ONE contiguous span: ds_read_b128 + global_store_dwordx4, lanes
// consecutive, 1 KiB per wave instruction, so almost every 128-byte line is written by a
// single store request.  The tile sits in LDS at the same 16-byte phase as its destination
// (tiles of an odd N start 8 bytes off every other time), so both sides of the copy are
// naturally aligned.  Only the issuing wave touches its tile: LDS operations of one wave
// complete in order, no barrier is needed.
// ---------------------------------------------------------------------------
template <int NC, int RT> constexpr int tile_units() { return (3 * NC * RT + 2 + 1) & ~1; }

// `tiles` holds TWO tiles per writing wave: tile t+1 is composed while tile t drains.
template <int NC, int NW, int E, int RT>
FG_DEV void write_obs_tiled(const float2* __restrict__ tables0, int env_stride, int w, float2* __restrict__ tiles,
                            float2* __restrict__ out_env0, size_t unit0, int El) {
    constexpr int N = NC;
    constexpr int WPE = (E >= NW) ? 1 : NW / E;
    static_assert((E >= NW) ? (E % NW == 0) : (NW % E == 0), "waves and envs must tile");
    static_assert(N <= 32 && N % RT == 0, "tiled writer: N <= 32, RT divides N");
    constexpr unsigned ROWU = 3u * N, NENV = ROWU * N, TU = ROWU * RT;
    constexpr int TILES_ENV = N / RT;                                  // tiles per env
    constexpr int MY_TILES = (TILES_ENV + WPE - 1) / WPE;              // of which this wave takes every WPE-th
    const int lane = threadIdx.x & 63;
    float2* tile0 = tiles + w * 2 * tile_units<NC, RT>();
    constexpr int RW = 64 / N;
    const int rsub = lane / N, u = lane - rsub * N;
    const bool act = rsub < RW;
    const int xoff = (u == 0) ? 4 * N : 0;
    constexpr int RS = 64 / (2 * N);
    const int ssub = lane / (2 * N), sidx = lane - ssub * 2 * N;
    const int e_first = (E >= NW) ? w : w / WPE, e_step = (E >= NW) ? NW : E;
    const int t_first = (E >= NW) ? 0 : w % WPE;
    const int n_env = (El > e_first) ? (El - e_first + e_step - 1) / e_step : 0;
    const int total = n_env * MY_TILES;

    int cur_env = -1;
    const float2* __restrict__ AA = tables0;
    float2 Pm = make_float2(0.f, 0.f), Pu = Pm, sv = Pm;
    auto locate = [&](int t, int& ee, int& r0) {                       // t-th tile of this wave
        const int ie = t / MY_TILES, it = t - ie * MY_TILES;
        ee = e_first + ie * e_step;
        r0 = (t_first + it * WPE) * RT;
    };
    auto compose = [&](int t) {
        int ee, r0; locate(t, ee, r0);
        if (r0 >= N) return;
        if (ee != cur_env) {                                           // per-env register cache
            cur_env = ee;
            AA = tables0 + (size_t)ee * env_stride;
            Pm = lds_if(act && u >= 1, AA, u - 1);
            Pu = lds_if(act && u >= 1, AA, u);
            sv = lds_if(ssub < RS, AA, N + sidx);
        }
        const unsigned par = (unsigned)((unit0 + (size_t)ee * NENV + (size_t)r0 * ROWU) & 1);
        float2* img = tile0 + (t & 1) * tile_units<NC, RT>() + par;     // no restrict: the two tiles alternate
#pragma unroll
        for (int rb = 0; rb < (FG_TILE_NT == 2 ? 0 : RT); rb += RW) {
            const int rl = rb + rsub;
            if (act && rl < RT) {
                const int r = r0 + rl;
                const float2 x = AA[xoff + r];
                const float2 c = (u - 1 >= r) ? Pu : Pm;
                img[(unsigned)rl * ROWU + (unsigned)u] = make_float2(c.x - x.x, c.y - x.y);
            }
        }
#pragma unroll
        for (int rb = 0; rb < (FG_TILE_NT == 2 ? 0 : RT); rb += RS) {
            const int rl = rb + ssub;
            if (ssub < RS && rl < RT) img[(unsigned)rl * ROWU + (unsigned)(N + sidx)] = sv;
        }
    };
    auto stream = [&](int t) {
        int ee, r0; locate(t, ee, r0);
        if (r0 >= N) return;
        const unsigned par = (unsigned)((unit0 + (size_t)ee * NENV + (size_t)r0 * ROWU) & 1);
        const float2* img = tile0 + (t & 1) * tile_units<NC, RT>() + par;
        float2* __restrict__ out = out_env0 + (size_t)ee * NENV + (size_t)r0 * ROWU;
        if (par && lane == 0) out[0] = img[0];
        constexpr unsigned NPMAX = TU >> 1;
        const unsigned npair = (TU - par) >> 1;
        const f32x4* src4 = reinterpret_cast<const f32x4*>(img + par);
        f32x4* __restrict__ dst4 = reinterpret_cast<f32x4*>(out + par);
#pragma unroll
        for (unsigned q0 = 0; q0 < NPMAX; q0 += 64) {
            const unsigned q = q0 + lane;
            if (q < npair) {
                if (FG_TILE_NT == 1) __builtin_nontemporal_store(src4[q], &dst4[q]);
                else if (FG_TILE_NT == 2) { const f32x4 cst = {1.f, 2.f, 3.f, 4.f}; dst4[q] = cst; }   // timing probe
                else dst4[q] = src4[q];
            }
        }
        if (((TU - par) & 1u) && lane == 63) out[TU - 1] = img[TU - 1];
    };
    if (total > 0) compose(0);
    for (int t = 0; t < total; ++t) {
        if (t + 1 < total) compose(t + 1);
        stream(t);
    }
}

// ---------------------------------------------------------------------------
// the fused step / rollout kernel
//   NC  compile-time agent count (0 = run-time a.N)
//   G   lanes reserved per environment for the per-agent phases (power of two >= N)
//   T   threads per workgroup (>= E * G); ALL T threads stream observations
//   E   environments per workgroup
//   IDX also emit the landmark-index assignments
// LDS per env: see env_block_floats(); observation unit u >= N of any row is A[u], unit 0 of row i is A[3N + i].
// ---------------------------------------------------------------------------
#ifndef FG_WPS
#define FG_WPS 0          // tuning: minimum waves per SIMD requested from the register allocator (0 = none)
#endif
template <int NC, int G, int T, int E, bool IDX, int WR, bool OPTS>
__global__ __launch_bounds__(T, ((FG_WPS) > 0 && !IDX && !OPTS) ? (FG_WPS) : 1) void step_kernel(const Args a) {
    // OPTS: World options no reference scenario enables (accel, max_speed, u_noise, walls);
    // compiled into a separate instantiation so that the common path keeps its registers.
    constexpr bool FLAT = (WR == 1);
    static_assert(E * G <= T && (G <= 64 || E == 1), "bad geometry");
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int N = NC ? NC : a.N;
    const int tid = threadIdx.x;
    const int e = (E == 1) ? 0 : tid / G;        // tid >= E*G: no agent, only streams observations
    const int i = (E == 1) ? tid : tid % G;
    const int b0 = blockIdx.x * E;
    const int b = b0 + e;
    const bool env_ok = (e < E) && (b < a.B);     // this thread's lane group owns a live env
    const bool valid = env_ok && (i < N);
    const int El = min(E, a.B - b0);

    const int NP = npad(N);
    float2* const A = env_tables(smem, e < E ? e : 0, N);
    float2* const V = A + 3 * N;
    float2* const NV = A + 4 * N;             // -velocity, read by the row writer
    float* const QX = reinterpret_cast<float*>(A + 5 * N);
    float* const QY = QX + NP;
    float* const PX = QY + NP;
    float* const PY = PX + NP;
    float* const SX = PY + NP;
    float* const SY = SX + NP;
    float* const scratch = reinterpret_cast<float*>(env_tables(smem, E, N));
    volatile int* const reset_flag = reinterpret_cast<volatile int*>(scratch) + 64;   // 2 ints after the 16x4 reduction partials

    const float one_minus_damp = 1.0f - a.p.damping;
    const float dt = a.p.dt;
    const float cutoff = a.p.dist_min + 18.0f * a.p.contact_margin;   // force beyond: < 1e2 k e^-18 ~ 1.5e-9
    const float cutoff2 = cutoff * cutoff;
    const float thr2 = (float)((double)a.p.collide_thresh * (double)a.p.collide_thresh);
    const float invN = 1.0f / (float)N;

    // ---- phase 1: state -> registers + LDS --------------------------------
    float2 p = make_float2(0.f, 0.f), v = make_float2(0.f, 0.f), s = make_float2(0.f, 0.f);
    int t_step = 0;
    const size_t sidx = (size_t)b * N + i;
    if (valid) {
        p = make_float2(a.px[sidx], a.py[sidx]);
        v = make_float2(a.vx[sidx], a.vy[sidx]);
        A[i] = p; V[i] = v; NV[i] = make_float2(-v.x, -v.y);
        QX[i] = p.x; QY[i] = p.y; PX[i] = p.x; PY[i] = p.y;
        if (a.do_post) {
            s = reinterpret_cast<const float2*>(a.shape)[sidx];
            A[2 * N - 1 + i] = s;
            SX[i] = s.x; SY[i] = s.y;
            if (i < N - 1) A[N + i] = make_float2(0.f, 0.f);
            if (i == 0) A[3 * N - 1] = reinterpret_cast<const float2*>(a.ivel)[b];
        }
    } else if (env_ok && i < NP) {              // sentinel partners of the packed pair loops
        QX[i] = FAR_AWAY; QY[i] = FAR_AWAY; PX[i] = FAR_AWAY; PY[i] = FAR_AWAY; SX[i] = FAR_AWAY; SY[i] = FAR_AWAY;
    }
    if (env_ok && a.step) t_step = a.step[b];
    if (tid < 2) reset_flag[tid] = 0;
    __syncthreads();

    for (int k = 0; k < a.K; ++k) {
        int slot = k;
        bool want_obs = a.do_post && a.obs != nullptr;
        if (a.obs_every > 1) { want_obs = want_obs && ((k + 1) % a.obs_every == 0); slot = k / a.obs_every; }
        // ---- phase 2: World.step ------------------------------------------
        if (a.do_phys) {
            if (valid) {
                const float2 u = reinterpret_cast<const float2*>(a.act)[((size_t)k * a.B + b) * N + i];
                float2 f = contact_force_packed(QX, QY, NP, i, p, a.p.contact_force, a.p.contact_margin,
                                                a.p.dist_min, cutoff2);
                if constexpr (OPTS) {
                    const float2 fa = action_force(a.p, u, (uint32_t)b, (uint32_t)i, a.p.rng_offset + k);
                    f.x += fa.x; f.y += fa.y;
                    if (a.p.num_walls > 0) wall_forces(a.p, p, 0.5f * a.p.dist_min, f.x, f.y);
                } else {
                    f.x += a.p.mass * (a.p.sensitivity * u.x);
                    f.y += a.p.mass * (a.p.sensitivity * u.y);
                }
                v.x = v.x * one_minus_damp + (f.x / a.p.mass) * dt;
                v.y = v.y * one_minus_damp + (f.y / a.p.mass) * dt;
                if constexpr (OPTS) v = clamp_speed(a.p, v);
                p.x += v.x * dt;
                p.y += v.y * dt;
                A[i] = p; V[i] = v; NV[i] = make_float2(-v.x, -v.y);
                PX[i] = p.x; PY[i] = p.y;
            }
            t_step += 1;
            // The barrier that publishes the post-step tables also carries one bit per group:
            // "some env of this workgroup finishes its episode in this step" (auto-reset only),
            // so the common no-reset step pays no extra barrier later.
            if (a.p.auto_reset && env_ok && i == 0 && t_step >= a.p.world_length) reset_flag[k & 1] = 1;
            __syncthreads();
        }

        if (a.do_post) {
            // ---- phase 3: reward -------------------------------------------
            float sums[4] = {valid ? p.x : 0.f, valid ? p.y : 0.f, valid ? v.x : 0.f, valid ? v.y : 0.f};
            env_reduce<G, T, 4, R_SUM, R_SUM, R_SUM, R_SUM>(sums, scratch);
            const float mx = sums[0] * invN, my = sums[1] * invN;
            const float mvx = sums[2] * invN, mvy = sums[3] * invN;
            const float ptx = p.x - mx, pty = p.y - my;        // centred own position
            const float tx = s.x + mx, ty = s.y + my;          // own ideal point, un-centred
            float rowmin = INFINITY, colmin = INFINITY;
            int cnt = 0, arg_lm = 0, arg_ag = 0;
            if (valid)
                reward_pass_packed<IDX>(PX, PY, SX, SY, NP, p, ptx, pty, tx, ty, thr2,
                                        rowmin, colmin, cnt, arg_lm, arg_ag);
            float red[3] = {valid ? rowmin : -INFINITY, valid ? colmin : -INFINITY, (float)cnt};
            env_reduce<G, T, 3, R_MAX, R_MAX, R_SUM, R_SUM>(red, scratch);
            const float H = sqrtf(fmaxf(red[0], red[1]));
            const float2 iv = A[3 * N - 1];
            const float ex = iv.x - mvx, ey = iv.y - mvy;
            const float velterm = sqrtf(ex * ex + ey * ey);
            const float indiv = (-H - velterm) - (float)cnt;
            const float shared = (float)(-(double)N * ((double)H + (double)velterm) - (double)red[2]);
            const bool is_done = t_step >= a.p.world_length;
            if (valid) {
                const size_t o = ((size_t)k * a.B + b) * N + i;
                if (a.rew) a.rew[o] = shared;
                if (a.indiv) a.indiv[o] = indiv;
                if (a.done) a.done[o] = is_done ? 1 : 0;
            }
            if (IDX) {
                // scipy's witnesses: first maximiser of the row/col minima
                float w[2] = {(valid && rowmin == red[0]) ? (float)i : 1e9f,
                              (valid && colmin == red[1]) ? (float)i : 1e9f};
                env_reduce<G, T, 2, R_MIN, R_MIN, R_MIN, R_MIN>(w, scratch);
                if (valid) {
                    if (a.near_lm) a.near_lm[sidx] = arg_lm;
                    if (a.near_ag) a.near_ag[sidx] = arg_ag;
                    if (a.hd_idx) {
                        if (i == (int)w[0]) { a.hd_idx[b * 4 + 0] = i; a.hd_idx[b * 4 + 1] = arg_lm; }
                        if (i == (int)w[1]) { a.hd_idx[b * 4 + 2] = i; a.hd_idx[b * 4 + 3] = arg_ag; }
                    }
                }
            }

            // ---- phase 4: vec-env auto reset --------------------------------
            if (a.p.auto_reset && reset_flag[k & 1] != 0) {          // workgroup-uniform
                if (tid == 0) reset_flag[(k + 1) & 1] = 0;
                const bool mine = is_done && env_ok;
                if (G > 64 ? mine : (__any(mine) != 0)) {
                    uint32_t c[4] = {(uint32_t)b, (uint32_t)i, (uint32_t)(a.p.rng_offset + k),
                                     (uint32_t)((a.p.rng_offset + k) >> 32)};
                    philox4x32(c, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
                    float raw[2] = {valid ? u_pm1(c[2]) : 0.f, valid ? u_pm1(c[3]) : 0.f};
                    const float rx = raw[0], ry = raw[1];
                    env_reduce<G, T, 2, R_SUM, R_SUM, R_SUM, R_SUM>(raw, scratch);
                    if (mine && valid) {
                        p = make_float2(u_pm1(c[0]), u_pm1(c[1]));
                        v = make_float2(0.f, 0.f);
                        s = make_float2(__builtin_fmaf(-raw[0], invN, rx), __builtin_fmaf(-raw[1], invN, ry));   // explicit fma: same bits in every kernel
                        A[i] = p; V[i] = v; NV[i] = v; A[2 * N - 1 + i] = s;
                        PX[i] = p.x; PY[i] = p.y; SX[i] = s.x; SY[i] = s.y;
                        reinterpret_cast<float2*>(a.shape)[sidx] = s;
                        if (i == 0) {
                            uint32_t c2[4] = {(uint32_t)b, 0xFFFFFFFFu, (uint32_t)(a.p.rng_offset + k),
                                              (uint32_t)((a.p.rng_offset + k) >> 32)};
                            philox4x32(c2, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
                            const float2 niv = make_float2(u_pm1(c2[0]), u_pm1(c2[1]));
                            A[3 * N - 1] = niv;
                            reinterpret_cast<float2*>(a.ivel)[b] = niv;
                        }
                    }
                    if (mine) t_step = 0;
                }
                __syncthreads();
            }

            // ---- phase 5: observations --------------------------------------
            if (want_obs && NC > 0 && !FLAT) {
                if constexpr (NC > 0 && WR == 0)
                    write_obs_rows<NC, T / 64, E>(env_tables(smem, 0, N), env_block_floats(NC) / 2, tid >> 6,
                                                  reinterpret_cast<float2*>(a.obs) +
                                                  ((size_t)slot * a.B + b0) * (size_t)(3 * NC * NC), El, 3);
                if constexpr (NC > 0 && WR >= 2) {
                    const size_t unit0 = ((size_t)slot * a.B + b0) * (size_t)(3 * NC * NC);
                    float2* tiles = env_tables(smem, E, N) + 36;            // after env blocks + 72 floats of scratch
                    write_obs_tiled<NC, T / 64, E, WR - 2 + 1>(env_tables(smem, 0, N), env_block_floats(NC) / 2, tid >> 6, tiles,
                                                               reinterpret_cast<float2*>(a.obs) + unit0, unit0, El);
                }
            } else if (want_obs) {
                const unsigned n3 = 3u * N;                // (x,y) units per row
                const unsigned nenv = n3 * N;              // units per env = N rows
                const size_t U0 = ((size_t)slot * a.B + b0) * nenv;
                const unsigned total = (unsigned)El * nenv;
                const unsigned head = (unsigned)(U0 & 1);  // region start not 16-byte aligned
                float2* const out2 = reinterpret_cast<float2*>(a.obs) + U0;
                // unit (rp, u): rp = e*N + row is the row index inside the group, u the unit in
                // the row.  Branch-free so that the LDS reads of several units overlap.
                auto unit = [&](unsigned rp, unsigned u) -> float2 {
                    const unsigned ee = (E == 1) ? 0u : rp / (unsigned)N;
                    const unsigned row = rp - ee * N;
                    const float2* AA = env_tables(smem, (int)ee, N);
                    const unsigned j = u - 1u;
                    const bool is_delta = j < (unsigned)(N - 1);
                    unsigned idx = is_delta ? j + (j >= row ? 1u : 0u) : u;
                    idx = (u == 0u) ? n3 + row : idx;
                    float2 val = AA[idx];
                    const float2 pi = AA[row];
                    val.x -= is_delta ? pi.x : 0.0f;
                    val.y -= is_delta ? pi.y : 0.0f;
                    return val;
                };
                if (head && tid == 0) out2[0] = unit(0u, 0u);
                const unsigned npair = (total - head) >> 1;
                f32x4* const out4 = reinterpret_cast<f32x4*>(out2 + head);
                const unsigned du = (2u * T) % n3, drow = (2u * T) / n3;
                unsigned q = head + 2u * tid;
                unsigned rp = q / n3;
                unsigned u = q - rp * n3;
#pragma unroll 2
                for (unsigned q2 = tid; q2 < npair; q2 += T) {
                    unsigned u1 = u + 1u, rp1 = rp;
                    if (u1 == n3) { u1 = 0u; rp1 += 1u; }
                    const float2 x0 = unit(rp, u), x1 = unit(rp1, u1);
                    const f32x4 w = {x0.x, x0.y, x1.x, x1.y};
                    out4[q2] = w;
                    u += du; rp += drow;
                    if (u >= n3) { u -= n3; rp += 1u; }
                }
                if (((total - head) & 1u) && tid == T - 1)
                    out2[total - 1] = unit((total - 1) / n3, (total - 1) % n3);
            }
        }

        if (k + 1 < a.K) {
            __syncthreads();            // obs phase done reading A/V before the next step writes them
            if (valid) { QX[i] = p.x; QY[i] = p.y; }
            __syncthreads();
        }
    }

    // ---- state write-back ---------------------------------------------------
    if (valid && (a.do_phys || a.p.auto_reset)) {
        a.px[sidx] = p.x; a.py[sidx] = p.y; a.vx[sidx] = v.x; a.vy[sidx] = v.y;
    }
    if (a.do_phys && a.step && env_ok && i == 0) a.step[b] = t_step;
}

// ---------------------------------------------------------------------------
// Pipelined K-step rollout (N <= 32): wave specialisation inside one workgroup.
//   producer waves (tid < TP): agents on lanes as in step_kernel; they run World.step + reward
//       of step k+1 while
//   writer waves   (tid >= TP): stream the observations of step k,
// handing over through double-buffered LDS tables and ONE workgroup barrier per step.
// In a single-step launch the pair loops and reductions sit in front of the store stream;
// here they hide under it, so the rollout runs at the store rate.  Everything a producer
// needs from other lanes is produced by its own wave (G <= 64): LDS operations of one wave
// complete in order and the reductions are in-register butterflies, so producers need no
// barrier among themselves.
// LDS per env (floats): tables[2][A[3N] | V[N] | NV[N]] (float2), then QX QY PX PY SX SY [NP].
// ---------------------------------------------------------------------------
__host__ __device__ constexpr int roll_block_floats(int n) { return 20 * n + 6 * npad(n); }

template <int NC, int G, int TP, int TW, int E, int WR>
__global__ __launch_bounds__(TP + TW) void rollout_kernel(const Args a) {
    static_assert(G <= 64 && E * G == TP && TW % 64 == 0 && TP % 64 == 0, "bad rollout geometry");
    constexpr int N = NC, NP = npad(NC), NWW = TW / 64;
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float* const smemf = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    const bool producer = tid < TP;
    const int e = producer ? tid / G : 0;
    const int i = tid % G;
    const int b0 = blockIdx.x * E;
    const int b = b0 + e;
    const bool env_ok = producer && (b < a.B);
    const bool valid = env_ok && (i < N);
    const int El = min(E, a.B - b0);
    float* const blk = smemf + e * roll_block_floats(N);
    float2* const TB0 = reinterpret_cast<float2*>(blk);                 // tables of buffer 0; buffer 1 at + 5N
    float* const QX = blk + 20 * N;
    float* const QY = QX + NP; float* const PX = QY + NP; float* const PY = PX + NP;
    float* const SX = PY + NP; float* const SY = SX + NP;

    const float one_minus_damp = 1.0f - a.p.damping;
    const float dt = a.p.dt;
    const float cutoff = a.p.dist_min + 18.0f * a.p.contact_margin;
    const float cutoff2 = cutoff * cutoff;
    const float thr2 = (float)((double)a.p.collide_thresh * (double)a.p.collide_thresh);
    const float invN = 1.0f / (float)N;

    float2 p = make_float2(0.f, 0.f), v = p, s = p, iv = p;
    int t_step = 0;
    const size_t sidx = (size_t)b * N + i;
    if (valid) {
        p = make_float2(a.px[sidx], a.py[sidx]);
        v = make_float2(a.vx[sidx], a.vy[sidx]);
        s = reinterpret_cast<const float2*>(a.shape)[sidx];
        QX[i] = p.x; QY[i] = p.y; SX[i] = s.x; SY[i] = s.y;
        if (i < N - 1) { TB0[N + i] = make_float2(0.f, 0.f); TB0[5 * N + N + i] = make_float2(0.f, 0.f); }
    } else if (env_ok && i < NP) {
        QX[i] = FAR_AWAY; QY[i] = FAR_AWAY; PX[i] = FAR_AWAY; PY[i] = FAR_AWAY; SX[i] = FAR_AWAY; SY[i] = FAR_AWAY;
    }
    if (env_ok) { iv = reinterpret_cast<const float2*>(a.ivel)[b]; if (a.step) t_step = a.step[b]; }

    // one producer step: World.step + reward of step k into table buffer (k & 1)
    auto produce = [&](int k) {
        float2* const A = TB0 + (k & 1) * 5 * N;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (valid) {
            const float2 u = reinterpret_cast<const float2*>(a.act)[((size_t)k * a.B + b) * N + i];
            float2 f = contact_force_packed(QX, QY, NP, i, p, a.p.contact_force, a.p.contact_margin,
                                            a.p.dist_min, cutoff2);
            f.x += a.p.mass * (a.p.sensitivity * u.x);
            f.y += a.p.mass * (a.p.sensitivity * u.y);
            v.x = v.x * one_minus_damp + (f.x / a.p.mass) * dt;
            v.y = v.y * one_minus_damp + (f.y / a.p.mass) * dt;
            p.x += v.x * dt;
            p.y += v.y * dt;
            PX[i] = p.x; PY[i] = p.y;
        }
        t_step += 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float sums[4] = {valid ? p.x : 0.f, valid ? p.y : 0.f, valid ? v.x : 0.f, valid ? v.y : 0.f};
        env_reduce<G, G, 4, R_SUM, R_SUM, R_SUM, R_SUM>(sums, nullptr);
        const float mx = sums[0] * invN, my = sums[1] * invN;
        const float mvx = sums[2] * invN, mvy = sums[3] * invN;
        float rowmin = INFINITY, colmin = INFINITY;
        int cnt = 0, arg_lm = 0, arg_ag = 0;
        if (valid)
            reward_pass_packed<false>(PX, PY, SX, SY, NP, p, p.x - mx, p.y - my, s.x + mx, s.y + my, thr2,
                                      rowmin, colmin, cnt, arg_lm, arg_ag);
        float red[3] = {valid ? rowmin : -INFINITY, valid ? colmin : -INFINITY, (float)cnt};
        env_reduce<G, G, 3, R_MAX, R_MAX, R_SUM, R_SUM>(red, nullptr);
        const float H = sqrtf(fmaxf(red[0], red[1]));
        const float ex = iv.x - mvx, ey = iv.y - mvy;
        const float velterm = sqrtf(ex * ex + ey * ey);
        const bool is_done = t_step >= a.p.world_length;
        if (valid) {
            const size_t o = ((size_t)k * a.B + b) * N + i;
            if (a.rew) a.rew[o] = (float)(-(double)N * ((double)H + (double)velterm) - (double)red[2]);
            if (a.indiv) a.indiv[o] = (-H - velterm) - (float)cnt;
            if (a.done) a.done[o] = is_done ? 1 : 0;
        }
        if (a.p.auto_reset) {
            const bool mine = is_done && env_ok;
            if (__any(mine) != 0) {
                uint32_t c[4] = {(uint32_t)b, (uint32_t)i, (uint32_t)(a.p.rng_offset + k),
                                 (uint32_t)((a.p.rng_offset + k) >> 32)};
                philox4x32(c, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
                float raw[2] = {valid ? u_pm1(c[2]) : 0.f, valid ? u_pm1(c[3]) : 0.f};
                const float rx = raw[0], ry = raw[1];
                env_reduce<G, G, 2, R_SUM, R_SUM, R_SUM, R_SUM>(raw, nullptr);
                uint32_t c2[4] = {(uint32_t)b, 0xFFFFFFFFu, (uint32_t)(a.p.rng_offset + k),
                                  (uint32_t)((a.p.rng_offset + k) >> 32)};
                philox4x32(c2, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
                if (mine) {
                    iv = make_float2(u_pm1(c2[0]), u_pm1(c2[1]));
                    t_step = 0;
                    if (valid) {
                        p = make_float2(u_pm1(c[0]), u_pm1(c[1]));
                        v = make_float2(0.f, 0.f);
                        s = make_float2(__builtin_fmaf(-raw[0], invN, rx), __builtin_fmaf(-raw[1], invN, ry));   // explicit fma: same bits in every kernel
                        SX[i] = s.x; SY[i] = s.y;
                        reinterpret_cast<float2*>(a.shape)[sidx] = s;
                        if (i == 0) reinterpret_cast<float2*>(a.ivel)[b] = iv;
                    }
                }
            }
        }
        if (valid) {                                   // publish this step's tables + next step's partners
            A[i] = p; A[3 * N + i] = v; A[4 * N + i] = make_float2(-v.x, -v.y);
            A[2 * N - 1 + i] = s;
            if (i == 0) A[3 * N - 1] = iv;
            QX[i] = p.x; QY[i] = p.y;
        }
    };

    if (producer) produce(0);
#if FG_WRITER_PRIO
    else __builtin_amdgcn_s_setprio(FG_WRITER_PRIO);       // writer waves win issue arbitration over producers
#endif
    __syncthreads();
    for (int k = 0; k < a.K; ++k) {
        if (producer) {
            if (k + 1 < a.K && !(FG_PROBES && a.probe)) produce(k + 1);      // probe 1/2: writers only
        } else {
            int slot = k;
            bool want_obs = a.obs != nullptr;
            if (a.obs_every > 1) { want_obs = want_obs && ((k + 1) % a.obs_every == 0); slot = k / a.obs_every; }
            if (want_obs) {
                const size_t unit0 = ((size_t)slot * a.B + b0) * (size_t)(3 * NC * NC);
                const float2* tables0 = reinterpret_cast<const float2*>(smemf) + (k & 1) * 5 * N;
                if constexpr (WR == 0)
                    write_obs_rows<NC, NWW, E>(tables0, roll_block_floats(N) / 2, (tid - TP) >> 6,
                                               reinterpret_cast<float2*>(a.obs) + unit0, El, 3);
                else
                    write_obs_tiled<NC, NWW, E, WR - 1>(tables0, roll_block_floats(N) / 2, (tid - TP) >> 6,
                                                        reinterpret_cast<float2*>(smemf + E * roll_block_floats(N)),
                                                        reinterpret_cast<float2*>(a.obs) + unit0, unit0, El);
            }
        }
        if (!(FG_PROBES && a.probe == 2)) __syncthreads();                    // probe 2: no hand-over sync
    }
    if (valid) { a.px[sidx] = p.x; a.py[sidx] = p.y; a.vx[sidx] = v.x; a.vy[sidx] = v.y; }
    if (a.step && env_ok && i == 0) a.step[b] = t_step;
}

// ---------------------------------------------------------------------------
// Pipelined K-step rollout for 64 < N <= 256: the same producer / writer split, with ONE
// producer wave per environment holding A = ceil(N/64) agents per lane (agent lane + 64 a), so
// that every reduction stays inside the wave and producers still need no barrier of their own.
// The partner loops load each partner pair once and update all A agents of the lane.
// ---------------------------------------------------------------------------
template <int NC, int A, int E, int TW>
__global__ __launch_bounds__(E * 64 + TW) void rollout_kernel_wide(const Args a) {
    static_assert(A * 64 >= NC && (A - 1) * 64 < NC && TW % 64 == 0, "bad wide rollout geometry");
    constexpr int N = NC, NP = npad(NC), NWW = TW / 64, TP = E * 64;
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float* const smemf = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const bool producer = tid < TP;
    const int e = producer ? tid >> 6 : 0;
    // K > 1: the workgroup owns E envs for K steps.  K == 1 (`groups` > 1): it owns `groups`
    // consecutive batches of E envs and pipelines over the batches instead of over the steps.
    const int NG = (a.K > 1) ? 1 : max(1, a.groups);
    const int wg0 = blockIdx.x * E * NG;
    int b = wg0 + e;
    bool env_ok = producer && (b < a.B);
    float* const blk = smemf + e * roll_block_floats(N);
    float2* const TB0 = reinterpret_cast<float2*>(blk);
    float* const QX = blk + 20 * N;
    float* const QY = QX + NP; float* const PX = QY + NP; float* const PY = PX + NP;
    float* const SX = PY + NP; float* const SY = SX + NP;

    const float one_minus_damp = 1.0f - a.p.damping;
    const float dt = a.p.dt;
    const float cutoff = a.p.dist_min + 18.0f * a.p.contact_margin;
    const float cutoff2 = cutoff * cutoff;
    const float thr2 = (float)((double)a.p.collide_thresh * (double)a.p.collide_thresh);
    const float invN = 1.0f / (float)N;
    const float inv_k = 1.0f / a.p.contact_margin;

    float2 p[A], v[A], s[A];
    bool valid[A];
    float2 iv = make_float2(0.f, 0.f);
    int t_step = 0;
    if (producer) {                                     // row-independent table entries and loop sentinels: once
#pragma unroll
        for (int q = 0; q < A; ++q) {
            const int i = lane + 64 * q;
            if (i < N - 1) { TB0[N + i] = make_float2(0.f, 0.f); TB0[5 * N + N + i] = make_float2(0.f, 0.f); }
            if (i >= N && i < NP) {
                QX[i] = FAR_AWAY; QY[i] = FAR_AWAY; PX[i] = FAR_AWAY; PY[i] = FAR_AWAY; SX[i] = FAR_AWAY; SY[i] = FAR_AWAY;
            }
        }
    }
    auto load_group = [&](int g) {                      // state of env batch g -> registers + partner arrays
        b = wg0 + g * E + e;
        env_ok = producer && (b < a.B);
#pragma unroll
        for (int q = 0; q < A; ++q) {
            const int i = lane + 64 * q;
            valid[q] = env_ok && i < N;
            p[q] = v[q] = s[q] = make_float2(0.f, 0.f);
            if (valid[q]) {
                const size_t o = (size_t)b * N + i;
                p[q] = make_float2(a.px[o], a.py[o]);
                v[q] = make_float2(a.vx[o], a.vy[o]);
                s[q] = reinterpret_cast<const float2*>(a.shape)[o];
                QX[i] = p[q].x; QY[i] = p[q].y; SX[i] = s[q].x; SY[i] = s[q].y;
            }
        }
        iv = make_float2(0.f, 0.f); t_step = 0;
        if (env_ok) { iv = reinterpret_cast<const float2*>(a.ivel)[b]; if (a.step) t_step = a.step[b]; }
    };
    auto store_group = [&]() {
#pragma unroll
        for (int q = 0; q < A; ++q) {
            if (valid[q]) {
                const size_t o = (size_t)b * N + lane + 64 * q;
                a.px[o] = p[q].x; a.py[o] = p[q].y; a.vx[o] = v[q].x; a.vy[o] = v[q].y;
            }
        }
        if (a.step && env_ok && lane == 0) a.step[b] = t_step;
    };

    auto produce = [&](int k, int buf) {
        float2* const T = TB0 + buf * 5 * N;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ---- World.step: all A agents of the lane against each partner pair ----
        float fx[A], fy[A];
#pragma unroll
        for (int q = 0; q < A; ++q) { fx[q] = 0.f; fy[q] = 0.f; }
        if (env_ok) {
            for (int j = 0; j < NP; j += 2) {
                const f32x2 qx = *reinterpret_cast<const f32x2*>(QX + j);
                const f32x2 qy = *reinterpret_cast<const f32x2*>(QY + j);
#pragma unroll
                for (int q = 0; q < A; ++q) {
                    const int i = lane + 64 * q;
                    const f32x2 dx = (f32x2){p[q].x, p[q].x} - qx, dy = (f32x2){p[q].y, p[q].y} - qy;
                    const f32x2 d2 = dx * dx + dy * dy;
                    const bool n0 = (d2.x < cutoff2) && (j != i) && valid[q];
                    const bool n1 = (d2.y < cutoff2) && (j + 1 != i) && valid[q];
                    if (n0 || n1) {
                        auto add = [&](float ddx, float ddy, float dd2) {
                            const float d = __builtin_amdgcn_sqrtf(dd2);
                            const float x = (a.p.dist_min - d) * inv_k;
                            const float pen = a.p.contact_margin * (fmaxf(x, 0.0f) + __logf(1.0f + __expf(-fabsf(x))));
                            const float c = a.p.contact_force * pen * __builtin_amdgcn_rcpf(d);
                            fx[q] += ddx * c; fy[q] += ddy * c;
                        };
                        if (n0) add(dx.x, dy.x, d2.x);
                        if (n1) add(dx.y, dy.y, d2.y);
                    }
                }
            }
        }
        // float sums are reduced per 64-agent slice and then combined slice by slice: the exact
        // association order of step_kernel (wave butterfly, then waves in order) -> bit-identical
        float sums[4] = {0.f, 0.f, 0.f, 0.f};
        float part[A][4];
#pragma unroll
        for (int q = 0; q < A; ++q) {
            part[q][0] = part[q][1] = part[q][2] = part[q][3] = 0.f;
            if (valid[q]) {
                const int i = lane + 64 * q;
                const float2 u = reinterpret_cast<const float2*>(a.act)[((size_t)k * a.B + b) * N + i];
                const float ffx = fx[q] + a.p.mass * (a.p.sensitivity * u.x);
                const float ffy = fy[q] + a.p.mass * (a.p.sensitivity * u.y);
                v[q].x = v[q].x * one_minus_damp + (ffx / a.p.mass) * dt;
                v[q].y = v[q].y * one_minus_damp + (ffy / a.p.mass) * dt;
                p[q].x += v[q].x * dt;
                p[q].y += v[q].y * dt;
                PX[i] = p[q].x; PY[i] = p[q].y;
                part[q][0] = p[q].x; part[q][1] = p[q].y; part[q][2] = v[q].x; part[q][3] = v[q].y;
            }
        }
        t_step += 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < A; ++q) {
            env_reduce<64, 64, 4, R_SUM, R_SUM, R_SUM, R_SUM>(part[q], nullptr);
            if (q == 0) { sums[0] = part[0][0]; sums[1] = part[0][1]; sums[2] = part[0][2]; sums[3] = part[0][3]; }
            else { sums[0] += part[q][0]; sums[1] += part[q][1]; sums[2] += part[q][2]; sums[3] += part[q][3]; }
        }
        const float mx = sums[0] * invN, my = sums[1] * invN;
        const float mvx = sums[2] * invN, mvy = sums[3] * invN;
        // ---- reward pass ----
        float rowmin[A], colmin[A];
        int cnt[A];
#pragma unroll
        for (int q = 0; q < A; ++q) { rowmin[q] = INFINITY; colmin[q] = INFINITY; cnt[q] = -1; }
        if (env_ok) {
            for (int j = 0; j < NP; j += 2) {
                const f32x2 qx = *reinterpret_cast<const f32x2*>(PX + j);
                const f32x2 qy = *reinterpret_cast<const f32x2*>(PY + j);
                const f32x2 sx = *reinterpret_cast<const f32x2*>(SX + j);
                const f32x2 sy = *reinterpret_cast<const f32x2*>(SY + j);
#pragma unroll
                for (int q = 0; q < A; ++q) {
                    const f32x2 cx = qx - (f32x2){p[q].x, p[q].x}, cy = qy - (f32x2){p[q].y, p[q].y};
                    const f32x2 dc = cx * cx + cy * cy;
                    cnt[q] += (dc.x < thr2 ? 1 : 0) + (dc.y < thr2 ? 1 : 0);
                    const float ptx = p[q].x - mx, pty = p[q].y - my;
                    const f32x2 rx = (f32x2){ptx, ptx} - sx, ry = (f32x2){pty, pty} - sy;
                    const f32x2 dr = rx * rx + ry * ry;
                    const float tx = s[q].x + mx, ty = s[q].y + my;
                    const f32x2 ux = qx - (f32x2){tx, tx}, uy = qy - (f32x2){ty, ty};
                    const f32x2 dq = ux * ux + uy * uy;
                    rowmin[q] = fminf(fminf(rowmin[q], dr.x), dr.y);
                    colmin[q] = fminf(fminf(colmin[q], dq.x), dq.y);
                }
            }
        }
        float red[3] = {-INFINITY, -INFINITY, 0.f};
#pragma unroll
        for (int q = 0; q < A; ++q) {
            if (valid[q]) {
                cnt[q] += (thr2 > 0.0f ? 0 : 1);
                red[0] = fmaxf(red[0], rowmin[q]); red[1] = fmaxf(red[1], colmin[q]); red[2] += (float)cnt[q];
            }
        }
        env_reduce<64, 64, 3, R_MAX, R_MAX, R_SUM, R_SUM>(red, nullptr);
        const float H = sqrtf(fmaxf(red[0], red[1]));
        const float ex = iv.x - mvx, ey = iv.y - mvy;
        const float velterm = sqrtf(ex * ex + ey * ey);
        const bool is_done = t_step >= a.p.world_length;
        const float shared = (float)(-(double)N * ((double)H + (double)velterm) - (double)red[2]);
#pragma unroll
        for (int q = 0; q < A; ++q) {
            if (valid[q]) {
                const size_t o = ((size_t)k * a.B + b) * N + lane + 64 * q;
                if (a.rew) a.rew[o] = shared;
                if (a.indiv) a.indiv[o] = (-H - velterm) - (float)cnt[q];
                if (a.done) a.done[o] = is_done ? 1 : 0;
            }
        }
        if (a.p.auto_reset && is_done && env_ok) {            // wave-uniform: the wave owns one env
            float raw[2] = {0.f, 0.f};
            float rawp[A][2];
            float rx[A], ry[A];
            uint32_t c0[A], c1[A];
#pragma unroll
            for (int q = 0; q < A; ++q) {
                uint32_t c[4] = {(uint32_t)b, (uint32_t)(lane + 64 * q), (uint32_t)(a.p.rng_offset + k),
                                 (uint32_t)((a.p.rng_offset + k) >> 32)};
                philox4x32(c, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
                c0[q] = c[0]; c1[q] = c[1];
                rx[q] = valid[q] ? u_pm1(c[2]) : 0.f; ry[q] = valid[q] ? u_pm1(c[3]) : 0.f;
                rawp[q][0] = rx[q]; rawp[q][1] = ry[q];
            }
#pragma unroll
            for (int q = 0; q < A; ++q) {
                env_reduce<64, 64, 2, R_SUM, R_SUM, R_SUM, R_SUM>(rawp[q], nullptr);
                if (q == 0) { raw[0] = rawp[0][0]; raw[1] = rawp[0][1]; } else { raw[0] += rawp[q][0]; raw[1] += rawp[q][1]; }
            }
            uint32_t c2[4] = {(uint32_t)b, 0xFFFFFFFFu, (uint32_t)(a.p.rng_offset + k),
                              (uint32_t)((a.p.rng_offset + k) >> 32)};
            philox4x32(c2, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
            iv = make_float2(u_pm1(c2[0]), u_pm1(c2[1]));
            t_step = 0;
#pragma unroll
            for (int q = 0; q < A; ++q) {
                if (valid[q]) {
                    const int i = lane + 64 * q;
                    const size_t o = (size_t)b * N + i;
                    p[q] = make_float2(u_pm1(c0[q]), u_pm1(c1[q]));
                    v[q] = make_float2(0.f, 0.f);
                    s[q] = make_float2(__builtin_fmaf(-raw[0], invN, rx[q]), __builtin_fmaf(-raw[1], invN, ry[q]));
                    SX[i] = s[q].x; SY[i] = s[q].y;
                    reinterpret_cast<float2*>(a.shape)[o] = s[q];
                }
            }
            if (lane == 0) reinterpret_cast<float2*>(a.ivel)[b] = iv;
        }
#pragma unroll
        for (int q = 0; q < A; ++q) {
            if (valid[q]) {
                const int i = lane + 64 * q;
                T[i] = p[q]; T[3 * N + i] = v[q]; T[4 * N + i] = make_float2(-v[q].x, -v[q].y);
                T[2 * N - 1 + i] = s[q];
                QX[i] = p[q].x; QY[i] = p[q].y;
            }
        }
        if (env_ok && lane == 0) T[3 * N - 1] = iv;
    };

    const bool per_step = a.K == 1;
    const int total = per_step ? NG : a.K;
    if (producer) { load_group(0); produce(0, 0); if (per_step) store_group(); }
    __syncthreads();
    for (int it = 0; it < total; ++it) {
        if (producer) {
            if (it + 1 < total) {
                if (per_step) { load_group(it + 1); produce(0, (it + 1) & 1); store_group(); }
                else produce(it + 1, (it + 1) & 1);
            }
        } else {
            const int k = per_step ? 0 : it;
            const int b0 = wg0 + (per_step ? it * E : 0);
            const int El = min(E, a.B - b0);
            int slot = k;
            bool want_obs = a.obs != nullptr && El > 0;
            if (a.obs_every > 1) { want_obs = want_obs && ((k + 1) % a.obs_every == 0); slot = k / a.obs_every; }
            if (want_obs) {
                const size_t unit0 = ((size_t)slot * a.B + b0) * (size_t)(3 * NC * NC);
                write_obs_rows<NC, NWW, E>(reinterpret_cast<const float2*>(smemf) + (it & 1) * 5 * N,
                                           roll_block_floats(N) / 2, (tid - TP) >> 6,
                                           reinterpret_cast<float2*>(a.obs) + unit0, El, 3);
            }
        }
        __syncthreads();
    }
    if (!per_step && producer) store_group();
}

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
    uint32_t c[4] = {(uint32_t)b, (uint32_t)i, (uint32_t)a.p.rng_offset, (uint32_t)(a.p.rng_offset >> 32)};
    philox4x32(c, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
    float raw[2] = {valid ? u_pm1(c[2]) : 0.f, valid ? u_pm1(c[3]) : 0.f};
    const float rx = raw[0], ry = raw[1];
    env_reduce<G, T, 2, R_SUM, R_SUM, R_SUM, R_SUM>(raw, scratch);
    if (mine) {
        const size_t sidx = (size_t)b * N + i;
        const float invN = 1.0f / (float)N;
        a.px[sidx] = u_pm1(c[0]); a.py[sidx] = u_pm1(c[1]);
        a.vx[sidx] = 0.f; a.vy[sidx] = 0.f;
        reinterpret_cast<float2*>(a.shape)[sidx] = make_float2(__builtin_fmaf(-raw[0], invN, rx), __builtin_fmaf(-raw[1], invN, ry));
        if (i == 0) {
            uint32_t c2[4] = {(uint32_t)b, 0xFFFFFFFFu, (uint32_t)a.p.rng_offset, (uint32_t)(a.p.rng_offset >> 32)};
            philox4x32(c2, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
            reinterpret_cast<float2*>(a.ivel)[b] = make_float2(u_pm1(c2[0]), u_pm1(c2[1]));
            if (a.step) a.step[b] = 0;
        }
    }
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
__global__ __launch_bounds__(256) void mt_reset_kernel(int B, int N, const uint8_t* __restrict__ mask,
                                                       uint32_t* __restrict__ mt_state,
                                                       float* px, float* py, float* vx, float* vy,
                                                       float* shape, float* ivel, float* lm_pos, int32_t* step) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_u32[];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= B || (mask && !mask[b])) return;
    uint32_t* const mt = lds_u32;                      // [624]
    uint32_t* const outs = lds_u32 + 624;              // [8N + 4] tempered outputs
    double* const dsum = reinterpret_cast<double*>(lds_u32 + 624 + ((8 * N + 4 + 1) & ~1));   // [2] mean of raw
    uint32_t* const gstate = mt_state + (size_t)b * 626;
    for (int q = tid; q < 624; q += 256) mt[q] = gstate[q];
    int pos = (int)gstate[624];
    __syncthreads();
    const int M = 8 * N + 4;
    int produced = 0;
    auto mix = [](uint32_t a, uint32_t b2) -> uint32_t {
        const uint32_t y = (a & 0x80000000u) | (b2 & 0x7fffffffu);
        return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    };
    while (produced < M) {
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
        const int take = min(624 - pos, M - produced);
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
        px[o] = (float)draw(2 * i); py[o] = (float)draw(2 * i + 1);
        vx[o] = 0.f; vy[o] = 0.f;
        const double rx = draw(2 * N + 2 * i), ry = draw(2 * N + 2 * i + 1);
        shape[2 * o] = (float)(rx - dsum[0]); shape[2 * o + 1] = (float)(ry - dsum[1]);
        if (lm_pos) { lm_pos[2 * o] = (float)rx; lm_pos[2 * o + 1] = (float)ry; }
    }
    if (tid == 0) {
        ivel[2 * b] = (float)draw(4 * N); ivel[2 * b + 1] = (float)draw(4 * N + 1);
        if (step) step[b] = 0;
        gstate[624] = (uint32_t)pos;
    }
    for (int q = tid; q < 624; q += 256) gstate[q] = mt[q];
}

// ---------------------------------------------------------------------------
// Landmark scenarios with few agents (N + M <= 64): basic_formation_env (BASELINE config 1),
// formation_hd_partial_env, formation_hd_partial_range_env, formation_hd_obs_env.
// One lane per movable entity (N agents, then M obstacles), one env per aligned group of G
// lanes of a wave.  Reference lines under formation_gym/envs/:
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
    const float* act; const float* lm; float* opos; float* ovel; int32_t* step;
    float* obs; float* rew; float* indiv; uint8_t* done; int32_t* near_ag;
};

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
    float2* const PRE = smem + e * (2 * NE + L);
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
    const float my_size = 0.5f * (i < N ? a.p.dist_min : 2.0f * a.sc.obstacle_size);
    if (a.do_phys) {
        if (is_agent || is_obst) {
            // World.step: all pairs of movable colliders, contact distance size_i + size_j
            float fx = 0.f, fy = 0.f;
            const float k = a.p.contact_margin;
            for (int j = 0; j < NE; ++j) {
                const float2 q = PRE[j];
                const float dmin = my_size + 0.5f * (j < N ? a.p.dist_min : 2.0f * a.sc.obstacle_size);
                const float cut = dmin + 18.0f * k;
                const float dx = p.x - q.x, dy = p.y - q.y;
                const float d2 = dx * dx + dy * dy;
                if (j != i && d2 < cut * cut) {
                    const float d = __builtin_amdgcn_sqrtf(d2);
                    const float x = (dmin - d) / k;
                    const float pen = k * (fmaxf(x, 0.0f) + __logf(1.0f + __expf(-fabsf(x))));
                    const float c = a.p.contact_force * pen * __builtin_amdgcn_rcpf(d);
                    fx += dx * c; fy += dy * c;
                }
            }
            if (is_agent) {
                const float2 u = reinterpret_cast<const float2*>(a.act)[sidx];
                const float2 fa = action_force(a.p, u, (uint32_t)b, (uint32_t)i, a.p.rng_offset);
                fx += fa.x; fy += fa.y;
            }
            if (a.p.num_walls > 0) wall_forces(a.p, p, my_size, fx, fy);
            v.x = v.x * (1.0f - a.p.damping) + (fx / a.p.mass) * a.p.dt;
            v.y = v.y * (1.0f - a.p.damping) + (fy / a.p.mass) * a.p.dt;
            if (is_agent) v = clamp_speed(a.p, v);
            p.x += v.x * a.p.dt; p.y += v.y * a.p.dt;
            POST[i] = p;
            if (is_agent) {
                a.px[sidx] = p.x; a.py[sidx] = p.y; a.vx[sidx] = v.x; a.vy[sidx] = v.y;
            } else {
                // the reward callback re-arms the obstacle velocity every step (:84-89)
                const bool falling = p.y > a.sc.obstacle_floor;
                reinterpret_cast<float2*>(a.opos)[oidx] = p;
                reinterpret_cast<float2*>(a.ovel)[oidx] =
                    make_float2(falling ? a.sc.obstacle_vx : 0.f, falling ? a.sc.obstacle_vy : 0.f);
            }
        }
        t_step += 1;
        __syncthreads();
    }
    float scratch_dummy[1];
    // ---- formation term ----
    float form = 0.f;      // basic: sum_l min_a |p_a - l| ; others: Hausdorff(centred agents, centred landmarks)
    if (kind == FG_SCN_BASIC) {
        float cover = 0.f;
        for (int l0 = 0; l0 < L; l0 += G) {
            const int l = l0 + i;
            if (live && l < L) {
                const float2 m = LM[l];
                float best = INFINITY; int barg = 0;
                for (int j = 0; j < N; ++j) {
                    const float2 q = POST[j];
                    const float dx = q.x - m.x, dy = q.y - m.y, d2 = dx * dx + dy * dy;
                    if (d2 < best) { best = d2; barg = j; }
                }
                cover += sqrtf(best);
                if (a.near_ag) a.near_ag[(size_t)b * L + l] = barg;
            }
        }
        float red[1] = {cover};
        env_reduce<G, G, 1, R_SUM, R_SUM, R_SUM, R_SUM>(red, scratch_dummy);
        form = red[0];
    } else {
        float s4[4] = {is_agent ? p.x : 0.f, is_agent ? p.y : 0.f, 0.f, 0.f};
        for (int l = i; live && l < L; l += G) { s4[2] += LM[l].x; s4[3] += LM[l].y; }
        env_reduce<G, G, 4, R_SUM, R_SUM, R_SUM, R_SUM>(s4, scratch_dummy);
        const float mx = s4[0] / (float)N, my = s4[1] / (float)N;
        const float lx = s4[2] / (float)L, ly = s4[3] / (float)L;
        float rowmin = -INFINITY, colmax = -INFINITY;
        if (is_agent) {                                         // min over landmarks for my agent
            rowmin = INFINITY;
            for (int l = 0; l < L; ++l) {
                const float dx = (p.x - mx) - (LM[l].x - lx), dy = (p.y - my) - (LM[l].y - ly);
                rowmin = fminf(rowmin, dx * dx + dy * dy);
            }
        }
        for (int l = i; live && l < L; l += G) {                // min over agents for my landmark(s)
            float cm = INFINITY;
            for (int j = 0; j < N; ++j) {
                const float dx = (POST[j].x - mx) - (LM[l].x - lx), dy = (POST[j].y - my) - (LM[l].y - ly);
                cm = fminf(cm, dx * dx + dy * dy);
            }
            colmax = fmaxf(colmax, cm);
        }
        float red[2] = {rowmin, colmax};
        env_reduce<G, G, 2, R_MAX, R_MAX, R_MAX, R_MAX>(red, scratch_dummy);
        form = sqrtf(fmaxf(red[0], red[1]));
    }
    // ---- collision counts ----
    int cnt = 0;
    if (is_agent) {
        const float thr = a.p.collide_thresh, thr2 = (float)((double)thr * (double)thr);
        for (int j = 0; j < N; ++j) {
            const float dx = POST[j].x - p.x, dy = POST[j].y - p.y;
            cnt += ((kind == FG_SCN_BASIC || j != i) && dx * dx + dy * dy < thr2) ? 1 : 0;
        }
        const float ot = 0.5f * a.p.dist_min + a.sc.obstacle_size, ot2 = (float)((double)ot * (double)ot);
        for (int j = N; j < NE; ++j) {
            const float dx = POST[j].x - p.x, dy = POST[j].y - p.y;
            cnt += (dx * dx + dy * dy < ot2) ? 1 : 0;
        }
    }
    float cs[1] = {(float)cnt};
    env_reduce<G, G, 1, R_SUM, R_SUM, R_SUM, R_SUM>(cs, scratch_dummy);
    const bool is_done = t_step >= a.p.world_length;
    // ---- outputs ----
    const int nbr = (kind == FG_SCN_PARTIAL) ? a.sc.num_obs : (N - 1);
    const int D = 2 + (kind == FG_SCN_BASIC ? 2 : 0) + 2 * L + 2 * M + 2 * nbr + 2 * (N - 1);
    if (is_agent) {
        if (a.rew) a.rew[sidx] = (float)(-(double)N * (double)form - (double)a.sc.penalty * (double)cs[0]);
        if (a.indiv) a.indiv[sidx] = -form - a.sc.penalty * (float)cnt;
        if (a.done) a.done[sidx] = is_done ? 1 : 0;
        float2* o = reinterpret_cast<float2*>(a.obs + sidx * D);
        int w = 0;
        o[w++] = v;
        if (kind == FG_SCN_BASIC) o[w++] = p;
        for (int l = 0; l < L; ++l) {
            const float2 m = LM[l];
            o[w++] = (kind == FG_SCN_BASIC) ? make_float2(m.x - p.x, m.y - p.y) : m;
        }
        for (int j = N; j < NE; ++j) { const float2 q = POST[j]; o[w++] = make_float2(q.x - p.x, q.y - p.y); }
        if (kind == FG_SCN_PARTIAL) {
            for (int kk = 0; kk < nbr; ++kk) {
                const float2 q = POST[(i + 1 + kk) % N];
                o[w++] = make_float2(q.x - p.x, q.y - p.y);
            }
        } else {
            const float r = (kind == FG_SCN_RANGE) ? a.sc.obs_range : INFINITY;
            for (int j = 0; j < N; ++j) if (j != i) {
                const float2 q = POST[j];
                o[w++] = make_float2(fminf(fmaxf(q.x - p.x, -r), r), fminf(fmaxf(q.y - p.y, -r), r));
            }
        }
        for (int j = 0; j < N - 1; ++j) o[w++] = make_float2(0.f, 0.f);
    }
    if (a.do_phys && a.step && live && i == 0) a.step[b] = t_step;
}

// ---------------------------------------------------------------------------
// host side: geometry selection, validation, launches
// ---------------------------------------------------------------------------
static thread_local char g_err[256] = "";

static int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

struct Geometry { int G, T, E, lds; };

static int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

template <int NC, int G, int T, int E, bool IDX, int WR, bool OPTS>
static hipError_t launch_v(const Args& a, int grid, int lds, hipStream_t st) {
    hipLaunchKernelGGL((step_kernel<NC, G, T, E, IDX, WR, OPTS>), dim3(grid), dim3(T), lds, st, a);
    return hipGetLastError();
}

using LaunchFn = hipError_t (*)(const Args&, int, int, hipStream_t);
// wr: observation writer 0 = register-cached rows, 1 = flat decode (run-time N), 1 + RT = LDS tiles of RT rows
struct Variant { int NC, G, T, E, wr; LaunchFn plain, idx, opts; };   // opts: IDX + World options
#define FG_VARIANT_W(NC, G, T, E, W) {NC, G, T, E, W, &launch_v<NC, G, T, E, false, W, false>, \
                                      &launch_v<NC, G, T, E, true, W, false>, nullptr}
#define FG_VARIANT_O(NC, G, T, E, W) {NC, G, T, E, W, &launch_v<NC, G, T, E, false, W, false>, \
                                      &launch_v<NC, G, T, E, true, W, false>, &launch_v<NC, G, T, E, true, W, true>}
#define FG_VARIANT(NC, G, T, E) FG_VARIANT_W(NC, G, T, E, ((NC) == 0 ? 1 : 0))
#define FG_VARIANT_FLAT(NC, G, T, E) FG_VARIANT_W(NC, G, T, E, 1)

// The first entry of a given NC is the default; the others are selectable with
// FG_GEOM="T,E" (tuning aid, see profiles/).  NC = 0 entries take N at run time.
static const Variant kVariants[] = {
    FG_VARIANT_O(3, 4, 128, 16, 0), FG_VARIANT(3, 4, 64, 16), FG_VARIANT(3, 4, 64, 8),
    FG_VARIANT_O(9, 16, 128, 4, 0), FG_VARIANT(9, 16, 64, 4), FG_VARIANT(9, 16, 64, 2), FG_VARIANT(9, 16, 128, 8),
    FG_VARIANT_O(27, 32, 256, 4, 0), FG_VARIANT(27, 32, 128, 4), FG_VARIANT(27, 32, 64, 2), FG_VARIANT(27, 32, 256, 8),
    FG_VARIANT(27, 32, 128, 2), FG_VARIANT(27, 32, 256, 2), FG_VARIANT(27, 32, 512, 4),
    FG_VARIANT_O(81, 128, 128, 1, 0), FG_VARIANT(81, 128, 256, 1), FG_VARIANT(81, 128, 512, 1),
    FG_VARIANT_O(243, 256, 256, 1, 0), FG_VARIANT(243, 256, 512, 1),
    // flat float4 writer kept for A/B runs (FG_FLAT=1)
    FG_VARIANT_FLAT(27, 32, 256, 4), FG_VARIANT_FLAT(27, 32, 128, 4), FG_VARIANT_FLAT(9, 16, 128, 4),
    FG_VARIANT_FLAT(81, 128, 128, 1), FG_VARIANT_FLAT(243, 256, 256, 1),
    // LDS-tiled writer, RT rows per tile (FG_FLAT = 1 + RT)
    FG_VARIANT_W(27, 32, 256, 4, 4), FG_VARIANT_W(27, 32, 128, 4, 4), FG_VARIANT_W(27, 32, 128, 2, 4), FG_VARIANT_W(27, 32, 64, 2, 4),
    FG_VARIANT_W(27, 32, 256, 4, 10), FG_VARIANT_W(27, 32, 128, 4, 10), FG_VARIANT_W(27, 32, 128, 2, 10), FG_VARIANT_W(27, 32, 64, 2, 10),
    FG_VARIANT_W(27, 32, 256, 2, 10), FG_VARIANT_W(27, 32, 256, 2, 4),
    FG_VARIANT_W(9, 16, 128, 4, 10), FG_VARIANT_W(9, 16, 64, 4, 10),
    FG_VARIANT_O(0, 4, 64, 16, 1), FG_VARIANT_O(0, 8, 64, 8, 1), FG_VARIANT_O(0, 16, 64, 4, 1), FG_VARIANT_O(0, 32, 128, 4, 1),
    FG_VARIANT_O(0, 64, 128, 2, 1), FG_VARIANT_O(0, 128, 128, 1, 1), FG_VARIANT_O(0, 256, 256, 1, 1),
    FG_VARIANT_O(0, 512, 512, 1, 1), FG_VARIANT_O(0, 1024, 1024, 1, 1),
};

static const Variant* variant_for(int N, int B = 0, bool need_opts = false) {
    if (N < 2 || N > FG_MAX_AGENTS) return nullptr;
    int want_t = 0, want_e = 0, want_flat = 0;
    if (need_opts) {                      // the first entry of every NC (and every generic one) carries OPTS
        for (const Variant& v : kVariants) if (v.NC == N && v.opts) return &v;
        B = 0;
    }
    // Size-aware default (MI355X sweep, profiles/): a batch that fills the chip many times over
    // streams best with 8 envs per workgroup; a single-generation batch (27 x 4096 = 4 workgroups
    // per CU) is latency-bound and prefers one env per wave with spare writer waves.
    if (N == 27 && B >= 32768) { want_t = 256; want_e = 8; }
    if (const char* s = getenv("FG_GEOM")) sscanf(s, "%d,%d", &want_t, &want_e);
    if (const char* s = getenv("FG_FLAT")) want_flat = atoi(s);
    const Variant* dflt = nullptr;
    for (const Variant& v : kVariants) {
        if (need_opts) break;
        if (v.NC != N || v.wr != want_flat) continue;
        if (!dflt) dflt = &v;
        if (v.T == want_t && v.E == want_e) return &v;
    }
    if (dflt) return dflt;
    const int G = N <= 64 ? (pow2ceil(N) < 4 ? 4 : pow2ceil(N)) : (pow2ceil(N) < 128 ? 128 : pow2ceil(N));
    for (const Variant& v : kVariants)
        if (v.NC == 0 && v.G == G) return &v;
    return nullptr;
}

static bool geometry_for(int N, Geometry* g, int B = 0, bool need_opts = false) {
    const Variant* v = variant_for(N, B, need_opts);
    if (!v) return false;
    g->G = v->G; g->T = v->T; g->E = v->E;
    g->lds = v->E * env_block_floats(N) * (int)sizeof(float) + 72 * (int)sizeof(float);
    if (v->wr >= 2) g->lds += 2 * (v->T / 64) * ((3 * N * (v->wr - 1) + 3) & ~1) * (int)sizeof(float2);
    return true;
}

static int launch_step(Args a, hipStream_t st) {
    Geometry g;
    const bool opts = a.p.num_walls > 0 || a.p.u_noise > 0.f || a.p.max_speed > 0.f || a.p.accel > 0.f;
    const Variant* v = variant_for(a.N, a.B, opts);
    if (!v || !geometry_for(a.N, &g, a.B, opts)) return fail(FG_ERR_UNSUPPORTED_N, "N must be in [2, 1024]%s");
    const bool idx = a.near_lm || a.near_ag || a.hd_idx;
    const int grid = (a.B + g.E - 1) / g.E;
    const hipError_t err = (opts ? v->opts : idx ? v->idx : v->plain)(a, grid, g.lds, st);
    if (err != hipSuccess) return fail(FG_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

// 64 < N <= 256: producer / writer pipelined kernel (rollout: over steps; single step: over env batches)
static int launch_wide(Args a, hipStream_t st) {
    const int N = a.N, B = a.B;
    hipError_t err = hipSuccess;
    int tw = 256;
    if (const char* e = getenv("FG_TW")) tw = atoi(e);
    if (a.K == 1) {
        // env batches per workgroup: enough to overlap batch g+1's pair loops with batch g's store
        // stream, few enough to keep every CU busy (MI355X sweep, profiles/README.md)
        const int E = (N == 243 && tw == 128) ? 2 : ((N == 81 && tw == 512) ? 8 : 4);
        const int batches = (B + E - 1) / E;
        int target = 256;                                  // workgroups in the grid: one per CU
        if (const char* e = getenv("FG_STEPWG")) target = atoi(e);
        a.groups = batches / (target > 0 ? target : 1);
        if (a.groups < 1) a.groups = 1;
        if (a.groups > 64) a.groups = 64;
    } else {
        a.groups = 1;
    }
#define FG_ROLLW(NCV, AV, EV, TWV)                                                                        \
    {   const int grid = (B + (EV) * a.groups - 1) / ((EV) * a.groups);                                  \
        const int lds = (EV) * roll_block_floats(NCV) * (int)sizeof(float);                              \
        if (lds > 64 * 1024) {   /* more than the default dynamic-LDS limit: opt in (once per kernel) */  \
            static bool raised = false;                                                                  \
            if (!raised) { (void)hipFuncSetAttribute((const void*)&rollout_kernel_wide<NCV, AV, EV, TWV>,          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds); raised = true; } }      \
        hipLaunchKernelGGL((rollout_kernel_wide<NCV, AV, EV, TWV>), dim3(grid), dim3((EV) * 64 + (TWV)), lds, st, a); \
        err = hipGetLastError(); }
    if (N == 81) { if (tw == 128) FG_ROLLW(81, 2, 4, 128) else if (tw == 512) FG_ROLLW(81, 2, 8, 512) else FG_ROLLW(81, 2, 4, 256) }
    else { if (tw == 128) FG_ROLLW(243, 4, 2, 128) else if (tw == 512) FG_ROLLW(243, 4, 4, 512) else FG_ROLLW(243, 4, 4, 256) }
#undef FG_ROLLW
    if (err != hipSuccess) return fail(FG_ERR_HIP, "pipelined launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

static int check_params(const FgParams* p) {
    if (!p) return fail(FG_ERR_BAD_ARG, "params is NULL%s");
    if (!(p->mass > 0.f) || !(p->contact_margin > 0.f) || !(p->dt > 0.f))
        return fail(FG_ERR_BAD_ARG, "params: mass, contact_margin and dt must be > 0%s");
    if (p->num_walls < 0 || p->num_walls > FG_MAX_WALLS || p->accel < 0.f || p->max_speed < 0.f || p->u_noise < 0.f)
        return fail(FG_ERR_BAD_ARG, "params: 0 <= num_walls <= 4, accel/max_speed/u_noise >= 0%s");
    return FG_OK;
}

}  // namespace fg

using namespace fg;

extern "C" {

int fg_abi_version(void) { return FG_ABI_VERSION; }

const char* fg_last_error(void) { return g_err; }

int fg_kernel_config(int N, int* threads, int* envs_per_wg, int* lds_bytes) {
    Geometry g;
    if (!geometry_for(N, &g)) return fail(FG_ERR_UNSUPPORTED_N, "N must be in [2, 1024]%s");
    if (threads) *threads = g.T;
    if (envs_per_wg) *envs_per_wg = g.E;
    if (lds_bytes) *lds_bytes = g.lds;
    return FG_OK;
}

int64_t fg_step_hd_bytes(int N) { return 24LL * N * N + 53LL * N + 16LL; }

int fg_step_hd(const FgParams* params, int B, int N,
               float* pos_x, float* pos_y, float* vel_x, float* vel_y,
               const float* act, float* ideal_shape, float* ideal_vel, int32_t* step,
               float* obs, float* reward, float* indiv_reward, uint8_t* done,
               int32_t* near_lm, int32_t* near_ag, int32_t* hd_idx, void* stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (B <= 0) return fail(FG_ERR_BAD_ARG, "B must be > 0%s");
    if (N < 3 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "formation_hd_env needs 3 <= N <= 1024%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !act || !ideal_shape || !ideal_vel || !step || !obs || !reward)
        return fail(FG_ERR_BAD_ARG, "fg_step_hd: a required pointer is NULL%s");
    if (((uintptr_t)obs & 15u) || ((uintptr_t)act & 7u) || ((uintptr_t)ideal_shape & 7u) || ((uintptr_t)ideal_vel & 7u))
        return fail(FG_ERR_ALIGNMENT, "obs must be 16-byte, act/ideal_shape/ideal_vel 8-byte aligned%s");
    Args a; memset(&a, 0, sizeof(a));
    a.p = *params; a.B = B; a.N = N; a.K = 1; a.obs_every = 1; a.do_phys = 1; a.do_post = 1;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y; a.act = act;
    a.shape = ideal_shape; a.ivel = ideal_vel; a.step = step;
    a.obs = obs; a.rew = reward; a.indiv = indiv_reward; a.done = done;
    a.near_lm = near_lm; a.near_ag = near_ag; a.hd_idx = hd_idx;
    {   // 81 / 243 agents: pipeline over env batches inside the launch (no index outputs, no World options)
        const char* nopipe = getenv("FG_NOPIPE");
        const bool opts = a.p.num_walls > 0 || a.p.u_noise > 0.f || a.p.max_speed > 0.f || a.p.accel > 0.f;
        if ((N == 81 || N == 243) && !(nopipe && atoi(nopipe)) && !opts && !near_lm && !near_ag && !hd_idx)
            return launch_wide(a, (hipStream_t)stream);
    }
    return launch_step(a, (hipStream_t)stream);
}

int fg_physics_step(const FgParams* params, int B, int N,
                    float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                    const float* act, void* stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (B <= 0) return fail(FG_ERR_BAD_ARG, "B must be > 0%s");
    if (N < 2 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "N must be in [2, 1024]%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !act)
        return fail(FG_ERR_BAD_ARG, "fg_physics_step: a required pointer is NULL%s");
    if ((uintptr_t)act & 7u) return fail(FG_ERR_ALIGNMENT, "act must be 8-byte aligned%s");
    Args a; memset(&a, 0, sizeof(a));
    a.p = *params; a.p.auto_reset = 0; a.B = B; a.N = N; a.K = 1; a.obs_every = 1; a.do_phys = 1; a.do_post = 0;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y; a.act = act;
    return launch_step(a, (hipStream_t)stream);
}

int fg_observe_hd(const FgParams* params, int B, int N,
                  const float* pos_x, const float* pos_y, const float* vel_x, const float* vel_y,
                  const float* ideal_shape, const float* ideal_vel, const int32_t* step,
                  float* obs, float* reward, float* indiv_reward, uint8_t* done,
                  int32_t* near_lm, int32_t* near_ag, int32_t* hd_idx, void* stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (B <= 0) return fail(FG_ERR_BAD_ARG, "B must be > 0%s");
    if (N < 3 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "formation_hd_env needs 3 <= N <= 1024%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !ideal_shape || !ideal_vel)
        return fail(FG_ERR_BAD_ARG, "fg_observe_hd: a required pointer is NULL%s");
    if (!obs && !reward) return fail(FG_ERR_BAD_ARG, "fg_observe_hd: obs and reward both NULL%s");
    if (((uintptr_t)obs & 15u) || ((uintptr_t)ideal_shape & 7u) || ((uintptr_t)ideal_vel & 7u))
        return fail(FG_ERR_ALIGNMENT, "obs must be 16-byte, ideal_shape/ideal_vel 8-byte aligned%s");
    Args a; memset(&a, 0, sizeof(a));
    a.p = *params; a.p.auto_reset = 0; a.B = B; a.N = N; a.K = 1; a.obs_every = 1; a.do_phys = 0; a.do_post = 1;
    a.px = const_cast<float*>(pos_x); a.py = const_cast<float*>(pos_y);
    a.vx = const_cast<float*>(vel_x); a.vy = const_cast<float*>(vel_y);
    a.shape = const_cast<float*>(ideal_shape); a.ivel = const_cast<float*>(ideal_vel);
    a.step = const_cast<int32_t*>(step);
    a.obs = obs; a.rew = reward; a.indiv = indiv_reward; a.done = done;
    a.near_lm = near_lm; a.near_ag = near_ag; a.hd_idx = hd_idx;
    return launch_step(a, (hipStream_t)stream);
}

int fg_rollout_hd(const FgParams* params, int B, int N, int K,
                  float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                  const float* act_seq, float* ideal_shape, float* ideal_vel, int32_t* step,
                  float* obs_seq, float* reward_seq, float* indiv_seq, uint8_t* done_seq,
                  int obs_every, void* stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (B <= 0 || K <= 0) return fail(FG_ERR_BAD_ARG, "B and K must be > 0%s");
    if (N < 3 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "formation_hd_env needs 3 <= N <= 1024%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !act_seq || !ideal_shape || !ideal_vel || !step || !reward_seq)
        return fail(FG_ERR_BAD_ARG, "fg_rollout_hd: a required pointer is NULL%s");
    if (((uintptr_t)obs_seq & 15u) || ((uintptr_t)act_seq & 7u) || ((uintptr_t)ideal_shape & 7u) || ((uintptr_t)ideal_vel & 7u))
        return fail(FG_ERR_ALIGNMENT, "obs_seq must be 16-byte, act_seq/ideal_shape/ideal_vel 8-byte aligned%s");
    Args a; memset(&a, 0, sizeof(a));
    a.p = *params; a.B = B; a.N = N; a.K = K; a.obs_every = obs_every < 1 ? 1 : obs_every;
    a.do_phys = 1; a.do_post = 1;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y; a.act = act_seq;
    a.shape = ideal_shape; a.ivel = ideal_vel; a.step = step;
    a.obs = obs_seq; a.rew = reward_seq; a.indiv = indiv_seq; a.done = done_seq;
    // K >= 2 at the specialised small N: producer / writer pipelined kernel
    if (FG_PROBES) { if (const char* e = getenv("FG_PROBE")) a.probe = atoi(e); }
    const char* nopipe = getenv("FG_NOPIPE");
    if (K >= 2 && !(nopipe && atoi(nopipe)) && (N == 81 || N == 243)) return launch_wide(a, (hipStream_t)stream);
    if (K >= 2 && !(nopipe && atoi(nopipe)) && (N == 27 || N == 9 || N == 3)) {
        int tw = 256;                      // defaults from the MI355X sweep (profiles/README.md)
        if (const char* e = getenv("FG_TW")) tw = atoi(e);
        hipStream_t st = (hipStream_t)stream;
        hipError_t err = hipSuccess;
        int wr = 10;
        if (const char* e = getenv("FG_ROLLWR")) wr = atoi(e);
#define FG_ROLL(NCV, GV, TPV, TWV, EV, WRV)                                                              \
        {   const int grid = (B + (EV) - 1) / (EV);                                                      \
            int lds = (EV) * roll_block_floats(NCV) * (int)sizeof(float);                                \
            if ((WRV) > 0) lds += 2 * ((TWV) / 64) * ((3 * (NCV) * ((WRV) - 1) + 3) & ~1) * (int)sizeof(float2); \
            hipLaunchKernelGGL((rollout_kernel<NCV, GV, TPV, TWV, EV, WRV>), dim3(grid), dim3((TPV) + (TWV)), lds, st, a); \
            err = hipGetLastError(); }
        if (N == 27) {
            int re = 16;                   // 16 envs per workgroup: 8 producer + 4 writer waves, one workgroup per CU
            if (const char* e = getenv("FG_ROLLE")) re = atoi(e);
            if (re == 2) { if (wr == 10) { if (tw == 64) FG_ROLL(27, 32, 64, 64, 2, 10) else FG_ROLL(27, 32, 64, 128, 2, 10) }
                           else { if (tw == 64) FG_ROLL(27, 32, 64, 64, 2, 0) else FG_ROLL(27, 32, 64, 128, 2, 0) } }
            else if (re == 16) { if (wr == 10) { if (tw == 256) FG_ROLL(27, 32, 512, 256, 16, 10) else FG_ROLL(27, 32, 512, 512, 16, 10) }
                                 else { if (tw == 256) FG_ROLL(27, 32, 512, 256, 16, 0) else FG_ROLL(27, 32, 512, 512, 16, 0) } }
            else if (re == 8 && tw == 512) { if (wr == 10) FG_ROLL(27, 32, 256, 512, 8, 10) else FG_ROLL(27, 32, 256, 512, 8, 0) }
            else if (re == 8) { if (wr == 10) { if (tw == 128) FG_ROLL(27, 32, 256, 128, 8, 10) else FG_ROLL(27, 32, 256, 256, 8, 10) }
                                else { if (tw == 128) FG_ROLL(27, 32, 256, 128, 8, 0) else FG_ROLL(27, 32, 256, 256, 8, 0) } }
            else
            if (wr == 10) { if (tw == 64) FG_ROLL(27, 32, 128, 64, 4, 10) else if (tw == 256) FG_ROLL(27, 32, 128, 256, 4, 10) else FG_ROLL(27, 32, 128, 128, 4, 10) }
            else { if (tw == 64) FG_ROLL(27, 32, 128, 64, 4, 0) else if (tw == 256) FG_ROLL(27, 32, 128, 256, 4, 0) else FG_ROLL(27, 32, 128, 128, 4, 0) }
        }
        else if (N == 9) FG_ROLL(9, 16, 64, 64, 4, 0)
        else FG_ROLL(3, 4, 64, 64, 16, 0)
#undef FG_ROLL
        if (err != hipSuccess) return fail(FG_ERR_HIP, "rollout launch failed: %s", hipGetErrorString(err));
        return FG_OK;
    }
    return launch_step(a, (hipStream_t)stream);
}

int fg_reset_hd(const FgParams* params, int B, int N, const uint8_t* mask,
                float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                float* ideal_shape, float* ideal_vel, int32_t* step, void* stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (B <= 0) return fail(FG_ERR_BAD_ARG, "B must be > 0%s");
    Geometry g;
    if (!geometry_for(N, &g)) return fail(FG_ERR_UNSUPPORTED_N, "N must be in [2, 1024]%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !ideal_shape || !ideal_vel)
        return fail(FG_ERR_BAD_ARG, "fg_reset_hd: a required pointer is NULL%s");
    Args a; memset(&a, 0, sizeof(a));
    a.p = *params; a.B = B; a.N = N;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y;
    a.shape = ideal_shape; a.ivel = ideal_vel; a.step = step;
    hipStream_t st = (hipStream_t)stream;
    const int G = g.G;
    int grid;
#define FG_RESET(GG, TT) grid = (B + (TT / GG) - 1) / (TT / GG); \
    hipLaunchKernelGGL((reset_kernel<GG, TT>), dim3(grid), dim3(TT), 0, st, a, mask)
    if (G == 4) { FG_RESET(4, 64); } else if (G == 8) { FG_RESET(8, 64); }
    else if (G == 16) { FG_RESET(16, 64); } else if (G == 32) { FG_RESET(32, 64); }
    else if (G == 64) { FG_RESET(64, 64); } else if (G == 128) { FG_RESET(128, 128); }
    else if (G == 256) { FG_RESET(256, 256); } else if (G == 512) { FG_RESET(512, 512); }
    else { FG_RESET(1024, 1024); }
#undef FG_RESET
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "reset launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

static int launch_scenario(const FgParams* params, const FgScenario* sc, int B, int N, int do_physics,
                           float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                           const float* act, const float* landmarks, float* obst_pos, float* obst_vel,
                           int32_t* step, float* obs, float* reward, float* indiv_reward, uint8_t* done,
                           int32_t* near_ag, void* stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (!sc) return fail(FG_ERR_BAD_ARG, "scenario descriptor is NULL%s");
    const int L = sc->num_landmarks, M = sc->num_obstacles;
    if (sc->kind < FG_SCN_BASIC || sc->kind > FG_SCN_OBSTACLE) return fail(FG_ERR_BAD_ARG, "unknown scenario kind%s");
    if (B <= 0 || L <= 0 || M < 0) return fail(FG_ERR_BAD_ARG, "B and L must be > 0, M >= 0%s");
    if (N < 2 || N + M > 64 || L > 1024) return fail(FG_ERR_UNSUPPORTED_N, "scenario kernel needs 2 <= N, N + M <= 64%s");
    if (sc->kind == FG_SCN_PARTIAL && (sc->num_obs < 0 || sc->num_obs > 1024)) return fail(FG_ERR_BAD_ARG, "bad num_obs%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !landmarks || !obs || (do_physics && (!act || !reward)) ||
        (M > 0 && (!obst_pos || !obst_vel)))
        return fail(FG_ERR_BAD_ARG, "scenario step: a required pointer is NULL%s");
    if (((uintptr_t)obs & 7u) || ((uintptr_t)landmarks & 7u) || (act && ((uintptr_t)act & 7u)) ||
        ((uintptr_t)obst_pos & 7u) || ((uintptr_t)obst_vel & 7u))
        return fail(FG_ERR_ALIGNMENT, "obs/landmarks/act/obstacle buffers must be 8-byte aligned%s");
    ScnArgs a; memset(&a, 0, sizeof(a));
    a.p = *params; a.sc = *sc; a.B = B; a.N = N; a.do_phys = do_physics ? 1 : 0;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y; a.act = act; a.lm = landmarks;
    a.opos = obst_pos; a.ovel = obst_vel; a.step = step;
    a.obs = obs; a.rew = reward; a.indiv = indiv_reward; a.done = done; a.near_ag = near_ag;
    const int G = pow2ceil(N + M) < 4 ? 4 : pow2ceil(N + M);
    const int E = 64 / G;
    const int grid = (B + E - 1) / E;
    const int lds = E * (2 * (N + M) + L) * (int)sizeof(float2);
    hipStream_t st = (hipStream_t)stream;
    if (G == 4) hipLaunchKernelGGL((scn_kernel<4, 64>), dim3(grid), dim3(64), lds, st, a);
    else if (G == 8) hipLaunchKernelGGL((scn_kernel<8, 64>), dim3(grid), dim3(64), lds, st, a);
    else if (G == 16) hipLaunchKernelGGL((scn_kernel<16, 64>), dim3(grid), dim3(64), lds, st, a);
    else if (G == 32) hipLaunchKernelGGL((scn_kernel<32, 64>), dim3(grid), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((scn_kernel<64, 64>), dim3(grid), dim3(64), lds, st, a);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "scenario launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

int fg_reset_hd_mt(int B, int N, const uint8_t* mask, uint32_t* mt_state,
                   float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                   float* ideal_shape, float* ideal_vel, float* landmark_pos, int32_t* step, void* stream) {
    if (B <= 0) return fail(FG_ERR_BAD_ARG, "B must be > 0%s");
    if (N < 2 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "N must be in [2, 1024]%s");
    if (!mt_state || !pos_x || !pos_y || !vel_x || !vel_y || !ideal_shape || !ideal_vel)
        return fail(FG_ERR_BAD_ARG, "fg_reset_hd_mt: a required pointer is NULL%s");
    const int lds = (624 + ((8 * N + 4 + 1) & ~1)) * (int)sizeof(uint32_t) + 2 * (int)sizeof(double);
    hipLaunchKernelGGL(mt_reset_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, B, N, mask, mt_state,
                       pos_x, pos_y, vel_x, vel_y, ideal_shape, ideal_vel, landmark_pos, step);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "mt reset launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

int fg_step_scenario(const FgParams* params, const FgScenario* scenario, int B, int N, int do_physics,
                     float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                     const float* act, const float* landmarks, float* obst_pos, float* obst_vel,
                     int32_t* step, float* obs, float* reward, float* indiv_reward, uint8_t* done,
                     void* stream) {
    return launch_scenario(params, scenario, B, N, do_physics, pos_x, pos_y, vel_x, vel_y, act, landmarks,
                           obst_pos, obst_vel, step, obs, reward, indiv_reward, done, nullptr, stream);
}

int fg_step_basic(const FgParams* params, int B, int N, int L, int do_physics,
                  float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                  const float* act, const float* landmarks, int32_t* step,
                  float* obs, float* reward, float* indiv_reward, uint8_t* done,
                  int32_t* near_ag, void* stream) {
    if (N > 64) return fail(FG_ERR_UNSUPPORTED_N, "basic_formation_env kernel needs 2 <= N <= 64%s");
    FgScenario sc; memset(&sc, 0, sizeof(sc));
    sc.kind = FG_SCN_BASIC; sc.num_landmarks = L; sc.penalty = 1.0f;
    return launch_scenario(params, &sc, B, N, do_physics, pos_x, pos_y, vel_x, vel_y, act, landmarks,
                           nullptr, nullptr, step, obs, reward, indiv_reward, done, near_ag, stream);
}

}  // extern "C"
