// fg_hd_lane_kernel.hpp - K-step rollouts of formation_hd_env at 3 and 4 agents: ONE ENVIRONMENT PER LANE.
// Part of libformation_hip (gfx950); included by formation_hip.hip, one translation unit.
//
// Three agents are what `make_env` and test.py default to (reference __init__.py:6, test.py:9-10: n = 3, one layer).  With a
// lane per agent (fg::rollout_kernel<3, 4, ...>) an env's observation rows are 72 bytes and a store instruction scatters
// sixteen 32-byte pieces: 3 x 65536 envs ran at 0.50 of the HBM rate.  Here - as for the landmark scenarios
// (fg_scn_lane_kernel.hpp, whose writer wave this kernel shares) - a lane owns a whole environment in registers, the pair
// loops are unrolled, and the 64 envs of a wave leave as one contiguous span through LDS.
//
// The arithmetic is that of fg::rollout_kernel / fg::step_kernel, operation for operation (contact_force_packed's marked-
// then-flushed pairs in ascending j, the lane-group butterflies' association order, the same Philox draws), so a launch of
// this kernel equals K single-step launches bit for bit: tests/test_gpu_launch_paths.py, tests/test_gpu_fuzz_rollout.py.
// Reference lines: environment.py:113-142, core.py:206-322, envs/formation_hd_env.py:38-95.
#ifndef FG_HD_LANE_KERNEL_HPP_
#define FG_HD_LANE_KERNEL_HPP_

#include "fg_common.hpp"
#include "fg_scn_lane_kernel.hpp"
#include "fg_policy_kernels.hpp"

namespace fg {

// (pw producer waves per workgroup: the block is the image of 64 pw envs, see lane_writer_wave)
__host__ __device__ constexpr int hd_lane_block_bytes(int n, int pw = 1) { return pw * (64 * scn_lane_pitch(3 * n * n) * 8 + 3 * 64 * n * 4); }
__host__ __device__ constexpr bool hd_lane_double(int n) { return 2 * hd_lane_block_bytes(n) <= 40 * 1024; }
__host__ __device__ constexpr int hd_lane_lds_bytes(int n, int pw = 1) { return (hd_lane_double(n) ? 2 : 1) * hd_lane_block_bytes(n, pw); }

// PER > 0: closed loop - the action of step k is the demo controller (PER-ary hierarchy, N = PER^L) on the state step k-1 left,
// evaluated by the env's lane on its registers (bfs_policy_lane); a.act is not read, a.act_out records the actions.  The
// lane-per-agent closed loop ran 4 x 65536 at 8.6 us/step against the open loop's 4.95 (0.44 of the HBM rate in real bytes).
template <int N, int PER = 0, int PW = 1>
__global__ __launch_bounds__(128 * PW) void hd_lane_kernel(const Args a) {
    constexpr bool POLICY = PER > 0;
    static_assert(N >= 3 && N <= 4, "one env per lane: the LDS block of 64 envs must leave room for four producer waves per CU");
    constexpr int G = 4;                                // the lane group of step_kernel / rollout_kernel at 3 and 4 agents
    constexpr int D = 6 * N, U = 3 * N * N, SU = scn_lane_pitch(U);
    constexpr bool DB = hd_lane_double(N);
    constexpr int BLOCK_UNITS = hd_lane_block_bytes(N, PW) / 8;
    constexpr int ENVS = 64 * PW;
    extern __shared__ __attribute__((aligned(16))) float2 smem_all[];
    const int lane = threadIdx.x & 63;
    const int per_xcd = (int)(gridDim.x >> 3);          // XCD-aware workgroup order, as in scn_lane_kernel
    const int wg = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int b0 = wg * ENVS;
    if (b0 >= a.B) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slot = wave * 64 + lane;                  // the lane's env inside the workgroup (producer waves)
    const int b = b0 + slot;
    const bool live = b < a.B;
    const int bl = live ? b : a.B - 1;
    const int El = min(ENVS, a.B - b0);
    const int KS = a.K;
    if (wave >= PW) {
        lane_writer_wave<N, D, DB, PW, PW>(smem_all, KS, a.B, b0, El, a.obs_every, a.obs, a.rew, a.indiv, a.done, lane, wave - PW);
        return;
    }
    // ---- PRODUCER wave: lane = env ----
    float2 p[N], v[N], s[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const size_t o = (size_t)bl * N + i;
        p[i] = make_float2(a.px[o], a.py[o]);
        v[i] = make_float2(a.vx[o], a.vy[o]);
        s[i] = reinterpret_cast<const float2*>(a.shape)[o];
    }
    float2 iv = reinterpret_cast<const float2*>(a.ivel)[bl];
    int t_step = a.step ? a.step[bl] : 0;
    bool fresh = false;                                 // an in-launch reset re-drew ideal_shape / ideal_vel: written back at the end

    const float one_minus_damp = 1.0f - a.p.damping;
    const float dt = a.p.dt;
    const float cutoff = a.p.dist_min + 18.0f * a.p.contact_margin;
    const float cutoff2 = cutoff * cutoff;
    const float thr2 = (float)((double)a.p.collide_thresh * (double)a.p.collide_thresh);
    const float invN = 1.0f / (float)N;
    const float kmargin = a.p.contact_margin, inv_k = 1.0f / kmargin, cf = a.p.contact_force, dmin = a.p.dist_min;
    const uint64_t rbase = rng_base(a.p);               // read once: no load from the device counter inside the step loop

    float2 u_next[N];
#pragma unroll
    for (int i = 0; i < N; ++i) u_next[i] = make_float2(0.f, 0.f);
    if constexpr (!POLICY) {
#pragma unroll
        for (int i = 0; i < N; ++i) u_next[i] = reinterpret_cast<const float2*>(a.act)[(size_t)bl * N + i];
    }

    for (int ks = 0; ks < KS; ++ks) {
        const size_t kb = (size_t)ks * a.B;
        float2 u_now[N];
        if constexpr (POLICY) {
            // get_action_BFS(ezpolicy, obs, PER) on the observation the previous step (or the reset) returned: its row 0 holds
            // p_j - p_0, exactly this subtraction
            float2 rel[N];
#pragma unroll
            for (int i = 0; i < N; ++i) rel[i] = make_float2(p[i].x - p[0].x, p[i].y - p[0].y);
            bfs_policy_lane<N, (PER > 0 ? PER : N)>(rel, s, a.pl, iv, u_now);
            if (a.act_out && live) {
#pragma unroll
                for (int i = 0; i < N; ++i) reinterpret_cast<float2*>(a.act_out)[(kb + b) * N + i] = u_now[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) u_now[i] = u_next[i];
            if (ks + 1 < KS) {
#pragma unroll
                for (int i = 0; i < N; ++i) u_next[i] = reinterpret_cast<const float2*>(a.act)[(kb + a.B + bl) * N + i];
            }
        }
        // ---- World.step: every pair once in lexicographic order = ascending j for each agent (contact_force_packed flushes
        // its marked partners in that order); the two forces of a pair are exact negatives
        float fx[N], fy[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { fx[i] = 0.f; fy[i] = 0.f; }
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = i + 1; j < N; ++j) {
                const float dx = p[i].x - p[j].x, dy = p[i].y - p[j].y;
                const float d2 = dx * dx + dy * dy;
                if (d2 < cutoff2) {
                    const float d = hw_sqrt(d2);
                    const float x = (dmin - d) * inv_k;
                    const float pen = kmargin * (rmax(x, 0.0f) + hw_log(1.0f + hw_exp(-rabs(x))));
                    const float c = cf * pen * hw_rcp(d);
                    fx[i] += dx * c; fy[i] += dy * c;
                    const float ex = -dx, ey = -dy;
                    fx[j] += ex * c; fy[j] += ey * c;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            fx[i] += a.p.mass * (a.p.sensitivity * u_now[i].x);
            fy[i] += a.p.mass * (a.p.sensitivity * u_now[i].y);
            v[i].x = v[i].x * one_minus_damp + (fx[i] / a.p.mass) * dt;
            v[i].y = v[i].y * one_minus_damp + (fy[i] / a.p.mass) * dt;
            p[i].x += v[i].x * dt;
            p[i].y += v[i].y * dt;
        }
        t_step += 1;
        // ---- Scenario.reward: centroid, mean velocity (lane-group butterfly order), Hausdorff minima, collision counts ----
        float gx[G], gy[G], hx[G], hy[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            gx[g] = g < N ? p[g < N ? g : 0].x : 0.f; gy[g] = g < N ? p[g < N ? g : 0].y : 0.f;
            hx[g] = g < N ? v[g < N ? g : 0].x : 0.f; hy[g] = g < N ? v[g < N ? g : 0].y : 0.f;
        }
        const float mx = lane_group_sum<G>(gx) * invN, my = lane_group_sum<G>(gy) * invN;
        const float mvx = lane_group_sum<G>(hx) * invN, mvy = lane_group_sum<G>(hy) * invN;
        float rowmax = -INFINITY, colmax = -INFINITY;
        int cnt[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float ptx = p[i].x - mx, pty = p[i].y - my;      // centred own position
            const float tx = s[i].x + mx, ty = s[i].y + my;        // own ideal point, un-centred
            float rowmin = INFINITY, colmin = INFINITY;
            int c = -1;                                            // the self pair (distance 0) is counted below
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float cx = p[j].x - p[i].x, cy = p[j].y - p[i].y;
                c += (cx * cx + cy * cy < thr2) ? 1 : 0;
                const float rx = ptx - s[j].x, ry = pty - s[j].y;
                rowmin = fminf(rowmin, rx * rx + ry * ry);
                const float ux = p[j].x - tx, uy = p[j].y - ty;
                colmin = fminf(colmin, ux * ux + uy * uy);
            }
            cnt[i] = c + (thr2 > 0.0f ? 0 : 1);
            rowmax = fmaxf(rowmax, rowmin); colmax = fmaxf(colmax, colmin);
        }
        float cs[G];
#pragma unroll
        for (int g = 0; g < G; ++g) cs[g] = g < N ? (float)cnt[g < N ? g : 0] : 0.f;
        const float ctot = lane_group_sum<G>(cs);
        const float H = sqrtf(fmaxf(rowmax, colmax));
        const float ex = iv.x - mvx, ey = iv.y - mvy;
        const float velterm = sqrtf(ex * ex + ey * ey);
        const bool is_done = t_step >= a.p.world_length;
        const float shared = (float)(-(double)N * ((double)H + (double)velterm) - (double)ctot);
        float indiv[N];
#pragma unroll
        for (int i = 0; i < N; ++i) indiv[i] = (-H - velterm) - (float)cnt[i];
        const uint32_t done_flag = is_done ? 1u : 0u;
        if (a.p.auto_reset && is_done) {                            // fg_reset_hd's draws (formation_hd_env.py:77-95)
            const uint64_t off = rbase + (uint64_t)ks;
            float rxs[G], rys[G], rx[N], ry[N];
#pragma unroll
            for (int g = 0; g < G; ++g) { rxs[g] = 0.f; rys[g] = 0.f; }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                uint32_t c[4] = {(uint32_t)(b + a.p.env_index_base), (uint32_t)i, (uint32_t)off, (uint32_t)(off >> 32)};
                philox4x32(c, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
                p[i] = make_float2(u_pm1(c[0]), u_pm1(c[1]));
                v[i] = make_float2(0.f, 0.f);
                rx[i] = u_pm1(c[2]); ry[i] = u_pm1(c[3]);
                rxs[i] = rx[i]; rys[i] = ry[i];
            }
            const float sx = lane_group_sum<G>(rxs), sy = lane_group_sum<G>(rys);
#pragma unroll
            for (int i = 0; i < N; ++i) s[i] = make_float2(__builtin_fmaf(-sx, invN, rx[i]), __builtin_fmaf(-sy, invN, ry[i]));
            uint32_t c2[4] = {(uint32_t)(b + a.p.env_index_base), 0xFFFFFFFFu, (uint32_t)off, (uint32_t)(off >> 32)};
            philox4x32(c2, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
            iv = make_float2(u_pm1(c2[0]), u_pm1(c2[1]));
            t_step = 0;
            fresh = true;
        }
        // ---- hand-over: the lane's [N][6N] observation block and its rewards into LDS, for the writer wave ----
        const bool want_obs = a.obs_every <= 1 || (ks + 1) % a.obs_every == 0;
        if (!DB) __syncthreads();                       // A (one block): the writer has read the block of step ks - 1
        float2* const smem = smem_all + (DB ? (ks & 1) * BLOCK_UNITS : 0);
        float* const s_rew = reinterpret_cast<float*>(smem + ENVS * SU);
        float* const s_ind = s_rew + ENVS * N;
        uint32_t* const s_done = reinterpret_cast<uint32_t*>(s_ind + ENVS * N);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            s_rew[slot * N + i] = shared; s_ind[slot * N + i] = indiv[i]; s_done[slot * N + i] = done_flag;
        }
        if (want_obs) {
            float2* const mine = smem + slot * SU;
#pragma unroll
            for (int i = 0; i < N; ++i) {                // row i: [v_i | p_j - p_i (j != i) | zeros | ideal_shape | ideal_vel]
                float2* const o = mine + i * (D / 2);
                int w = 0;
                o[w++] = v[i];
#pragma unroll
                for (int j = 0; j < N; ++j)
                    if (j != i) o[w++] = make_float2(p[j].x - p[i].x, p[j].y - p[i].y);
#pragma unroll
                for (int j = 0; j < N - 1; ++j) o[w++] = make_float2(0.f, 0.f);
#pragma unroll
                for (int j = 0; j < N; ++j) o[w++] = s[j];
                o[w++] = iv;
            }
        }
        __syncthreads();                                // B: published
    }
    if (live) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const size_t o = (size_t)b * N + i;
            a.px[o] = p[i].x; a.py[o] = p[i].y; a.vx[o] = v[i].x; a.vy[o] = v[i].y;
            if (fresh) reinterpret_cast<float2*>(a.shape)[o] = s[i];
        }
        if (fresh) reinterpret_cast<float2*>(a.ivel)[b] = iv;
        if (a.step) a.step[b] = t_step;
    }
}

}  // namespace fg

#endif  // FG_HD_LANE_KERNEL_HPP_
