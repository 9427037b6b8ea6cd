// fg_scn_lane_kernel.hpp - The landmark scenarios at their reference shapes: ONE ENVIRONMENT PER LANE, every count a
// compile-time constant.  Part of libformation_hip (gfx950); included by formation_hip.hip, one translation unit.
//
// fg::scn_kernel (fg_aux_kernels.hpp) gives one lane to every movable entity and walks run-time loops over 3-7 entities:
// at 3 agents 3 of 4 lanes work, the loops cost as many scalar and branch instructions as vector ones, and the kernel
// runs at 0.3-0.4 of the HBM rate (profiles/r03_scn_pmc.txt).  Here a lane owns a whole environment: agents, obstacles
// and landmarks live in its registers, every pair loop is unrolled, nothing is reduced across lanes, and the 64
// environments of a wave form ONE contiguous span of the [B][N][D] observation tensor - composed row by row in LDS
// (conflict-free: odd row pitch) and streamed out with lane-consecutive 16-byte stores.
// A workgroup is PW PRODUCER waves (lane = env: 64 PW environments, one span of the tensor) and as many WRITER waves (twice
// as many for basic_formation_env): the producers run World.step + reward of step k+1 while the writers stream step k's
// observations, rewards and done flags from LDS to global memory; one or two workgroup barriers per step hand the LDS
// block(s) back and forth.  PW = 1 everywhere but for large batches into HBM-size buffers, where four producer waves per
// workgroup - a 256-env span per workgroup and step - run 7-10 % faster (formation_hip.hip: scn_lane_wide).  The split also keeps the two kinds of memory traffic on
// different waves: on gfx9 loads and stores share one counter (vmcnt), so a wave that both prefetches its next actions
// and stores its outputs has to drain its whole store stream before it can use the prefetched action - the one-wave form
// of this kernel spent a third of every step doing that (basic 3 x 65536: 3.99 us/step against the two-wave form's figure
// in profiles/r04_scenario_rollout.md).
//
// The arithmetic is scn_kernel's, operation for operation (same helper functions, same expressions, sums in the
// association order of its lane-group butterflies), so the two kernels agree bit for bit: tests/test_gpu_scenario_lane.py.
// Reference lines under /root/reference/formation_gym/envs/:
//   basic     observation basic_formation_env.py:29-41, reward :43-52 (self "collision" included)
//   partial   observation formation_hd_partial_env.py:38-57 (ring neighbours), reward :59-72
//   range     observation formation_hd_partial_range_env.py:38-52 (clipped), reward as partial
//   obstacle  observation formation_hd_obs_env.py:44-58, reward :60-99 incl. the obstacle velocity override (:84-89)
//   World.step core.py:206-277, 289-322 (all pairs of movable colliders)
#ifndef FG_SCN_LANE_KERNEL_HPP_
#define FG_SCN_LANE_KERNEL_HPP_

#include "fg_common.hpp"
#include "fg_aux_kernels.hpp"

namespace fg {

// sum of G values in the association order of env_reduce's butterfly over an aligned group of G lanes:
// a balanced tree over the slots in natural order
template <int G> FG_DEV float lane_group_sum(const float (&x)[G]) {
    float t[G];
#pragma unroll
    for (int g = 0; g < G; ++g) t[g] = x[g];
#pragma unroll
    for (int s = 1; s < G; s <<= 1) {
#pragma unroll
        for (int g = 0; g < G; g += 2 * s) t[g] = t[g] + t[g + s];
    }
    return t[0];
}

__host__ __device__ constexpr int scn_group_lanes(int entities) {     // scn_kernel's G: pow2 >= N + M, at least 4
    int g = 4;
    while (g < entities) g <<= 1;
    return g;
}
__host__ __device__ constexpr int scn_obs_dim(int kind, int n, int l, int m, int nbr) {
    return 2 + (kind == FG_SCN_BASIC ? 2 : 0) + 2 * l + 2 * m + 2 * nbr + 2 * (n - 1);
}
// LDS pitch (float2 units) of one env's [N][D] block: odd, so that the lanes' ds_write_b64 fall on distinct banks
__host__ __device__ constexpr int scn_lane_pitch(int units) { return units | 1; }

// NBR: neighbours observed = num_obs (partial) or N - 1 (the other kinds)
// LDS of one hand-over block: the observation block [64][SU] (float2), then reward / individual reward / done of the
// 64 x N agents as three arrays of 64 N dwords
__host__ __device__ constexpr int scn_lane_block_bytes(int kind, int n, int l, int m, int nbr, int pw = 1) {
    return pw * (64 * scn_lane_pitch(n * scn_obs_dim(kind, n, l, m, nbr) / 2) * 8 + 3 * 64 * n * 4);
}
// Two blocks (the producer composes step k+1 while the writer streams step k: one barrier per step, nobody waits for the
// compose) where four workgroups per CU still fit the 160 KiB - basic_formation_env's 16 KiB blocks; the larger rows of
// the other scenarios keep ONE block and two barriers per step (two blocks would halve the workgroups per CU).
__host__ __device__ constexpr bool scn_lane_double(int kind, int n, int l, int m, int nbr) {
    return 2 * scn_lane_block_bytes(kind, n, l, m, nbr) <= 40 * 1024;
}
__host__ __device__ constexpr int scn_lane_lds_bytes(int kind, int n, int l, int m, int nbr, int pw = 1) {
    return (scn_lane_double(kind, n, l, m, nbr) ? 2 : 1) * scn_lane_block_bytes(kind, n, l, m, nbr, pw);
}

// The WRITER wave of a one-env-per-lane workgroup (the landmark scenarios here, formation_hd_env in fg_hd_lane_kernel.hpp):
// block ks (published at barrier B) -> global memory, while the producer computes step ks + 1.  A block = the 64 envs'
// [N][D] observation rows at an odd pitch of U | 1 float2 units, then reward / individual reward / done of the 64 x N agents
// as three arrays of 64 N dwords; DB = two blocks taking turns (one barrier per step) instead of one (two barriers).
// NWW writer waves share the block: wave w takes the store instructions w, w + NWW, ... of the span (and of the reward /
// done arrays).  More writer waves do NOT buy store bandwidth here - every CU already has its write queue full (64 outstanding
// 64-byte requests per CU x 330-480 cycles of write latency is the chip's store ceiling, profiles/r04_store_path_pmc.txt,
// r05_place_channels.md): two waves gain 2 % at basic_formation_env's small blocks (3.08 -> 3.02 us/step at 65536 envs) and lose
// 2-6 % at the larger rows of the other scenarios, three lose 2-20 % (profiles/r05_lane_writers_ab.txt).
__host__ __device__ constexpr int scn_lane_writers(int kind) { return kind == FG_SCN_BASIC ? 2 : 1; }
// PW producer waves share the workgroup: the block is then the image of 64 PW envs (env = 64 * producer wave + lane) and the
// workgroup's span 64 PW envs long.
template <int N, int D, bool DB, int NWW = 1, int PW = 1>
FG_DEV void lane_writer_wave(const float2* smem_all, int KS, int B, int b0, int El, int obs_every,
                             float* __restrict__ obs, float* __restrict__ rew, float* __restrict__ indiv, uint8_t* __restrict__ done, int lane,
                             int w = 0) {
    constexpr int U = N * D / 2, SU = scn_lane_pitch(U);
    constexpr int ENVS = 64 * PW;
    constexpr int BLOCK_UNITS = ENVS * SU + (3 * ENVS * N) / 2;
    for (int ks = 0; ks < KS; ++ks) {
        if (!DB) __syncthreads();                   // A: the block of step ks - 1 has been read (nothing to do for ks = 0)
        __syncthreads();                            // B: the block of step ks is complete
        const float2* const smem = smem_all + (DB ? (ks & 1) * BLOCK_UNITS : 0);
        const float* const s_rew = reinterpret_cast<const float*>(smem + ENVS * SU);
        const float* const s_ind = s_rew + ENVS * N;
        const uint32_t* const s_done = reinterpret_cast<const uint32_t*>(s_ind + ENVS * N);
        const size_t kb = (size_t)ks * B;
        const bool want_obs = obs != nullptr && (obs_every <= 1 || (ks + 1) % obs_every == 0);
        if (want_obs) {
            // unit q of the workgroup's span = unit (q mod U) of env (q div U); 64 units per instruction, lanes consecutive
            const size_t ob = (size_t)(obs_every > 1 ? ks / obs_every : ks) * B;
            float2* const out = reinterpret_cast<float2*>(obs + (ob + (size_t)b0) * N * D);
            if (El == ENVS) {
                // a full block: 32 U pairs of units, one 16-byte store per lane and instruction (1 KiB per wave instruction; the
                // two units of a pair may sit in different rows of the LDS image.  8-byte stores: basic 3.2-3.45 -> 3.1 us/step,
                // partial 6.8-7.5 -> 6.5, profiles/r04_lane_x4_ab.txt).  An odd U makes the image contiguous (pitch U | 1 = U):
                // the pair is ONE 16-byte LDS read, lanes consecutive, no bank conflicts (the two 8-byte reads at a 16-byte lane
                // stride met two-way: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.21 in profiles/r04_scn_pmc.txt).
                constexpr int NP2 = ENVS / 2 * U, IT = (NP2 + 63) / 64;
                constexpr int MINE = (IT + NWW - 1) / NWW;          // store instructions of one writer wave
                f32x4* const out4 = reinterpret_cast<f32x4*>(out);
                const f32x4* const img4 = reinterpret_cast<const f32x4*>(smem);
#pragma unroll
                for (int t0 = 0; t0 < MINE; t0 += 8) {
                    f32x4 r[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        if (t0 + c < MINE) {
                            const int pair = (w + NWW * (t0 + c)) * 64 + lane;
                            const bool ok = pair < NP2;
                            if constexpr (SU == U) {
                                r[c] = img4[ok ? pair : 0];
                            } else {
                                const int q = 2 * pair;
                                const int r0 = q / U, c0 = q - r0 * U;
                                int r1 = r0, c1 = c0 + 1;
                                if (c1 == U) { c1 = 0; r1 += 1; }
                                const float2 x0 = smem[ok ? r0 * SU + c0 : 0], x1 = smem[ok ? r1 * SU + c1 : 0];
                                r[c] = (f32x4){x0.x, x0.y, x1.x, x1.y};
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const int pair = (w + NWW * (t0 + c)) * 64 + lane;
                        if (t0 + c < MINE && pair < NP2) out4[pair] = r[c];
                    }
                }
            } else {
                const int units = El * U;
                for (int q = w * 64 + lane; q < units; q += 64 * NWW) {
                    const int row = q / U, col = q - row * U;
                    out[q] = smem[row * SU + col];
                }
            }
        }
        // reward, individual reward, done of the 64 x N agents: [K][B][N], the workgroup's slice is contiguous; the 3 N store
        // instructions are dealt over the writer waves
        const int cnt = El * N;
        const size_t o0 = (kb + b0) * N + lane;
        if (rew) {                                  // (one uniform branch per array, not per store)
#pragma unroll
            for (int c = 0; c < N * PW; ++c) if (c % NWW == w && c * 64 + lane < cnt) rew[o0 + c * 64] = s_rew[c * 64 + lane];
        }
        if (indiv) {
#pragma unroll
            for (int c = 0; c < N * PW; ++c) if ((c + 1) % NWW == w && c * 64 + lane < cnt) indiv[o0 + c * 64] = s_ind[c * 64 + lane];
        }
        if (done) {
#pragma unroll
            for (int c = 0; c < N * PW; ++c) if ((c + 2) % NWW == w && c * 64 + lane < cnt) done[o0 + c * 64] = (uint8_t)s_done[c * 64 + lane];
        }
    }
}

// PW producer waves per workgroup (64 PW envs, one span of the observation tensor), PW * scn_lane_writers(KIND) writer waves
template <int KIND, int N, int L, int M, int NBR, int PW = 1>
__global__ __launch_bounds__(64 * PW * (1 + scn_lane_writers(KIND))) void scn_lane_kernel(const ScnArgs a) {
    constexpr bool DB = scn_lane_double(KIND, N, L, M, NBR);
    constexpr int BLOCK_UNITS = scn_lane_block_bytes(KIND, N, L, M, NBR, PW) / 8;   // float2 units from one block to the next
    constexpr int ENVS = 64 * PW, NWW = PW * scn_lane_writers(KIND);
    constexpr int NE = N + M;
    constexpr int G = scn_group_lanes(NE);
    constexpr int D = scn_obs_dim(KIND, N, L, M, NBR);
    constexpr int U = N * D / 2;                        // float2 units per env
    constexpr int SU = scn_lane_pitch(U);
    constexpr bool BASIC = KIND == FG_SCN_BASIC;
    static_assert(NE <= 8 && L <= 8, "one env per lane: a handful of entities");
    extern __shared__ __attribute__((aligned(16))) float2 smem_all[];
    const int lane = threadIdx.x & 63;
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2: consecutive workgroup ids take consecutive
    // 64-env spans WITHIN an XCD's eighth of the batch, so that the cache lines two neighbouring spans share (done bytes,
    // the ends of an observation span) meet in one L2 instead of leaving two XCDs as partial-line writes.
    // (the host launches 8 x ceil(workgroups / 8) of them; the ones beyond the batch leave at once)
    const int per_xcd = (int)(gridDim.x >> 3);
    const int wg = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int b0 = wg * ENVS;
    if (b0 >= a.B) return;                              // uniform over the workgroup, before any barrier
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slot = wave * 64 + lane;                  // the lane's env inside the workgroup (producer waves)
    const int b = b0 + slot;
    const bool live = b < a.B;
    const int bl = live ? b : a.B - 1;                  // loads of a lane beyond the batch stay in range; it stores nothing
    const int El = min(ENVS, a.B - b0);
    const int KS = a.K > 1 ? a.K : 1;

    if (wave >= PW) {
        lane_writer_wave<N, D, DB, NWW, PW>(smem_all, KS, a.B, b0, El, a.obs_every, a.obs, a.rew, a.indiv, a.done, lane, wave - PW);
        return;
    }

    float2 p[NE], v[NE], lm[L];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const size_t s = (size_t)bl * N + i;
        p[i] = make_float2(a.px[s], a.py[s]);
        v[i] = make_float2(a.vx[s], a.vy[s]);
    }
#pragma unroll
    for (int k = 0; k < M; ++k) {
        p[N + k] = reinterpret_cast<const float2*>(a.opos)[(size_t)bl * M + k];
        v[N + k] = reinterpret_cast<const float2*>(a.ovel)[(size_t)bl * M + k];
    }
#pragma unroll
    for (int l = 0; l < L; ++l) lm[l] = reinterpret_cast<const float2*>(a.lm)[(size_t)bl * L + l];
    int t_step = a.step ? a.step[bl] : 0;
    bool fresh_lm = false;                              // landmarks re-drawn by an in-launch reset: written back at the end

    const float k_margin = a.p.contact_margin;
    const float half_agent = 0.5f * a.p.dist_min, half_obst = 0.5f * (2.0f * a.sc.obstacle_size);
    const float thr = a.p.collide_thresh, thr2 = (float)((double)thr * (double)thr);
    const float ot = 0.5f * a.p.dist_min + a.sc.obstacle_size, ot2 = (float)((double)ot * (double)ot);
    float2 u_next[N];
#pragma unroll
    for (int i = 0; i < N; ++i) u_next[i] = make_float2(0.f, 0.f);
    if (a.do_phys) {
#pragma unroll
        for (int i = 0; i < N; ++i) u_next[i] = reinterpret_cast<const float2*>(a.act)[(size_t)bl * N + i];
    }

    const uint64_t rbase = rng_base(a.p);               // read once: no load from the device counter inside the step loop
    for (int ks = 0; ks < KS; ++ks) {
        const uint64_t off = rbase + (uint64_t)ks;
        const size_t kb = (size_t)ks * a.B;
        float2 u_now[N];
#pragma unroll
        for (int i = 0; i < N; ++i) u_now[i] = u_next[i];
        if (a.do_phys && ks + 1 < KS) {                 // the action of step ks + 1 is fetched while step ks runs
#pragma unroll
            for (int i = 0; i < N; ++i) u_next[i] = reinterpret_cast<const float2*>(a.act)[(kb + a.B + bl) * N + i];
        }
        if (a.do_phys) {
            // ---- World.step: every pair once, in lexicographic order - which is ascending j for each entity, the
            // order core.py:240-262 (and scn_kernel) accumulates in; the pair's two forces are exact negatives
            float fx[NE], fy[NE];
#pragma unroll
            for (int i = 0; i < NE; ++i) { fx[i] = 0.f; fy[i] = 0.f; }
#pragma unroll
            for (int i = 0; i < NE; ++i) {
#pragma unroll
                for (int j = i + 1; j < NE; ++j) {
                    const float dmin = (i < N ? half_agent : half_obst) + (j < N ? half_agent : half_obst);
                    const float cut = dmin + 18.0f * k_margin;
                    const float dx = p[i].x - p[j].x, dy = p[i].y - p[j].y;
                    const float d2 = dx * dx + dy * dy;
                    if (d2 < cut * cut) {
                        const float d = __builtin_amdgcn_sqrtf(d2);
                        const float x = (dmin - d) / k_margin;
                        const float pen = k_margin * (fmaxf(x, 0.0f) + __logf(1.0f + __expf(-fabsf(x))));
                        const float c = a.p.contact_force * pen * __builtin_amdgcn_rcpf(d);
                        fx[i] += dx * c; fy[i] += dy * c;
                        const float ex = -dx, ey = -dy;                 // p_j - p_i, exactly
                        fx[j] += ex * c; fy[j] += ey * c;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const float2 fa = action_force(a.p, agent_props_of(a.p, i, false), u_now[i], (uint32_t)(b + a.p.env_index_base),
                                               (uint32_t)i, off);
                fx[i] += fa.x; fy[i] += fa.y;
            }
            if (a.p.num_walls > 0) {                    // (one uniform branch around all entities)
#pragma unroll
                for (int i = 0; i < NE; ++i) wall_forces(a.p, p[i], i < N ? half_agent : half_obst, fx[i], fy[i]);
            }
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                v[i].x = v[i].x * (1.0f - a.p.damping) + (fx[i] / a.p.mass) * a.p.dt;
                v[i].y = v[i].y * (1.0f - a.p.damping) + (fy[i] / a.p.mass) * a.p.dt;
                if (i < N) v[i] = clamp_speed(a.p.max_speed, v[i]);
                p[i].x += v[i].x * a.p.dt; p[i].y += v[i].y * a.p.dt;
                if (i >= N) {                           // the reward callback re-arms the obstacle velocity every step (:84-89)
                    const bool falling = p[i].y > a.sc.obstacle_floor;
                    v[i] = make_float2(falling ? a.sc.obstacle_vx : 0.f, falling ? a.sc.obstacle_vy : 0.f);
                }
            }
            t_step += 1;
        }
        // ---- formation term ----
        float form;
        if constexpr (BASIC) {
            float slot[G];
#pragma unroll
            for (int g = 0; g < G; ++g) slot[g] = 0.f;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                float best = INFINITY; int barg = 0;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float dx = p[j].x - lm[l].x, dy = p[j].y - lm[l].y, d2 = dx * dx + dy * dy;
                    if (d2 < best) { best = d2; barg = j; }
                }
                slot[l % G] += sqrtf(best);
                if (a.near_ag && live) a.near_ag[(kb + b) * L + l] = barg;
            }
            form = lane_group_sum<G>(slot);
        } else {
            float sx[G], sy[G], tx[G], ty[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                sx[g] = g < N ? p[g < N ? g : 0].x : 0.f; sy[g] = g < N ? p[g < N ? g : 0].y : 0.f;
                tx[g] = 0.f; ty[g] = 0.f;
            }
#pragma unroll
            for (int l = 0; l < L; ++l) { tx[l % G] += lm[l].x; ty[l % G] += lm[l].y; }
            const float mx = lane_group_sum<G>(sx) * a.inv_n, my = lane_group_sum<G>(sy) * a.inv_n;
            const float lx = lane_group_sum<G>(tx) * a.inv_l, ly = lane_group_sum<G>(ty) * a.inv_l;
            float rowmax = -INFINITY, colmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < N; ++i) {                       // min over landmarks for agent i
                float rowmin = INFINITY;
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const float dx = (p[i].x - mx) - (lm[l].x - lx), dy = (p[i].y - my) - (lm[l].y - ly);
                    rowmin = fminf(rowmin, dx * dx + dy * dy);
                }
                rowmax = fmaxf(rowmax, rowmin);
            }
#pragma unroll
            for (int l = 0; l < L; ++l) {                       // min over agents for landmark l
                float cm = INFINITY;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float dx = (p[j].x - mx) - (lm[l].x - lx), dy = (p[j].y - my) - (lm[l].y - ly);
                    cm = fminf(cm, dx * dx + dy * dy);
                }
                colmax = fmaxf(colmax, cm);
            }
            form = sqrtf(fmaxf(rowmax, colmax));
        }
        // ---- collision counts (pairs shared: |p_j - p_i|^2 is the same number from either side) ----
        int cnt[N];
#pragma unroll
        for (int i = 0; i < N; ++i) cnt[i] = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if constexpr (BASIC) {                                      // the self pair (basic_formation_env.py:49-51): distance 0,
                const float dx = p[i].x - p[i].x, dy = p[i].y - p[i].y; //   NaN for a NaN position, as scn_kernel computes it
                cnt[i] += (dx * dx + dy * dy < thr2) ? 1 : 0;
            }
#pragma unroll
            for (int j = i + 1; j < N; ++j) {
                const float dx = p[j].x - p[i].x, dy = p[j].y - p[i].y;
                const int hit = (dx * dx + dy * dy < thr2) ? 1 : 0;
                cnt[i] += hit; cnt[j] += hit;
            }
#pragma unroll
            for (int k = 0; k < M; ++k) {
                const float dx = p[N + k].x - p[i].x, dy = p[N + k].y - p[i].y;
                cnt[i] += (dx * dx + dy * dy < ot2) ? 1 : 0;
            }
        }
        int total = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) total += cnt[i];
        const bool is_done = t_step >= a.p.world_length;
        const float shared = (float)(-(double)N * (double)form - (double)a.sc.penalty * (double)(float)total);
        float indiv[N];
#pragma unroll
        for (int i = 0; i < N; ++i) indiv[i] = -form - a.sc.penalty * (float)cnt[i];
        const uint32_t done_flag = is_done ? 1u : 0u;       // the finished step's flag: the reset below does not change it
        if (a.p.auto_reset && a.do_phys && is_done) {
            // the vec-env worker's rule (env_wrappers.py:14-18): the env restarts at once, the RESET observation goes out
            // with the finished step's reward / done.  The draws fg_reset_scenario makes.
#pragma unroll
            for (int i = 0; i < N; ++i) { p[i] = scn_fresh_pm1(a.p, b, (uint32_t)i, off); v[i] = make_float2(0.f, 0.f); }
#pragma unroll
            for (int k = 0; k < M; ++k) {
                p[N + k] = scn_fresh_obstacle(a.p, b, k, a.sc.num_obstacles, off);
                v[N + k] = make_float2(a.sc.obstacle_vx, a.sc.obstacle_vy);
            }
#pragma unroll
            for (int l = 0; l < L; ++l) lm[l] = scn_fresh_pm1(a.p, b, SCN_LANDMARK_CODE | (uint32_t)l, off);
            fresh_lm = true;
            t_step = 0;
        }
        // ---- hand-over: the lane's [N][D] observation block and its rewards into LDS, for the writer wave ----
        const bool want_obs = a.obs_every <= 1 || (ks + 1) % a.obs_every == 0;      // uniform over the launch
        // one block: A - the writer has read the block of step ks - 1.  Two blocks: block (ks & 1) was last read for step
        // ks - 2, which the writer finished before it arrived at barrier B of step ks - 1
        if (!DB) __syncthreads();
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
            const float r = (KIND == FG_SCN_RANGE) ? a.sc.obs_range : INFINITY;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float2* const o = mine + i * (D / 2);
                int w = 0;
                o[w++] = v[i];
                if constexpr (BASIC) o[w++] = p[i];
#pragma unroll
                for (int l = 0; l < L; ++l) o[w++] = BASIC ? make_float2(lm[l].x - p[i].x, lm[l].y - p[i].y) : lm[l];
#pragma unroll
                for (int k = 0; k < M; ++k) o[w++] = make_float2(p[N + k].x - p[i].x, p[N + k].y - p[i].y);
                if constexpr (KIND == FG_SCN_PARTIAL) {
#pragma unroll
                    for (int kk = 0; kk < NBR; ++kk) {
                        int j = i + 1 + kk;                             // (i + 1 + kk) mod N
                        while (j >= N) j -= N;
                        o[w++] = make_float2(p[j].x - p[i].x, p[j].y - p[i].y);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < N - 1; ++t) {
                        const int j = t < i ? t : t + 1;                // the t-th OTHER agent, index order
                        o[w++] = make_float2(fminf(fmaxf(p[j].x - p[i].x, -r), r), fminf(fmaxf(p[j].y - p[i].y, -r), r));
                    }
                }
#pragma unroll
                for (int t = 0; t < N - 1; ++t) o[w++] = make_float2(0.f, 0.f);
            }
        }
        __syncthreads();                                // B: published
    }   // steps
    if (live && a.do_phys) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const size_t s = (size_t)b * N + i;
            a.px[s] = p[i].x; a.py[s] = p[i].y; a.vx[s] = v[i].x; a.vy[s] = v[i].y;
        }
#pragma unroll
        for (int k = 0; k < M; ++k) {
            reinterpret_cast<float2*>(a.opos)[(size_t)b * M + k] = p[N + k];
            reinterpret_cast<float2*>(a.ovel)[(size_t)b * M + k] = v[N + k];
        }
        if (fresh_lm) {
#pragma unroll
            for (int l = 0; l < L; ++l) reinterpret_cast<float2*>(a.lm)[(size_t)b * L + l] = lm[l];
        }
        if (a.step) a.step[b] = t_step;
    }
}

}  // namespace fg

#endif  // FG_SCN_LANE_KERNEL_HPP_
