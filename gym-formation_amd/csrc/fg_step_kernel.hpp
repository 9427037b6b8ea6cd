// fg_step_kernel.hpp - The fused single-launch step kernel (also the K-loop fallback and run-time N).
// Part of libformation_hip (gfx950); included by formation_hip.hip, one translation unit.
#ifndef FG_STEP_KERNEL_HPP_
#define FG_STEP_KERNEL_HPP_

#include "fg_common.hpp"
#include "fg_pair_loops.hpp"
#include "fg_obs_writers.hpp"


namespace fg {

constexpr int FG_STEP_WIDE_PACE = 1;     // s_sleep after every row of the rows writer in single-step launches above 64 agents

// ---------------------------------------------------------------------------
// the fused step / rollout kernel
//   NC  compile-time agent count (0 = run-time a.N)
//   G   lanes reserved per environment for the per-agent phases (power of two >= N)
//   T   threads per workgroup (>= E * G); ALL T threads stream observations
//   E   environments per workgroup
//   IDX also emit the landmark-index assignments
// LDS per env: see env_block_floats(); observation unit u >= N of any row is A[u], unit 0 of row i is A[3N + i].
// ---------------------------------------------------------------------------
template <int NC, int G, int T, int E, bool IDX, bool OPTS, bool KONE = false>
__global__ __launch_bounds__(T)
void step_kernel(const int pre_B, const int pre_N, const real* __restrict__ pre_px, const real* __restrict__ pre_py,
                 const real* __restrict__ pre_vx, const real* __restrict__ pre_vy, const real* __restrict__ pre_shape,
                 const real* __restrict__ pre_ivel, const int32_t* __restrict__ pre_step, const Args a) {
    // The leading scalar arguments repeat what phase 1 needs (batch size, agent count, state pointers): with
    // -amdgpu-kernarg-preload-count=16 the command processor hands them over in SGPRs at wave start, so the state
    // loads go out at once instead of behind a cold s_load of the argument block (one memory round trip off the
    // launch -> first-store chain of a single-step launch).  Everything else is read from `a` as before.
    // OPTS: World options no reference scenario enables (accel, max_speed, u_noise, walls);
    // compiled into a separate instantiation so that the common path keeps its registers.
    constexpr bool FLAT = (NC == 0);          // observation writer: register-cached rows (compile-time N) or flat decode
    static_assert(E * G <= T && (G <= 64 || E == 1), "bad geometry");
    extern __shared__ __attribute__((aligned(16))) real2 smem[];
    const int N = NC ? NC : (pre_N & 0xFFFF);
    // pre_N >> 16 = S > 1 (one env per workgroup only): S workgroups share an env and each streams 1/S of its
    // observation block - the second launch of the split step for batches with fewer envs than the chip has CUs
    // (host: launch_step).  Carried in the preloaded scalar so that the state loads do not wait for `a`.
    const int split = (E == 1) ? (pre_N >> 16) : 0;
    const unsigned sdiv = split > 1 ? (unsigned)split : 1u;
    const int part = (int)(blockIdx.x % sdiv);
    const int tid = threadIdx.x;
    const int e = (E == 1) ? 0 : tid / G;        // tid >= E*G: no agent, only streams observations
    const int i = (E == 1) ? tid : tid % G;
    const int b0 = (int)(blockIdx.x / sdiv) * E;
    const int b = b0 + e;
    const bool env_ok = (e < E) && (b < pre_B);   // this thread's lane group owns a live env
    const bool valid = env_ok && (i < N);
    const int El = min(E, pre_B - b0);

    const int NP = npad(N);
    constexpr int NPS = (NC > 0 && npad(NC > 0 ? NC : 1) <= 16) ? npad(NC > 0 ? NC : 1) : 0;   // small N: partners fetched up front
    // pair loops of 17..32 agents: 7 iterations (14 partners) per unrolled block, so that a block's LDS reads are in flight
    // together - only two of a SIMD's waves own agents in this geometry, little else hides the round trips
    // (27 x 4096: 15.6 -> 15.3 us, profiles/r02_step/step27_micro.txt)
    constexpr int UNR = (NC > 16 && NC <= 32) ? 7 : (T >= 1024 ? 1 : 2);   // 1024 threads: 128 VGPRs per lane, no room to unroll
    real2* const A = env_tables(smem, e < E ? e : 0, N);
    real2* const V = A + 3 * N;
    real2* const NV = A + 4 * N;             // -velocity, read by the row writer
    real* const QX = reinterpret_cast<real*>(A + 5 * N);
    real* const QY = QX + NP;
    real* const PX = QY + NP;
    real* const PY = PX + NP;
    real* const SX = PY + NP;
    real* const SY = SX + NP;
    real* const scratch = reinterpret_cast<real*>(env_tables(smem, E, N));
    volatile int* const reset_flag = reinterpret_cast<volatile int*>(scratch + 64);   // 2 ints after the 16x4 reduction partials
    // OPTS only, behind the 72 scratch floats (host: step_lds_bytes): per-agent mass and size of ALL agents (the same in
    // every env: one copy per workgroup) and each env's communication states - FgParams.agent_props / comm_state
    real* const MS = scratch + 72;
    real* const SZ = MS + NP;
    real* const FL = SZ + NP;                        // per-agent flags (FG_AGENT_*), as numbers
    real2* const CT = reinterpret_cast<real2*>(FL + NP) + (e < E ? e : 0) * N;
    const bool het = OPTS && a.p.agent_props != nullptr;
    const bool comm = OPTS && FLAT && a.p.comm_state != nullptr;
    const AgentProps me = agent_props_of(a.p, i, OPTS && i < N);

    const real one_minus_damp = 1.0f - a.p.damping;
    const real dt = a.p.dt;
    const real cutoff = a.p.dist_min + (FG_F64 ? 40.0f : 18.0f) * a.p.contact_margin;   // force beyond: < 1e2 k e^-18 ~ 1.5e-9 (fp64 build: e^-40)
    const real cutoff2 = cutoff * cutoff;
    const real thr2 = (real)((double)a.p.collide_thresh * (double)a.p.collide_thresh);
    const real invN = NC ? 1.0f / (real)(NC ? NC : 1) : a.inv_n;      // compile-time N: folded to the same correctly rounded value
    // ---- phase 1: state -> registers + LDS --------------------------------
    real2 p = make_real2(0.f, 0.f), v = make_real2(0.f, 0.f), s = make_real2(0.f, 0.f);
    int t_step = 0;
    const size_t sidx = (size_t)b * N + i;
    if (valid) {
        p = make_real2(pre_px[sidx], pre_py[sidx]);
        v = make_real2(pre_vx[sidx], pre_vy[sidx]);
        A[i] = p; V[i] = v; NV[i] = make_real2(-v.x, -v.y);
        QX[i] = p.x; QY[i] = p.y; PX[i] = p.x; PY[i] = p.y;
        if (a.do_post) {
            s = reinterpret_cast<const real2*>(pre_shape)[sidx];
            A[2 * N - 1 + i] = s;
            SX[i] = s.x; SY[i] = s.y;
            if (i < N - 1) A[N + i] = make_real2(0.f, 0.f);
            if (i == 0) A[3 * N - 1] = reinterpret_cast<const real2*>(pre_ivel)[b];
        }
    } else if (env_ok && i < NP) {              // sentinel partners of the packed pair loops
        QX[i] = FAR_AWAY; QY[i] = FAR_AWAY; PX[i] = FAR_AWAY; PY[i] = FAR_AWAY; SX[i] = FAR_AWAY; SY[i] = FAR_AWAY;
    }
    if (env_ok && pre_step) t_step = pre_step[b];
    if (tid < 2) reset_flag[tid] = 0;
    if constexpr (OPTS) {
        if (het && e == 0 && i < NP) {
            MS[i] = i < N ? me.mass : real(1); SZ[i] = i < N ? me.size : real(0); FL[i] = i < N ? (real)me.flags : real(0);
        }
        if (comm && valid) CT[i] = reinterpret_cast<const real2*>(a.p.comm_state)[sidx];
    }
    __syncthreads();

    // KONE: the single-step instantiation (K = 1, obs_every = 1 known at compile time: no step loop, no slot arithmetic)
    const int K = KONE ? 1 : a.K;
    // (Streaming the static two thirds of every row - zeros | ideal_shape | ideal_vel - from the agent-less waves while the others
    // run the pair loops does NOT shorten a single-step launch: 27 x 4096 15.2 -> 16.1 us (each 648-byte row then leaves as two
    // partial-line pieces), 32 x 4096 18.3 -> 17.9 where rows are whole lines: profiles/r05_step_early_ab.txt.)
    const uint64_t rbase = rng_base(a.p);            // read once: no load from the device counter inside the step loop
    for (int k = 0; k < K; ++k) {
        // the agent's own properties as the step loop reads them: the 1024-thread instantiations (128-VGPR budget) fetch them
        // again in every step instead of holding seven registers across the loop (they spilled to scratch memory)
        AgentProps mk = me;
        if constexpr (OPTS && T >= 1024) {
            int io = i;
            asm volatile("" : "+v"(io));
            mk = agent_props_of(a.p, io, io < N);
        }
        int slot = k;
        bool want_obs = a.do_post && a.obs != nullptr;
        if (!KONE && a.obs_every > 1) { want_obs = want_obs && ((k + 1) % a.obs_every == 0); slot = k / a.obs_every; }
        // ---- phase 2: World.step ------------------------------------------
        if (a.do_phys) {
            if (valid) {
                const real2 u = reinterpret_cast<const real2*>(a.act)[((size_t)k * pre_B + b) * N + i];
                real2 f;
                if (het) f = contact_force_het(QX, QY, MS, SZ, FL, N, i, p, mk.mass, mk.size, mk.flags, a.p.contact_force, a.p.contact_margin);
                else f = contact_force_packed<NPS, UNR>(QX, QY, NP, i, p, a.p.contact_force, a.p.contact_margin,
                                                        a.p.dist_min, cutoff2);
                if constexpr (OPTS) {
                    const real2 fa = action_force(a.p, me, u, (uint32_t)(b + a.p.env_index_base), (uint32_t)i, rbase + k);
                    f.x += fa.x; f.y += fa.y;
                    if (a.p.num_walls > 0) wall_forces(a.p, p, mk.size, f.x, f.y, (mk.flags & FG_AGENT_GHOST) != 0);
                } else {
                    f.x += a.p.mass * (a.p.sensitivity * u.x);
                    f.y += a.p.mass * (a.p.sensitivity * u.y);
                }
                const real m_own = OPTS ? mk.mass : a.p.mass;
                if (!(OPTS && (mk.flags & FG_AGENT_IMMOVABLE))) {        // core.py:266-267: an immovable entity keeps its state
                    v.x = v.x * one_minus_damp + (f.x / m_own) * dt;
                    v.y = v.y * one_minus_damp + (f.y / m_own) * dt;
                    if constexpr (OPTS) v = clamp_speed(mk.max_speed, v);
                    p.x += v.x * dt;
                    p.y += v.y * dt;
                }
                A[i] = p; V[i] = v; NV[i] = make_real2(-v.x, -v.y);
                PX[i] = p.x; PY[i] = p.y;
            }
            t_step += 1;
            // The barrier that publishes the post-step tables also carries one bit per group:
            // "some env of this workgroup finishes its episode in this step" (auto-reset only),
            // so the common no-reset step pays no extra barrier later.
            if (a.p.auto_reset && env_ok && i == 0 && t_step >= a.p.world_length) reset_flag[k & 1] = 1;
            __syncthreads();
            // the other parity slot is re-armed every step (it was last read in step k-1, which the barrier above
            // closed, and is next written in step k+1, after the barrier that ends this step)
            if (tid == 0) reset_flag[(k + 1) & 1] = 0;
        }

        if (a.do_post) {
          if (!a.obs_only) {
            // ---- phase 3: reward -------------------------------------------
            real sums[4] = {valid ? p.x : 0.f, valid ? p.y : 0.f, valid ? v.x : 0.f, valid ? v.y : 0.f};
            env_reduce<G, T, 4, R_SUM, R_SUM, R_SUM, R_SUM>(sums, scratch);
            const real mx = sums[0] * invN, my = sums[1] * invN;
            const real mvx = sums[2] * invN, mvy = sums[3] * invN;
            const real ptx = p.x - mx, pty = p.y - my;        // centred own position
            const real tx = s.x + mx, ty = s.y + my;          // own ideal point, un-centred
            real rowmin = INFINITY, colmin = INFINITY;
            int cnt = 0, arg_lm = 0, arg_ag = 0;
            if (valid)
                reward_pass_packed<IDX, NPS, UNR>(PX, PY, SX, SY, NP, p, ptx, pty, tx, ty, thr2,
                                        rowmin, colmin, cnt, arg_lm, arg_ag);
            if (het && valid)                                // per-pair penalty distance; `if agent.collide:` (formation_hd_env.py:71)
                cnt = (mk.flags & FG_AGENT_NO_COLLIDE) ? 0 : collision_count_het(PX, PY, SZ, N, i, p, mk.size, a.coll_scale);
            real red[3] = {valid ? rowmin : -INFINITY, valid ? colmin : -INFINITY, (real)cnt};
            env_reduce<G, T, 3, R_MAX, R_MAX, R_SUM, R_SUM>(red, scratch);
            const real H = rsqrt_(rmax(red[0], red[1]));
            const real2 iv = A[3 * N - 1];
            const real ex = iv.x - mvx, ey = iv.y - mvy;
            const real velterm = rsqrt_(ex * ex + ey * ey);
            const real indiv = (-H - velterm) - (real)cnt;
            const real shared = (real)(-(double)N * ((double)H + (double)velterm) - (double)red[2]);
            const bool is_done = t_step >= a.p.world_length;
            if (valid) {
                int io = i;
                if constexpr (T >= 1024) asm volatile("" : "+v"(io));   // 128-VGPR budget: no per-output base addresses kept across the step loop
                const size_t o = ((size_t)k * pre_B + b) * N + io;
                if (a.rew) a.rew[o] = shared;
                if (a.indiv) a.indiv[o] = indiv;
                if (a.done) a.done[o] = is_done ? 1 : 0;
            }
            if (IDX) {
                // scipy's witnesses: first maximiser of the row/col minima
                real w[2] = {(valid && rowmin == red[0]) ? (real)i : 1e9f,
                              (valid && colmin == red[1]) ? (real)i : 1e9f};
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
                const bool mine = is_done && env_ok;
                if (G > 64 ? mine : (__any(mine) != 0)) {
                    uint32_t c[4] = {(uint32_t)(b + a.p.env_index_base), (uint32_t)i, (uint32_t)(rbase + k),
                                     (uint32_t)((rbase + k) >> 32)};
                    philox4x32(c, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
                    real raw[2] = {valid ? u_pm1(c[2]) : 0.f, valid ? u_pm1(c[3]) : 0.f};
                    const real rx = raw[0], ry = raw[1];
                    env_reduce<G, T, 2, R_SUM, R_SUM, R_SUM, R_SUM>(raw, scratch);
                    if (mine && valid) {
                        p = make_real2(u_pm1(c[0]), u_pm1(c[1]));
                        v = make_real2(0.f, 0.f);
                        s = make_real2(rfma(-raw[0], invN, rx), rfma(-raw[1], invN, ry));   // explicit fma: same bits in every kernel
                        // The agent index as this rarely taken branch sees it is opaque: the ten LDS / global addresses below are
                        // then computed here, not hoisted out of the step loop into registers that live (and, in the 1024-thread
                        // instantiations with their 128-VGPR budget, spill to scratch memory) for the whole kernel.
                        int ir = i;
                        asm volatile("" : "+v"(ir));
                        const size_t sr = (size_t)b * N + ir;
                        A[ir] = p; V[ir] = v; NV[ir] = v; A[2 * N - 1 + ir] = s;
                        PX[ir] = p.x; PY[ir] = p.y; SX[ir] = s.x; SY[ir] = s.y;
                        reinterpret_cast<real2*>(a.shape)[sr] = s;
                        if (i == 0) {
                            uint32_t c2[4] = {(uint32_t)(b + a.p.env_index_base), 0xFFFFFFFFu, (uint32_t)(rbase + k),
                                              (uint32_t)((rbase + k) >> 32)};
                            philox4x32(c2, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
                            const real2 niv = make_real2(u_pm1(c2[0]), u_pm1(c2[1]));
                            A[3 * N - 1] = niv;
                            reinterpret_cast<real2*>(a.ivel)[b] = niv;
                        }
                    }
                    if (mine) t_step = 0;
                }
                __syncthreads();
            }
          }
            // ---- phase 5: observations --------------------------------------
            if (want_obs && !FLAT) {
                if constexpr (NC > 0)
                    write_obs_rows<NC, T / 64, E, (NC > 64 ? FG_STEP_WIDE_PACE : 0)>(env_tables(smem, 0, N), env_block_floats(NC) / 2, tid >> 6,
                                                  reinterpret_cast<real2*>(a.obs) +
                                                  ((size_t)slot * pre_B + b0) * (size_t)a.obs_pitch, (size_t)a.obs_pitch, El, 3,
                                                  part, split > 1 ? split : 1, !a.p.obs_placed);
            } else if (want_obs) {
                const unsigned n3 = 3u * N;                // (x,y) units per row
                const unsigned nenv = n3 * N;              // units per env = N rows
                // contiguous [B][N][6N]: the El envs of the workgroup are ONE span; padded env pitch: one span per env
                const bool contig = (size_t)a.obs_pitch == (size_t)nenv;
                const int spans = contig ? 1 : El;
                for (int sp = 0; sp < spans; ++sp) {
                    const size_t U0 = ((size_t)slot * pre_B + b0 + sp) * (size_t)a.obs_pitch;
                    const unsigned total = contig ? (unsigned)El * nenv : nenv;
                    const unsigned rp_base = (unsigned)sp * (unsigned)N;
                    const unsigned head = (unsigned)(U0 & 1);  // region start not 16-byte aligned
                    real2* const out2 = reinterpret_cast<real2*>(a.obs) + U0;
                    // unit (rp, u): rp = e*N + row is the row index inside the group, u the unit in
                    // the row.  Branch-free so that the LDS reads of several units overlap.
                    auto unit = [&](unsigned rp, unsigned u) -> real2 {
                        rp += rp_base;
                        const unsigned ee = (E == 1) ? 0u : rp / (unsigned)N;
                        const unsigned row = rp - ee * N;
                        const real2* AA = env_tables(smem, (int)ee, N);
                        const unsigned j = u - 1u;
                        const bool is_delta = j < (unsigned)(N - 1);
                        unsigned idx = is_delta ? j + (j >= row ? 1u : 0u) : u;
                        idx = (u == 0u) ? n3 + row : idx;
                        real2 val = AA[idx];
                        const real2 pi = AA[row];
                        val.x -= is_delta ? pi.x : 0.0f;
                        val.y -= is_delta ? pi.y : 0.0f;
                        if constexpr (OPTS) {                  // communication block: c_j of the other agents, j ascending
                            const unsigned jc = u - (unsigned)N;
                            if (comm && jc < (unsigned)(N - 1)) {
                                unsigned eo = ee * N;
                                if constexpr (T >= 1024) asm volatile("" : "+v"(eo));   // 128-VGPR budget: the table's address is made here
                                const real2* C = reinterpret_cast<const real2*>(FL + NP) + eo;
                                val = C[jc + (jc >= row ? 1u : 0u)];
                            }
                        }
                        return val;
                    };
                    if (head && tid == 0 && part == 0) out2[0] = unit(0u, 0u);
                    const unsigned npair = (total - head) >> 1;
                    realx4* const out4 = reinterpret_cast<realx4*>(out2 + head);
                    const unsigned du = (2u * T) % n3, drow = (2u * T) / n3;
                    // split step: this workgroup's share of the env's 16-byte pairs (all of them otherwise)
                    const unsigned q_lo = split > 1 ? (unsigned)((unsigned long long)npair * (unsigned)part / (unsigned)split) : 0u;
                    const unsigned q_hi = split > 1 ? (unsigned)((unsigned long long)npair * (unsigned)(part + 1) / (unsigned)split) : npair;
                    unsigned q = head + 2u * (q_lo + tid);
                    unsigned rp = q / n3;
                    unsigned u = q - rp * n3;
                    // 1024-thread workgroups have 128 VGPRs per lane: two units in flight there spilled (IDX / OPTS instantiations)
                    constexpr int FLAT_UNR = (T >= 1024) ? 1 : 2;
#pragma unroll FLAT_UNR
                    for (unsigned q2 = q_lo + tid; q2 < q_hi; q2 += T) {
                        unsigned u1 = u + 1u, rp1 = rp;
                        if (u1 == n3) { u1 = 0u; rp1 += 1u; }
                        const real2 x0 = unit(rp, u), x1 = unit(rp1, u1);
                        const realx4 w = {x0.x, x0.y, x1.x, x1.y};
                        out4[q2] = w;
                        u += du; rp += drow;
                        if (u >= n3) { u -= n3; rp += 1u; }
                    }
                    if (((total - head) & 1u) && tid == T - 1 && part == (split > 1 ? split - 1 : 0))
                        out2[total - 1] = unit((total - 1) / n3, (total - 1) % n3);
                }
            }
        }
        if (k + 1 < K) {
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

}  // namespace fg

#endif  // FG_STEP_KERNEL_HPP_
