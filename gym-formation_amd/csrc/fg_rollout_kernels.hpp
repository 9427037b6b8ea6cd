// fg_rollout_kernels.hpp - Producer / writer pipelined kernels (rollout over steps, single step over env batches).
// Part of libformation_hip (gfx950); included by formation_hip.hip, one translation unit.
#ifndef FG_ROLLOUT_KERNELS_HPP_
#define FG_ROLLOUT_KERNELS_HPP_

#include "fg_common.hpp"
#include "fg_pair_loops.hpp"
#include "fg_obs_writers.hpp"
#include "fg_policy_kernels.hpp"

namespace fg {

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
// LDS of the writer waves behind the env blocks, in float2 units: two tiles per wave (WR = 1 + rows per tile), the gather
// writer's table (WR = FG_WR_GATHER), nothing for the rows writer (WR = 0)
template <int NC, int WR, int NWW, int E> constexpr int roll_writer_units() {
    if constexpr (WR == 0) return 0;
    else if constexpr (WR == FG_WR_GATHER)                                     // the per-env table or the span table (one entry per piece)
        return gather_lut_units<NC>() > E * 3 * NC * NC / 2 ? gather_lut_units<NC>() : E * 3 * NC * NC / 2;
    else return 2 * NWW * tile_units<NC, WR - 1>();
}

// Dynamic LDS of rollout_kernel: env blocks | writer tiles / gather table | controller tables | reward hand-over.
// The reward hand-over - shared reward, individual reward, done flag of the workgroup's E x N agents, two buffers like the tables -
// lets the WRITER waves store them (lane-consecutive, whole 64-byte requests) so that the producer waves issue no store at all:
// on gfx9 loads and stores share one counter, and a producer's wait for its prefetched action was also a wait for its own reward
// stores behind the workgroup's observation stream.  Taken wherever it fits the 160 KiB.
template <int NC, int TW, int E, int WR, int PER> constexpr int roll_lds_base_bytes() {
    return E * roll_block_floats(NC) * (int)sizeof(real) + roll_writer_units<NC, WR, TW / 64, E>() * (int)sizeof(float2) +
           (PER > 0 ? E * policy_block_units(NC) * (int)sizeof(float2) : 0);
}
template <int NC, int E> constexpr int roll_rew_floats() { return 2 * 3 * E * NC; }
template <int NC, int TW, int E, int WR, int PER> constexpr bool roll_rew_in_lds() {
    // the gather-writer instantiations (8 / 9 agents: 8 % of their bytes are rewards and done flags, in 4-byte pieces from
    // the producers); the tile-writer kernels of 16-32 agents have no registers to spare for it (the headline kernel would
    // spill 16) and 1.4 % of their bytes to gain
    return WR == FG_WR_GATHER &&
           roll_lds_base_bytes<NC, TW, E, WR, PER>() + roll_rew_floats<NC, E>() * (int)sizeof(float) <= 160 * 1024;
}
template <int NC, int TW, int E, int WR, int PER> constexpr int roll_lds_bytes() {
    return roll_lds_base_bytes<NC, TW, E, WR, PER>() +
           (roll_rew_in_lds<NC, TW, E, WR, PER>() ? roll_rew_floats<NC, E>() * (int)sizeof(float) : 0);
}

// Workgroups of <= 512 threads must keep 4 waves per SIMD (<= 128 VGPRs): at 9 agents x >= 8192 envs two such
// workgroups share a CU, and a build whose writer needed 133 VGPRs ran that shape at half the rate.  (The 8-ary controller
// holds three 8-entry float2 tables per lane: its instantiations get the 168-register budget instead of spilling 57.)
template <int NC, int G, int TP, int TW, int E, int WR, int PER = 0, bool STREAM = false>
__global__ __launch_bounds__(TP + TW) __attribute__((amdgpu_waves_per_eu((TP + TW) <= 512 && PER < 8 ? 4 : 3)))
void rollout_kernel(const Args a) {
    constexpr bool POLICY = PER > 0;
    // WR: observation writer of the writer waves, 0 = register-cached rows, 1 + RT = LDS tiles of RT rows
    // STREAM: HBM-streaming form of the LDS-tile writer (fg_obs_writers.hpp)
    // PER > 0: closed loop - the action of step k is the demo controller (PER-ary hierarchy, N = PER^L) on the state
    //         step k-1 left, evaluated by the env's own lane group; a.act is not read, a.act_out records the actions
    static_assert(G <= 64 && E * G == TP && TW % 64 == 0 && TP % 64 == 0, "bad rollout geometry");
    constexpr int N = NC, NP = npad(NC), NWW = TW / 64;
    constexpr int NPS = NP <= 16 ? NP : 0;              // small N: partners fetched up front (fg_pair_loops.hpp)
    extern __shared__ __attribute__((aligned(16))) real2 smem[];
    real* const smemf = reinterpret_cast<real*>(smem);
    const int tid = threadIdx.x;
    const bool producer = tid < TP;
    const int e = producer ? tid / G : 0;
    const int i = tid % G;
    // env -> workgroup mapping: blocked (envs b0 .. b0+E-1), one contiguous span of observations per workgroup
    // (dealing envs round-robin over the workgroups was measured slower and is gone: profiles/README.md)
    // (dealing envs round-robin over the workgroups - a compact chip-wide window of the store stream - was measured
    // slower again in round 2, with and without a line-aligned env pitch: profiles/r02_pitch/)
    const int b0 = (int)blockIdx.x * E;
    const int b = b0 + e;
    const bool env_ok = producer && (b < a.B);
    const bool valid = env_ok && (i < N);
    const int El = min(E, a.B - b0);
    real* const blk = smemf + e * roll_block_floats(N);
    real2* const TB0 = reinterpret_cast<real2*>(blk);                 // tables of buffer 0; buffer 1 at + 5N
    real* const QX = blk + 20 * N;
    real* const QY = QX + NP; real* const PX = QY + NP; real* const PY = PX + NP;
    real* const SX = PY + NP; real* const SY = SX + NP;

    const real one_minus_damp = 1.0f - a.p.damping;
    const real dt = a.p.dt;
    const real cutoff = a.p.dist_min + (FG_F64 ? 40.0f : 18.0f) * a.p.contact_margin;
    const real cutoff2 = cutoff * cutoff;
    const real thr2 = (real)((double)a.p.collide_thresh * (double)a.p.collide_thresh);
    const real invN = 1.0f / (real)N;
    // read once: a (possible) load from the device counter inside the step loop makes the compiler drain the action
    // prefetch in every step (one counter for all outstanding memory operations on gfx9): 9 x 4096 x 128 2.42 -> 2.25
    // us/step, 8 x 8192 2.63 -> 2.39 (profiles/r04_rew_lds_ab.txt)
    const uint64_t rbase = rng_base(a.p);

    real2 p = make_real2(0.f, 0.f), v = p, s = p, iv = p;
    int t_step = 0;
    const size_t sidx = (size_t)b * N + i;
    if (valid) {
        p = make_real2(a.px[sidx], a.py[sidx]);
        v = make_real2(a.vx[sidx], a.vy[sidx]);
        s = reinterpret_cast<const real2*>(a.shape)[sidx];
        QX[i] = p.x; QY[i] = p.y; SX[i] = s.x; SY[i] = s.y;
        if (i < N - 1) { TB0[N + i] = make_real2(0.f, 0.f); TB0[5 * N + N + i] = make_real2(0.f, 0.f); }
    } else if (env_ok && i < NP) {
        QX[i] = FAR_AWAY; QY[i] = FAR_AWAY; PX[i] = FAR_AWAY; PY[i] = FAR_AWAY; SX[i] = FAR_AWAY; SY[i] = FAR_AWAY;
    }
    if (env_ok) { iv = reinterpret_cast<const real2*>(a.ivel)[b]; if (a.step) t_step = a.step[b]; }

    // Actions are fetched ahead of their step (small N leaves too little work between the load and its use to cover an HBM
    // round trip inside one step), in one of two forms:
    //  ACT4  four steps' actions back to back once per four steps (cur = produce calls 4 m ... 4 m + 3, nxt = the next four):
    //        fewer, larger read events between the store streams of a launch that streams to HBM - 27 x 4096 x 20 11.76 ->
    //        11.52 us/step, 9 x 4096 x 128 1.85 -> 1.60, 8 x 8192 2.43 -> 2.33, 32 x 4096 15.9 -> 15.5, 64 x 2048 30.8 -> 30.0
    //        (profiles/r04_batch4_ab.txt).  Eight more live registers: taken where the workgroup's register budget is 168
    //        (more than 512 threads) and by the gather-writer instantiations; the small chain-bound workgroups lose with it
    //        (9 x 4096 x 20 1.51 -> 1.59, 16 x 4096 4.18 -> 4.33) and keep
    //  else  one step ahead, two registers alternating roles over a loop unrolled by two, so that no copy (and with it the
    //        load's wait) lands inside the issuing step.
    // (with the reward hand-over through LDS a producer issues no stores at all, yet the four-step batches stay: without them
    // 8 x 65536 ran 18.0 -> 19.4 us/step and 9 x 16384 5.98 -> 6.16 - fewer, larger read events in a write-only stream are what
    // they buy, profiles/r05_rew_lds_ab.txt)
    constexpr bool ACT4 = !POLICY && ((TP + TW) > 512 || WR == FG_WR_GATHER);
    real2 u_even = make_real2(0.f, 0.f), u_odd = u_even;
    const size_t act_stride = (size_t)a.B * N;                  // real2 units between consecutive steps
    const real2* const act0 = reinterpret_cast<const real2*>(a.act) + (valid ? sidx : 0);
    const real2* act_next = act0 + (a.K > 1 ? act_stride : 0);
    // (four-step batches: a scalar base per step + the lane's 32-bit offset, so that no 64-bit per-lane pointer is held - and,
    // in the 128-register instantiations, spilled - across the step loop; B N < 2^29 entries)
    const uint32_t lane_off = valid ? (uint32_t)sidx : 0u;
    real2 cur[4], nxt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { cur[t] = make_real2(0.f, 0.f); nxt[t] = cur[t]; }
    auto load_batch = [&](real2 (&dst)[4], int j0) {          // actions of steps j0 ... j0 + 3 (clamped to the last step)
        if (valid) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                dst[t] = (reinterpret_cast<const real2*>(a.act) + (size_t)min(j0 + t, a.K - 1) * act_stride)[lane_off];
        }
    };
    if constexpr (ACT4) load_batch(cur, 0);
    else if (!POLICY && valid) u_even = *act0;
    // closed loop: controller tables of this env behind the env blocks and the writers' tiles
    real2* const pol_tab = reinterpret_cast<real2*>(smemf + E * roll_block_floats(N)) + roll_writer_units<NC, WR, NWW, E>() +
                            e * policy_block_units(N);
    // reward hand-over (see roll_lds_base_bytes): [2][rew | indiv | done][E][N]
    constexpr bool REWLDS = roll_rew_in_lds<NC, TW, E, WR, PER>();
    real* const rew_lds = smemf + roll_lds_base_bytes<NC, TW, E, WR, PER>() / (int)sizeof(real);

    // one producer step: World.step + reward of step k into table buffer (k & 1);
    // u_cur = action of step k (loaded during step k-1), u_nxt receives the action of step k+1
    auto produce = [&](int k, const real2& u_cur, real2& u_nxt) {
        real2* const A = TB0 + (k & 1) * 5 * N;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        real2 u_act = u_cur;
        if constexpr (POLICY) {
            // get_action_BFS(ezpolicy, obs, 3) on the observation the previous step (or the reset) returned:
            // its row 0 holds p_j - p_0, exactly this subtraction; QX / QY hold the current positions
            if (valid) { pol_tab[i] = make_real2(p.x - QX[0], p.y - QY[0]); pol_tab[N + i] = s; }
            WaveSync()();
            const real2* res = bfs_policy_env<(PER > 0 ? PER : 3), policy_levels_ct<NC, (PER > 0 ? PER : 3)>()>(pol_tab, N, a.pl, iv, i, G, WaveSync());
            if (valid) {
                u_act = res[i];
                if (a.act_out) reinterpret_cast<real2*>(a.act_out)[((size_t)k * a.B + b) * N + i] = u_act;
            }
        }
        if (valid) {
            // always issued (clamped to the last step): with a known number of younger loads the
            // wait for u_cur can leave this prefetch in flight
            if constexpr (!POLICY && !ACT4) {
                u_nxt = *act_next;
                act_next += (k + 2 < a.K) ? act_stride : 0;
            }
            real2 f = contact_force_packed<NPS>(QX, QY, NP, i, p, a.p.contact_force, a.p.contact_margin,
                                                 a.p.dist_min, cutoff2);
            f.x += a.p.mass * (a.p.sensitivity * u_act.x);
            f.y += a.p.mass * (a.p.sensitivity * u_act.y);
            v.x = v.x * one_minus_damp + (f.x / a.p.mass) * dt;
            v.y = v.y * one_minus_damp + (f.y / a.p.mass) * dt;
            p.x += v.x * dt;
            p.y += v.y * dt;
            PX[i] = p.x; PY[i] = p.y;
        }
        t_step += 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        real sums[4] = {valid ? p.x : 0.f, valid ? p.y : 0.f, valid ? v.x : 0.f, valid ? v.y : 0.f};
        env_reduce<G, G, 4, R_SUM, R_SUM, R_SUM, R_SUM>(sums, nullptr);
        const real mx = sums[0] * invN, my = sums[1] * invN;
        const real mvx = sums[2] * invN, mvy = sums[3] * invN;
        real rowmin = INFINITY, colmin = INFINITY;
        int cnt = 0, arg_lm = 0, arg_ag = 0;
        if (valid)
            reward_pass_packed<false, NPS>(PX, PY, SX, SY, NP, p, p.x - mx, p.y - my, s.x + mx, s.y + my, thr2,
                                           rowmin, colmin, cnt, arg_lm, arg_ag);
        real red[3] = {valid ? rowmin : -INFINITY, valid ? colmin : -INFINITY, (real)cnt};
        env_reduce<G, G, 3, R_MAX, R_MAX, R_SUM, R_SUM>(red, nullptr);
        const real H = rsqrt_(rmax(red[0], red[1]));
        const real ex = iv.x - mvx, ey = iv.y - mvy;
        const real velterm = rsqrt_(ex * ex + ey * ey);
        const bool is_done = t_step >= a.p.world_length;
        if (valid) {
            const real shared = (real)(-(double)N * ((double)H + (double)velterm) - (double)red[2]);
            const real own = (-H - velterm) - (real)cnt;
            if constexpr (REWLDS) {
                real* const R = rew_lds + (k & 1) * 3 * E * N + e * N + i;
                R[0] = shared; R[E * N] = own; reinterpret_cast<uint32_t*>(R)[2 * E * N] = is_done ? 1u : 0u;
            } else {
                const size_t o = ((size_t)k * a.B + b) * N + i;
                if (a.rew) a.rew[o] = shared;
                if (a.indiv) a.indiv[o] = own;
                if (a.done) a.done[o] = is_done ? 1 : 0;
            }
        }
        if (a.p.auto_reset) {
            const bool mine = is_done && env_ok;
            if (__any(mine) != 0) {
                uint32_t c[4] = {(uint32_t)(b + a.p.env_index_base), (uint32_t)i, (uint32_t)(rbase + k),
                                 (uint32_t)((rbase + k) >> 32)};
                philox4x32(c, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
                real raw[2] = {valid ? u_pm1(c[2]) : 0.f, valid ? u_pm1(c[3]) : 0.f};
                const real rx = raw[0], ry = raw[1];
                env_reduce<G, G, 2, R_SUM, R_SUM, R_SUM, R_SUM>(raw, nullptr);
                uint32_t c2[4] = {(uint32_t)(b + a.p.env_index_base), 0xFFFFFFFFu, (uint32_t)(rbase + k),
                                  (uint32_t)((rbase + k) >> 32)};
                philox4x32(c2, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
                if (mine) {
                    iv = make_real2(u_pm1(c2[0]), u_pm1(c2[1]));
                    t_step = 0;
                    if (valid) {
                        p = make_real2(u_pm1(c[0]), u_pm1(c[1]));
                        v = make_real2(0.f, 0.f);
                        s = make_real2(rfma(-raw[0], invN, rx), rfma(-raw[1], invN, ry));   // explicit fma: same bits in every kernel
                        SX[i] = s.x; SY[i] = s.y;
                        int ir = i, br = b;                // opaque: the global addresses of this rarely taken branch are made
                        asm volatile("" : "+v"(ir), "+v"(br));   // here, not kept in registers across the step loop
                        reinterpret_cast<real2*>(a.shape)[(size_t)br * N + ir] = s;
                        if (i == 0) reinterpret_cast<real2*>(a.ivel)[br] = iv;
                    }
                }
            }
        }
        if (valid) {                                   // publish this step's tables + next step's partners
            A[i] = p; A[3 * N + i] = v; A[4 * N + i] = make_real2(-v.x, -v.y);
            A[2 * N - 1 + i] = s;
            if (i == 0) A[3 * N - 1] = iv;
            QX[i] = p.x; QY[i] = p.y;
        }
    };

    // the span form of the gather writer for full workgroups of a contiguous tensor whose span starts on a 16-byte boundary
    // (slot * B * 3 N^2 even: every step alike); partial workgroups, padded env pitches and odd slabs take the per-env form
    constexpr bool SPAN_OK = WR == FG_WR_GATHER && (E * 3 * NC * NC) % 2 == 0 && (E - 1) * (roll_block_floats(NC) / 2) + 5 * NC < 32768;
    // (pieces in flight per writer wave: 4; 2 in the workgroups of up to 512 threads, which keep 4 waves per SIMD = 128 registers;
    // 1 in the 1024-thread workgroups - 128 registers again, and two pieces spilled two of them)
    using Span = SpanGather<(SPAN_OK ? NC : 2), NWW, (SPAN_OK ? E : 2), ((TP + TW) > 768 ? 1 : (TP + TW) <= 512 ? 2 : 4)>;
    uint2* const span_lut = reinterpret_cast<uint2*>(smemf + E * roll_block_floats(N));
    bool use_span = false;
    if constexpr (SPAN_OK)
        use_span = El == E && a.obs_pitch == 3LL * N * N && (((size_t)a.B * (size_t)a.obs_pitch) & 1) == 0 &&
                   (((size_t)b0 * (size_t)a.obs_pitch) & 1) == 0;
    if constexpr (WR == FG_WR_GATHER) {                   // the writers' table of operands, while the producers run step 0
        if (!producer) {
            if (use_span) Span::setup(span_lut, (tid - TP) >> 6, roll_block_floats(N) / 2);
            else build_gather_lut<NC>(reinterpret_cast<uint4*>(smemf + E * roll_block_floats(N)), tid - TP, TW);
        }
    }
    if (producer) __builtin_amdgcn_s_setprio(FG_PRODUCER_PRIO);   // the producers' dependent chain bounds small-N rollouts
    if (producer) produce(0, ACT4 ? cur[0] : u_even, u_odd);
    // every prologue load has landed before the loop: inside it the only loads in flight are the
    // action prefetches, and no leftover prologue dependency makes the compiler drain them early
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
    __syncthreads();
    // hand-over step k: producers run step k+1 (consuming u_cur) while writers stream step k
    auto pipeline_step = [&](int k, const real2& u_cur, real2& u_nxt) {
        if (producer) {
            if (k + 1 < a.K) produce(k + 1, u_cur, u_nxt);
        } else {
            if constexpr (REWLDS) {                          // step k's rewards and done flags: the workgroup's slice is contiguous
                const real* const R = rew_lds + (k & 1) * 3 * E * N;
                const int cnt = El * N;
                const size_t o0 = ((size_t)k * a.B + b0) * N;
                int q0 = tid - TP;
                asm volatile("" : "+v"(q0));                // addresses made here from the step's scalar base, not three 64-bit
                                                            // per-lane pointers kept (and spilled) across the step loop
                for (int q = q0; q < cnt; q += TW) {
                    if (a.rew) a.rew[o0 + q] = R[q];
                    if (a.indiv) a.indiv[o0 + q] = R[E * N + q];
                    if (a.done) a.done[o0 + q] = (uint8_t)reinterpret_cast<const uint32_t*>(R)[2 * E * N + q];
                }
            }
            int slot = k;
            bool want_obs = a.obs != nullptr;
            if (a.obs_every > 1) { want_obs = want_obs && ((k + 1) % a.obs_every == 0); slot = k / a.obs_every; }
            if (want_obs) {
                const size_t unit0 = ((size_t)slot * a.B + b0) * (size_t)a.obs_pitch;
                const size_t env_units = (size_t)a.obs_pitch;
                const real2* tables0 = reinterpret_cast<const real2*>(smemf) + (k & 1) * 5 * N;
                if constexpr (WR == 0)
                    write_obs_rows<NC, NWW, E>(tables0, roll_block_floats(N) / 2, (tid - TP) >> 6,
                                               reinterpret_cast<real2*>(a.obs) + unit0, env_units, El, 3);
                else if constexpr (WR == FG_WR_GATHER) {
                  if (use_span)
                    Span::write(span_lut, tables0, (tid - TP) >> 6, reinterpret_cast<real2*>(a.obs) + unit0);
                  else {
                    GatherEntries<(WR == FG_WR_GATHER ? NC : 2)> gather_entries;      // (the rare form: no state kept across steps)
                    write_obs_gather<NC, NWW, E>(tables0, roll_block_floats(N) / 2, (tid - TP) >> 6,
                                                 reinterpret_cast<const uint4*>(smemf + E * roll_block_floats(N)),
                                                 reinterpret_cast<real2*>(a.obs) + unit0, unit0, env_units, El, gather_entries);
                  }
                }
                else
                    write_obs_tiled<NC, NWW, E, WR - 1, STREAM>(tables0, roll_block_floats(N) / 2, (tid - TP) >> 6,
                                                                reinterpret_cast<real2*>(smemf + E * roll_block_floats(N)),
                                                                reinterpret_cast<real2*>(a.obs) + unit0, unit0, env_units, El);
            }
        }
        __syncthreads();
    };
    if constexpr (ACT4) {
        for (int k = 0; k < a.K; k += 4) {
            if (producer && k + 4 < a.K) load_batch(nxt, k + 4);   // first needed three steps from now
            pipeline_step(k, cur[1], u_odd);
            if (k + 1 < a.K) pipeline_step(k + 1, cur[2], u_odd);
            if (k + 2 < a.K) pipeline_step(k + 2, cur[3], u_odd);
            if (k + 3 < a.K) pipeline_step(k + 3, nxt[0], u_odd);
#pragma unroll
            for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
        }
    } else {
        for (int k = 0; k < a.K; k += 2) {
            pipeline_step(k, u_odd, u_even);
            if (k + 1 < a.K) pipeline_step(k + 1, u_even, u_odd);
        }
    }
    if (valid) {
        int is = i, bs = b;                                // (opaque: the state's addresses are not held across the step loop)
        asm volatile("" : "+v"(is), "+v"(bs));
        const size_t so = (size_t)bs * N + is;
        a.px[so] = p.x; a.py[so] = p.y; a.vx[so] = v.x; a.vy[so] = v.y;
    }
    if (a.step && env_ok && i == 0) a.step[b] = t_step;
}

// ---------------------------------------------------------------------------
// Pipelined K-step rollout for 64 < N <= 256: the same producer / writer split, with ONE
// producer wave per environment holding A = ceil(N/64) agents per lane (agent lane + 64 a), so
// that every reduction stays inside the wave and producers still need no barrier of their own.
// The partner loops load each partner pair once and update all A agents of the lane.
// ---------------------------------------------------------------------------
// BATCHES: the single-step form (a.K == 1) - the workgroup owns a.groups consecutive batches of E envs and pipelines over the
// batches instead of over the steps (open loop only).
template <int NC, int A, int E, int TW, int PER = 0, bool BATCHES = false>
__global__ __launch_bounds__(E * 64 + TW) void rollout_kernel_wide(const Args a) {
    constexpr bool POLICY = PER > 0;
    static_assert(!(BATCHES && POLICY), "the batch-pipelined single step takes its actions from the caller");
    static_assert(A * 64 >= NC && (A - 1) * 64 < NC && TW % 64 == 0, "bad wide rollout geometry");
    constexpr int N = NC, NP = npad(NC), NWW = TW / 64, TP = E * 64;
    extern __shared__ __attribute__((aligned(16))) real2 smem[];
    real* const smemf = reinterpret_cast<real*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index as a scalar: the env, its validity and every branch on them are wave-uniform (s_cbranch, no exec masks
    // held in SGPR pairs across the step loop - the kernel sat at the 102-SGPR limit and spilled 20-46 of them)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < E;
    const int e = producer ? wave : 0;
    const int NG = BATCHES ? max(1, a.groups) : 1;
    const int wg0 = blockIdx.x * E * NG;
    int b = wg0 + e;
    bool env_ok = producer && (b < a.B);
    real* const blk = smemf + e * roll_block_floats(N);
    real2* const TB0 = reinterpret_cast<real2*>(blk);
    real* const QX = blk + 20 * N;
    real* const QY = QX + NP; real* const PX = QY + NP; real* const PY = PX + NP;
    real* const SX = PY + NP; real* const SY = SX + NP;

    const real one_minus_damp = 1.0f - a.p.damping;
    const real dt = a.p.dt;
    const real cutoff = a.p.dist_min + (FG_F64 ? 40.0f : 18.0f) * a.p.contact_margin;
    const real cutoff2 = cutoff * cutoff;
    const real thr2 = (real)((double)a.p.collide_thresh * (double)a.p.collide_thresh);
    const real invN = 1.0f / (real)N;
    const real inv_k = 1.0f / a.p.contact_margin;
    const uint64_t rbase = rng_base(a.p);               // read once, not inside the step loop (see rollout_kernel)

    real2 p[A], v[A], s[A];
    // The action of the NEXT produce call, loaded one call ahead and in front of the current call's reward stores: a load at
    // the point of use queues up behind the workgroup's observation stream (round 4: open loop slower than closed loop,
    // 125 x 4096 260.7 vs 231.9 us/step), and its wait must not include stores issued after it (one counter on gfx9).
    real2 u_pre[A];
    // agent lane + 64 q of the wave's env exists: slices below the last one are full, so their test is the scalar env_ok
    auto valid = [&](int q) { return env_ok && (q < A - 1 || lane < N - 64 * (A - 1)); };
    real2 iv = make_real2(0.f, 0.f);
    int t_step = 0;
    if (producer) {                                     // row-independent table entries and loop sentinels: once
#pragma unroll
        for (int q = 0; q < A; ++q) {
            const int i = lane + 64 * q;
            if (i < N - 1) { TB0[N + i] = make_real2(0.f, 0.f); TB0[5 * N + N + i] = make_real2(0.f, 0.f); }
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
            p[q] = v[q] = s[q] = make_real2(0.f, 0.f);
            if (valid(q)) {
                const size_t o = (size_t)b * N + i;
                p[q] = make_real2(a.px[o], a.py[o]);
                v[q] = make_real2(a.vx[o], a.vy[o]);
                s[q] = reinterpret_cast<const real2*>(a.shape)[o];
                QX[i] = p[q].x; QY[i] = p[q].y; SX[i] = s[q].x; SY[i] = s[q].y;
            }
        }
        iv = make_real2(0.f, 0.f); t_step = 0;
        if (env_ok) { iv = reinterpret_cast<const real2*>(a.ivel)[b]; if (a.step) t_step = a.step[b]; }
    };
    auto load_actions = [&](int k) {                    // of step k -> u_pre
#pragma unroll
        for (int q = 0; q < A; ++q) {
            u_pre[q] = make_real2(0.f, 0.f);
            if (valid(q)) u_pre[q] = reinterpret_cast<const real2*>(a.act)[((size_t)k * a.B + b) * N + lane + 64 * q];
        }
    };
    auto store_group = [&]() {
#pragma unroll
        for (int q = 0; q < A; ++q) {
            if (valid(q)) {
                const size_t o = (size_t)b * N + lane + 64 * q;
                a.px[o] = p[q].x; a.py[o] = p[q].y; a.vx[o] = v[q].x; a.vy[o] = v[q].y;
            }
        }
        if (a.step && env_ok && lane == 0) a.step[b] = t_step;
    };

    real2* const pol_tab = reinterpret_cast<real2*>(smemf + E * roll_block_floats(N)) + e * policy_block_units(N);
    auto produce = [&](int k, int buf) {
        real2* const T = TB0 + buf * 5 * N;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const real2* pol_res = nullptr;
        if constexpr (POLICY) {                         // closed loop: see rollout_kernel
#pragma unroll
            for (int q = 0; q < A; ++q) {
                const int i = lane + 64 * q;
                if (valid(q)) { pol_tab[i] = make_real2(p[q].x - QX[0], p[q].y - QY[0]); pol_tab[N + i] = s[q]; }
            }
            WaveSync()();
            pol_res = bfs_policy_env<(PER > 0 ? PER : 3), policy_levels_ct<NC, (PER > 0 ? PER : 3)>()>(pol_tab, N, a.pl, iv, lane, 64, WaveSync());
        }
        real2 u_cur[A];
        if constexpr (!POLICY) {
#pragma unroll
            for (int q = 0; q < A; ++q) u_cur[q] = u_pre[q];
            if constexpr (!BATCHES) load_actions(min(k + 1, a.K - 1));   // always issued (clamped): a known number of younger loads
        }
        // ---- World.step: all A agents of the lane against each partner pair ----
        real fx[A], fy[A];
#pragma unroll
        for (int q = 0; q < A; ++q) { fx[q] = 0.f; fy[q] = 0.f; }
        if (env_ok) {
            for (int j = 0; j < NP; j += 2) {
                const realx2 qx = *reinterpret_cast<const realx2*>(QX + j);
                const realx2 qy = *reinterpret_cast<const realx2*>(QY + j);
#pragma unroll
                for (int q = 0; q < A; ++q) {
                    const int i = lane + 64 * q;
                    const realx2 dx = (realx2){p[q].x, p[q].x} - qx, dy = (realx2){p[q].y, p[q].y} - qy;
                    const realx2 d2 = dx * dx + dy * dy;
                    const bool n0 = (d2.x < cutoff2) && (j != i) && valid(q);
                    const bool n1 = (d2.y < cutoff2) && (j + 1 != i) && valid(q);
                    if (n0 || n1) {
                        auto add = [&](real ddx, real ddy, real dd2) {
                            const real d = hw_sqrt(dd2);
                            const real x = (a.p.dist_min - d) * inv_k;
                            const real pen = a.p.contact_margin * (rmax(x, real(0)) + hw_log(1.0f + hw_exp(-rabs(x))));
                            const real c = a.p.contact_force * pen * hw_rcp(d);
                            fx[q] += ddx * c; fy[q] += ddy * c;
                        };
                        if (n0) add(dx.x, dy.x, d2.x);
                        if (n1) add(dx.y, dy.y, d2.y);
                    }
                }
            }
        }
        // real sums are reduced per 64-agent slice and then combined slice by slice: the exact
        // association order of step_kernel (wave butterfly, then waves in order) -> bit-identical
        real sums[4] = {0.f, 0.f, 0.f, 0.f};
        real part[A][4];
#pragma unroll
        for (int q = 0; q < A; ++q) {
            part[q][0] = part[q][1] = part[q][2] = part[q][3] = 0.f;
            if (valid(q)) {
                const int i = lane + 64 * q;
                real2 u;
                if constexpr (POLICY) {
                    u = pol_res[i];
                    if (a.act_out) reinterpret_cast<real2*>(a.act_out)[((size_t)k * a.B + b) * N + i] = u;
                } else {
                    u = u_cur[q];
                }
                const real ffx = fx[q] + a.p.mass * (a.p.sensitivity * u.x);
                const real ffy = fy[q] + a.p.mass * (a.p.sensitivity * u.y);
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
        const real mx = sums[0] * invN, my = sums[1] * invN;
        const real mvx = sums[2] * invN, mvy = sums[3] * invN;
        // ---- reward pass ----
        real rowmin[A], colmin[A];
        int cnt[A];
#pragma unroll
        for (int q = 0; q < A; ++q) { rowmin[q] = INFINITY; colmin[q] = INFINITY; cnt[q] = -1; }
        if (env_ok) {
            for (int j = 0; j < NP; j += 2) {
                const realx2 qx = *reinterpret_cast<const realx2*>(PX + j);
                const realx2 qy = *reinterpret_cast<const realx2*>(PY + j);
                const realx2 sx = *reinterpret_cast<const realx2*>(SX + j);
                const realx2 sy = *reinterpret_cast<const realx2*>(SY + j);
#pragma unroll
                for (int q = 0; q < A; ++q) {
                    const realx2 cx = qx - (realx2){p[q].x, p[q].x}, cy = qy - (realx2){p[q].y, p[q].y};
                    const realx2 dc = cx * cx + cy * cy;
                    cnt[q] += (dc.x < thr2 ? 1 : 0) + (dc.y < thr2 ? 1 : 0);
                    const real ptx = p[q].x - mx, pty = p[q].y - my;
                    const realx2 rx = (realx2){ptx, ptx} - sx, ry = (realx2){pty, pty} - sy;
                    const realx2 dr = rx * rx + ry * ry;
                    const real tx = s[q].x + mx, ty = s[q].y + my;
                    const realx2 ux = qx - (realx2){tx, tx}, uy = qy - (realx2){ty, ty};
                    const realx2 dq = ux * ux + uy * uy;
                    rowmin[q] = rmin(rmin(rowmin[q], dr.x), dr.y);
                    colmin[q] = rmin(rmin(colmin[q], dq.x), dq.y);
                }
            }
        }
        real red[3] = {-INFINITY, -INFINITY, 0.f};
#pragma unroll
        for (int q = 0; q < A; ++q) {
            if (valid(q)) {
                cnt[q] += (thr2 > 0.0f ? 0 : 1);
                red[0] = rmax(red[0], rowmin[q]); red[1] = rmax(red[1], colmin[q]); red[2] += (real)cnt[q];
            }
        }
        env_reduce<64, 64, 3, R_MAX, R_MAX, R_SUM, R_SUM>(red, nullptr);
        const real H = rsqrt_(rmax(red[0], red[1]));
        const real ex = iv.x - mvx, ey = iv.y - mvy;
        const real velterm = rsqrt_(ex * ex + ey * ey);
        const bool is_done = t_step >= a.p.world_length;
        const real shared = (real)(-(double)N * ((double)H + (double)velterm) - (double)red[2]);
#pragma unroll
        for (int q = 0; q < A; ++q) {
            if (valid(q)) {
                const size_t o = ((size_t)k * a.B + b) * N + lane + 64 * q;
                if (a.rew) a.rew[o] = shared;
                if (a.indiv) a.indiv[o] = (-H - velterm) - (real)cnt[q];
                if (a.done) a.done[o] = is_done ? 1 : 0;
            }
        }
        if (a.p.auto_reset && is_done && env_ok) {            // wave-uniform: the wave owns one env
            real raw[2] = {0.f, 0.f};
            real rawp[A][2];
            real rx[A], ry[A];
            uint32_t c0[A], c1[A];
#pragma unroll
            for (int q = 0; q < A; ++q) {
                uint32_t c[4] = {(uint32_t)(b + a.p.env_index_base), (uint32_t)(lane + 64 * q), (uint32_t)(rbase + k),
                                 (uint32_t)((rbase + k) >> 32)};
                philox4x32(c, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
                c0[q] = c[0]; c1[q] = c[1];
                rx[q] = valid(q) ? u_pm1(c[2]) : 0.f; ry[q] = valid(q) ? u_pm1(c[3]) : 0.f;
                rawp[q][0] = rx[q]; rawp[q][1] = ry[q];
            }
#pragma unroll
            for (int q = 0; q < A; ++q) {
                env_reduce<64, 64, 2, R_SUM, R_SUM, R_SUM, R_SUM>(rawp[q], nullptr);
                if (q == 0) { raw[0] = rawp[0][0]; raw[1] = rawp[0][1]; } else { raw[0] += rawp[q][0]; raw[1] += rawp[q][1]; }
            }
            uint32_t c2[4] = {(uint32_t)(b + a.p.env_index_base), 0xFFFFFFFFu, (uint32_t)(rbase + k),
                              (uint32_t)((rbase + k) >> 32)};
            philox4x32(c2, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32));
            iv = make_real2(u_pm1(c2[0]), u_pm1(c2[1]));
            t_step = 0;
#pragma unroll
            for (int q = 0; q < A; ++q) {
                if (valid(q)) {
                    const int i = lane + 64 * q;
                    const size_t o = (size_t)b * N + i;
                    p[q] = make_real2(u_pm1(c0[q]), u_pm1(c1[q]));
                    v[q] = make_real2(0.f, 0.f);
                    s[q] = make_real2(rfma(-raw[0], invN, rx[q]), rfma(-raw[1], invN, ry[q]));
                    SX[i] = s[q].x; SY[i] = s[q].y;
                    reinterpret_cast<real2*>(a.shape)[o] = s[q];
                }
            }
            if (lane == 0) reinterpret_cast<real2*>(a.ivel)[b] = iv;
        }
#pragma unroll
        for (int q = 0; q < A; ++q) {
            if (valid(q)) {
                const int i = lane + 64 * q;
                T[i] = p[q]; T[3 * N + i] = v[q]; T[4 * N + i] = make_real2(-v[q].x, -v[q].y);
                T[2 * N - 1 + i] = s[q];
                QX[i] = p[q].x; QY[i] = p[q].y;
            }
        }
        if (env_ok && lane == 0) T[3 * N - 1] = iv;
    };

    constexpr bool per_step = BATCHES;
    const int total = per_step ? NG : a.K;
    // Two loops, one per role, meeting at the same 1 + total workgroup barriers: the branch is wave-uniform (scalar), and
    // each loop keeps only its own role's pointers and constants in SGPRs (one common loop held both sets live and spilled).
    if (producer) {
        load_group(0);
        if constexpr (!POLICY) load_actions(0);
        produce(0, 0); if (per_step) store_group();
        __syncthreads();
        for (int it = 0; it < total; ++it) {
            if (it + 1 < total) {
                if (per_step) { load_group(it + 1); load_actions(0); produce(0, (it + 1) & 1); store_group(); }
                else produce(it + 1, (it + 1) & 1);
            }
            __syncthreads();
        }
        if (!per_step) store_group();
    } else {
        const int w = wave - E;
        __syncthreads();
        for (int it = 0; it < total; ++it) {
            const int k = per_step ? 0 : it;
            const int b0 = wg0 + (per_step ? it * E : 0);
            const int El = min(E, a.B - b0);
            int slot = k;
            bool want_obs = a.obs != nullptr && El > 0;
            if (a.obs_every > 1) { want_obs = want_obs && ((k + 1) % a.obs_every == 0); slot = k / a.obs_every; }
            if (want_obs) {
                const size_t unit0 = ((size_t)slot * a.B + b0) * (size_t)a.obs_pitch;
                // (1 KiB store instructions on the absolute 64-byte grid, every unit recomputed from two LDS reads - the span form of
                // the gather writer - lose here: 81 x 2048 49.5 -> 54.7 us/step, 243 x 8192 1 740 -> 2 065, with eight writer
                // waves 57 / 1 943: profiles/r05_wide_span_ab.txt.  The register-cached rows writer stays.)
                write_obs_rows<NC, NWW, E>(reinterpret_cast<const real2*>(smemf) + (it & 1) * 5 * N,
                                           roll_block_floats(N) / 2, w,
                                           reinterpret_cast<real2*>(a.obs) + unit0, (size_t)a.obs_pitch, El, 3);
            }
            __syncthreads();
        }
    }
}

}  // namespace fg

#endif  // FG_ROLLOUT_KERNELS_HPP_
