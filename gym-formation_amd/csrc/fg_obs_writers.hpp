// fg_obs_writers.hpp - Observation writers: register-cached rows, LDS tiles, and the table-driven gather writer (8 / 9 agents).
// Part of libformation_hip (gfx950); included by formation_hip.hip, one translation unit.
#ifndef FG_OBS_WRITERS_HPP_
#define FG_OBS_WRITERS_HPP_

#include "fg_common.hpp"

namespace fg {

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
template <class V2> FG_DEV V2 lds_if(bool c, const V2* __restrict__ p, int idx_if_true) {
    V2 t = p[c ? idx_if_true : 0];
    t.x = c ? t.x : 0.0f; t.y = c ? t.y : 0.0f;
    return t;
}

constexpr bool FG_ROWS_MERGED = true;      // N > 64: rows as chunks of 64 consecutive units across the field boundary (below)
template <int NC, int NW, int E, int PACE = 0>
FG_DEV void write_obs_rows(const real2* __restrict__ tables0, int env_stride, int w,
                           real2* __restrict__ out_env0, size_t env_units, int El, int parts,
                           int wg_part = 0, int wg_parts = 1, bool pace_on = true) {
    // tables0 / env_stride: A-table of env 0 and the distance (real2) to the next env's;
    // out_env0 / env_units: observation block of env 0 and the distance (real2 units) to the next env's
    //   (3 N^2 when the [B][N][6N] tensor is contiguous; larger with a padded env pitch or strided env ownership)
    // w: index of this wave among the NW waves that share the job
    // parts: bit 0 = relative-position units, bit 1 = static units (zeros | shape | ideal_vel)
    // wg_part / wg_parts (N > 64, one env per workgroup): this workgroup is one of wg_parts that share the env
    constexpr int N = NC;
    constexpr int WPE = (E >= NW) ? 1 : NW / E;             // waves sharing one env
    constexpr int ESTEP = (E >= NW) ? NW : 1;               // env stride of one wave
    static_assert((E >= NW) ? (E % NW == 0) : (NW % E == 0), "waves and envs must tile");
    const int lane = threadIdx.x & 63;
    const int row0 = (E >= NW) ? 0 : w % WPE;
    constexpr unsigned ROWU = 3u * N;                       // units per row
    for (int ee = (E >= NW) ? w : w / WPE; ee < El; ee += (E >= NW ? ESTEP : E)) {
        const real2* __restrict__ AA = tables0 + (size_t)ee * env_stride;
        real2* __restrict__ out = out_env0 + (size_t)ee * env_units;
        if constexpr (N <= 64) {
            // Blocks of RW = 64/N rows: one wave store covers the relative-position part of the
            // whole block, then the static part (zeros | ideal_shape | ideal_vel, the same for every
            // row: register-resident) of the same rows follows at once, so that the cache lines a
            // row shares with its neighbours are completed back to back.
            constexpr int RW = 64 / N;
            const int rsub = lane / N, u = lane - rsub * N;
            const bool act = rsub < RW;
            const real2 zero = make_real2(0.f, 0.f);
            const real2 Pm = lds_if(act && u >= 1, AA, u - 1);
            const real2 Pu = lds_if(act && u >= 1, AA, u);
            const int xoff = (u == 0) ? 4 * N : 0;          // lane u = 0 reads -v_row (NV = A + 4N)
            constexpr int CS = (2 * N + 63) / 64;           // 64-unit chunks of the static part
            constexpr int RS = (2 * N <= 64) ? 64 / (2 * N) : 1;   // rows per static store
            const int ssub = (2 * N <= 64) ? lane / (2 * N) : 0;
            const int sidx = (2 * N <= 64) ? lane - ssub * 2 * N : lane;
            real2 sv[CS];
#pragma unroll
            for (int c = 0; c < CS; ++c)
                sv[c] = lds_if(ssub < RS && sidx + 64 * c < 2 * N, AA, N + sidx + 64 * c);
#pragma unroll 2
            for (int rb = row0; rb < N; rb += RW * WPE) {
                const int r = rb + rsub * WPE;
                if ((parts & 1) && act && r < N) {
                    const real2 x = AA[xoff + r];
                    const real2 c = (u - 1 >= r) ? Pu : Pm;
                    out[(unsigned)r * ROWU + (unsigned)u] = make_real2(c.x - x.x, c.y - x.y);
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
        } else if constexpr (FG_ROWS_MERGED && ((3 * N + 63) / 64 < (N + 63) / 64 + (2 * N + 63) / 64)) {
            // ---- N > 64: one row per iteration as ceil(3N / 64) chunks of 64 consecutive units, whatever field a unit
            // belongs to: 4 store instructions per 243-unit row at 81 agents instead of 2 (relative positions) + 3 (static
            // part) with two poorly filled ones (81 x 2048 rollout: 56.3 -> 55.5 us/step, profiles/r03_wide/ab_rows.txt).
            // Taken only where it saves a store per row: at 243 agents both forms need 12 and this one measured 1.5 % slower.  Whether lane's unit of chunk c is a relative position (u < N) or static
            // (u >= N: A[u], the same for every row) does not depend on the row, so both kinds sit in the same registers:
            // static lanes hold A[u] in BOTH position registers and subtract nothing.
            constexpr int CR = (3 * N + 63) / 64;                 // chunks per row
            constexpr int C_MIX = N / 64;                         // the chunk that holds unit N (dyn and static lanes)
            real2 Pm[CR], Pu[CR];
#pragma unroll
            for (int c = 0; c < CR; ++c) {
                const int u = lane + 64 * c;
                if (c <= C_MIX) {
                    const bool dyn = u >= 1 && u < N, stat = u >= N && u < 3 * N;
                    Pm[c] = lds_if(dyn || stat, AA, stat ? u : u - 1);
                    Pu[c] = lds_if(dyn || stat, AA, u);
                } else {
                    Pm[c] = Pu[c] = lds_if(u < 3 * N, AA, u);
                }
            }
            const int nshare = WPE * wg_parts, share = wg_part * WPE + row0;     // (wave, workgroup) shares of the rows
            const int PER = (N + nshare - 1) / nshare;
            const int r_end = min(N, (share + 1) * PER);
#pragma unroll 2
            for (int r = share * PER; r < r_end; ++r) {
                const real2 xp = AA[r];                    // p_row, wave-uniform broadcast
                const real2 x0 = AA[(lane == 0 ? 4 * N : 0) + r];   // lane 0 of chunk 0: -v_row
                real2* __restrict__ orow = out + (unsigned)r * ROWU;
#pragma unroll
                for (int c = 0; c < CR; ++c) {
                    const int u = lane + 64 * c;
                    real2 val;
                    if (c < C_MIX || (c == C_MIX && (N % 64) == 0)) {           // every lane a relative position
                        const real2 x = (c == 0) ? x0 : xp;
                        const real2 cc = (u - 1 >= r) ? Pu[c] : Pm[c];
                        val = make_real2(cc.x - x.x, cc.y - x.y);
                    } else if (c == C_MIX) {                                     // relative positions, then static units
                        const bool dyn = u < N;
                        const real2 x = (c == 0) ? x0 : xp;
                        const real2 cc = (u - 1 >= r) ? Pu[c] : Pm[c];
                        val = make_real2(cc.x - (dyn ? x.x : 0.0f), cc.y - (dyn ? x.y : 0.0f));
                    } else {
                        val = Pu[c];
                    }
                    if (u < 3 * N) orow[u] = val;
                }
                if (PACE > 0 && pace_on) __builtin_amdgcn_s_sleep(PACE);
            }
        } else {
            // ---- N > 64: one row per iteration, register-cached chunks of 64 units ----
            constexpr int CD = (N + 63) / 64, CS = (2 * N + 63) / 64;
            const real2 zero = make_real2(0.f, 0.f);
            real2 Pm[CD], Pu[CD], sv[CS];
#pragma unroll
            for (int c = 0; c < CD; ++c) {
                const int u = lane + 64 * c;
                Pm[c] = lds_if(u >= 1 && u < N, AA, u - 1);
                Pu[c] = lds_if(u >= 1 && u < N, AA, u);
            }
#pragma unroll
            for (int c = 0; c < CS; ++c) sv[c] = lds_if(lane + 64 * c < 2 * N, AA, N + lane + 64 * c);
            // the WPE waves that share an env take contiguous row ranges (one contiguous run of the observation
            // block per wave; every WPE-th row measured 2-3 % slower at 81 x 2048, profiles/README.md)
            const int nshare = WPE * wg_parts, share = wg_part * WPE + row0;     // (wave, workgroup) shares of the rows
            const int PER = (N + nshare - 1) / nshare;
            const int r_end = min(N, (share + 1) * PER);
#pragma unroll 2
            for (int r = share * PER; r < r_end; ++r) {
                const real2 xp = AA[r];                    // p_row, wave-uniform broadcast
                const real2 x0 = AA[(lane == 0 ? 4 * N : 0) + r];
                real2* __restrict__ orow = out + (unsigned)r * ROWU;
#pragma unroll
                for (int c = 0; c < CD; ++c) {
                    const int u = lane + 64 * c;
                    const real2 x = (c == 0) ? x0 : xp;
                    const real2 cc = (u - 1 >= r) ? Pu[c] : Pm[c];
                    if ((parts & 1) && u < N) orow[u] = make_real2(cc.x - x.x, cc.y - x.y);
                }
#pragma unroll
                for (int c = 0; c < CS; ++c)
                    if ((parts & 2) && lane + 64 * c < 2 * N) orow[N + lane + 64 * c] = sv[c];
                // PACE: 64 idle cycles after every row.  Single-step launches at 81 agents have 16 waves per CU bursting
                // rows at once: 69.6-71.4 -> 65.9-67.9 us (81 x 2048) on an ordinary allocation; no effect in the pipelined
                // rollout kernels (4-8 writer waves per CU) and a small loss for the agent counts whose step buffer sits in
                // the Infinity Cache.  A buffer spread over the device memory (FgParams.obs_placed) takes the bursts:
                // 54.0 -> 53.4 us without the pause (pace_on = false)
                if (PACE > 0 && pace_on) __builtin_amdgcn_s_sleep(PACE);
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
// Line ownership (waves own whole envs, E >= NW): every agent count of the reference is odd, so consecutive env
// blocks share a 128-byte line, and that line would be written in two instalments by two different waves at
// different times - partial-line writes that cost ~4 % of the stream (profiles/r02_store/env_order.txt: "pad
// written" against "pad skipped").  The wave of env e therefore also stores the first units of env e+1 up to the next
// line boundary (composed from env e+1's tables, same LDS buffer) and the wave of env e+1 starts at that boundary;
// a workgroup's 16-env span starts and ends on line boundaries, so every store covers whole lines.  With a padded env
// pitch (next multiple of 128 bytes) the same slot holds the pad, which is then written as zeros.
// ---------------------------------------------------------------------------
template <int NC, int RT> constexpr int tile_units() { return (3 * NC * RT + 16 + 2 + 1) & ~1; }

// `tiles` holds TWO tiles per writing wave: tile t+1 is composed while tile t drains.
// STREAM: the HBM-streaming form (line ownership + paced stores, see above and stream() below); false: the plain form
// (every wave stores exactly its env's bytes, all LDS reads of a tile in flight before its stores).
template <int NC, int NW, int E, int RT, bool STREAM>
FG_DEV void write_obs_tiled(const float2* __restrict__ tables0, int env_stride, int w, float2* __restrict__ tiles,
                            float2* __restrict__ out_env0, size_t unit0, size_t env_units, int El) {
    constexpr int N = NC;
    constexpr int WPE = (E >= NW) ? 1 : NW / E;
    static_assert((E >= NW) ? (E % NW == 0) : (NW % E == 0), "waves and envs must tile");
    static_assert(N <= 32 && N % RT == 0, "tiled writer: N <= 32, RT divides N");
    constexpr unsigned ROWU = 3u * N, TU = ROWU * RT;
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
    // (the workgroup's tiles dealt round-robin over its writer waves - ONE stream of NW consecutive tiles per workgroup
    // instead of NW streams an env apart - measured slower: 27 x 4096 x 20 13.67 vs 12.9-13.07 us/step on an ordinary allocation,
    // 11.65 vs 11.43 placed, profiles/r05_deal_ab.txt)
    const int total = n_env * MY_TILES;

    // line ownership: units of the NEXT env (or pad) a wave appends to its env's last tile / skips at its first tile
    constexpr unsigned ENVU = ROWU * N;
    const unsigned pad = (unsigned)(env_units - ENVU);
    // STREAM = the launch streams to HBM (host: 27 agents, < 16 384 envs, rollout buffer beyond the Infinity Cache): line
    // ownership here, paced stores in stream().  Ownership alone is worth ~2 % and a steadier rate, the pacing ~8 %.
    // (9 agents, one 1 944-byte tile per env: the extra compose step costs more than the lines save - plain form.)
    static_assert(!STREAM || (E >= NW && NC >= 16), "streaming form: waves own whole envs, rows of >= 16 units");
    const bool own = STREAM && ((unit0 & 15) == 0) && (pad == 0 || (pad < 16 && (env_units & 15) == 0));
    auto head_units = [&](int ee) -> unsigned {                        // of env ee, owned by the wave of env ee - 1
        return (own && pad == 0) ? (16u - (unsigned)(((size_t)ee * env_units) & 15)) & 15u : 0u;
    };
    auto extra_units = [&](int ee) -> unsigned {                       // appended behind env ee's last row
        if (!own) return 0u;
        if (pad) return pad;
        return (ee + 1 < El) ? head_units(ee + 1) : 0u;
    };

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
        const unsigned par = (unsigned)((unit0 + (size_t)ee * env_units + (size_t)r0 * ROWU) & 1);
        float2* img = tile0 + (t & 1) * tile_units<NC, RT>() + par;     // no restrict: the two tiles alternate
#pragma unroll
        for (int rb = 0; rb < RT; rb += RW) {
            const int rl = rb + rsub;
            if (act && rl < RT) {
                const int r = r0 + rl;
                const float2 x = AA[xoff + r];
                const float2 c = (u - 1 >= r) ? Pu : Pm;
                img[(unsigned)rl * ROWU + (unsigned)u] = make_float2(c.x - x.x, c.y - x.y);
            }
        }
#pragma unroll
        for (int rb = 0; rb < RT; rb += RS) {
            const int rl = rb + ssub;
            if (ssub < RS && rl < RT) img[(unsigned)rl * ROWU + (unsigned)(N + sidx)] = sv;
        }
        if (STREAM && r0 + RT == N) {                                  // last tile of the env: the line's remainder
            const unsigned x = extra_units(ee);
            if ((unsigned)lane < x) {
                float2 val = make_float2(0.f, 0.f);                    // padded pitch: zeros
                if (pad == 0) {                // row 0 of env ee + 1: [v_0 | p_u - p_0 (1 <= u < N) | A[u] (u >= N) ...]
                    static_assert(ROWU >= 16, "the line remainder must stay inside row 0 of the next env");
                    const float2* __restrict__ A1 = tables0 + (size_t)(ee + 1) * env_stride;
                    const float2 a = A1[lane == 0 ? 3 * N : lane], b = A1[0];
                    const bool rel = lane >= 1 && lane < N;
                    val = make_float2(a.x - (rel ? b.x : 0.f), a.y - (rel ? b.y : 0.f));
                }
                img[TU + (unsigned)lane] = val;
            }
        }
    };
    auto stream = [&](int t) {
        int ee, r0; locate(t, ee, r0);
        if (r0 >= N) return;
        const unsigned par = (unsigned)((unit0 + (size_t)ee * env_units + (size_t)r0 * ROWU) & 1);
        const float2* img = tile0 + (t & 1) * tile_units<NC, RT>() + par;
        float2* __restrict__ out = out_env0 + (size_t)ee * env_units + (size_t)r0 * ROWU;
        if constexpr (!STREAM) {
            if (par && lane == 0) out[0] = img[0];
            constexpr unsigned NPMAX = TU >> 1;
            const unsigned npair = (TU - par) >> 1;
            const f32x4* src4 = reinterpret_cast<const f32x4*>(img + par);
            f32x4* __restrict__ dst4 = reinterpret_cast<f32x4*>(out + par);
#pragma unroll
            for (unsigned q0 = 0; q0 < NPMAX; q0 += 64) {
                const unsigned q = q0 + lane;
                if (q < npair) dst4[q] = src4[q];
            }
            if (((TU - par) & 1u) && lane == 63) out[TU - 1] = img[TU - 1];
        } else {
            const unsigned s = (r0 == 0) ? head_units(ee) : 0u;        // first unit this wave stores
            const unsigned end = TU + ((r0 + RT == N) ? extra_units(ee) : 0u);
            const unsigned p2 = (par + s) & 1u;                        // 8-byte head to reach 16-byte alignment
            if (p2 && lane == 0) out[s] = img[s];
            constexpr unsigned NPMAX = (TU + 16) >> 1;
            const unsigned npair = (end - s - p2) >> 1;
            const f32x4* src4 = reinterpret_cast<const f32x4*>(img + s + p2);
            f32x4* __restrict__ dst4 = reinterpret_cast<f32x4*>(out + s + p2);
            // The first NFULL store instructions are full whatever s / end are (at most 15 units are skipped): ONE 1 KiB
            // store per LDS round trip and wave, then 64 idle cycles - no bursts.  Interleaved rounds on several boxes,
            // 27 x 4096 x 20 steps: 12.8-13.4 us/step against 13.6-14.7 for bursts of five stores (13.3-13.8 without the
            // pause, 14.3 with twice the pause: profiles/r02_pitch/pacing.txt).  Buffers that live in the Infinity Cache
            // and batches of many workgroup generations are faster in the plain form (host: launch_roll).
            constexpr unsigned NFULL = ((TU - 16) >> 1) / 64;
#pragma unroll
            for (unsigned c = 0; c < NFULL; ++c) {
                const f32x4 v = src4[c * 64 + lane];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dst4[c * 64 + lane] = v;                                // (non-temporal stores: no gain, profiles/r03_wide/ab_27_nt_placed.txt)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (unsigned q0 = NFULL * 64; q0 < NPMAX; q0 += 64) {
                const unsigned q = q0 + lane;
                if (q < npair) dst4[q] = src4[q];
            }
            if (((end - s - p2) & 1u) && lane == 63) out[end - 1] = img[end - 1];
        }
    };
    if (total > 0) compose(0);
    for (int t = 0; t < total; ++t) {
        if (t + 1 < total) compose(t + 1);
        stream(t);
    }
}

// ---------------------------------------------------------------------------
// Gather writer for small envs (N <= 9: the whole env is one span of 3 N^2 units): no LDS image.  Every unit of an env's
// [N][6N] block is ONE subtraction of two entries of the env's tables - p_j - p_i, 0 - (-v_i), or entry - 0 for the static
// units (the zeros of the table serve as the 0) - so a lane that stores 16 bytes needs four table reads and two packed
// subtractions.  Which entries, and where the 16 bytes go, depends only on the lane, the pass and the 8-byte phase of the
// env's first unit: a table of (operand offsets, destination, kind) per [phase][pass][lane], built once per workgroup in
// LDS, replaces the tile writer's index arithmetic, its exec-mask regions and all but two of its LDS round trips per env
// (write_obs_tiled at 9 agents: ~240 instructions and 11 dependent LDS waits per env, the bound of the 9-agent rollouts,
// profiles/r04_trace_ab.txt).  Same subtractions on the same operands as the other writers: bit-identical output.
// ---------------------------------------------------------------------------
constexpr int FG_WR_GATHER = 64;                                        // rollout_kernel's WR value for this writer
template <int NC> constexpr int gather_passes() { return (3 * NC * NC / 2 + 2 + 63) / 64; }
template <int NC> constexpr int gather_lut_units() { return 2 * gather_passes<NC>() * 64 * 2; }   // float2 units (16 B per entry)

// entry: x = a0 | b0 << 16, y = a1 | b1 << 16 (float2-unit offsets in the env's table buffer: unit = T[a] - T[b]),
//        z = destination unit inside the env | kind << 16 (2 = two units, 16-byte store; 1 = the first unit only; 0 = idle)
template <int NC>
FG_DEV void build_gather_lut(uint4* __restrict__ lut, int t, int nthreads) {
    constexpr int N = NC, ROWU = 3 * N, TU = ROWU * N, P = gather_passes<NC>() * 64;
    static_assert(N >= 2 && 5 * N < 65536, "gather writer: table offsets are 16-bit, the table's zeros are the 0 operand");
    auto operands = [&](int unit, unsigned& a, unsigned& b) {        // unit of the env block = T[a] - T[b]
        const int r = unit / ROWU, u = unit - r * ROWU;
        if (u == 0) { a = N; b = 4 * N + r; }                         // 0 - (-v_r): the operation the other writers do
        else if (u < N) { a = (u - 1 >= r) ? u : u - 1; b = r; }      // p_j - p_r, j != r in order
        else { a = u; b = N; }                                        // zeros | ideal shape | ideal velocity: entry - 0
    };
    for (int e = t; e < 2 * P; e += nthreads) {
        const int par = e / P, q = e - par * P;
        const int npair = (TU - par) >> 1;
        const bool tail = ((TU - par) & 1) != 0;                      // a unit behind the last pair
        int unit = -1, kind = 0;
        if (q < npair) { unit = 2 * q + par; kind = 2; }
        else if (par && q == npair) { unit = 0; kind = 1; }           // the unit in front of the first pair
        else if (tail && q == npair + par) { unit = TU - 1; kind = 1; }
        unsigned a0 = N, b0 = N, a1 = N, b1 = N;
        if (kind >= 1) operands(unit, a0, b0);
        if (kind == 2) operands(unit + 1, a1, b1);
        lut[e] = make_uint4(a0 | (b0 << 16), a1 | (b1 << 16), (unsigned)(unit < 0 ? 0 : unit) | ((unsigned)kind << 16), 0u);
    }
}

// The lane's table entries of one 8-byte phase, kept in registers across envs and steps: which phase an env block starts at
// changes only when the env pitch or the batch's slab is an odd number of units, so a writer wave reads its entries from LDS
// once per launch instead of once per env and step (one of the two LDS round trips in front of every env's stores).
template <int NC> struct GatherEntries {
    uint32_t ab0[gather_passes<NC>()], ab1[gather_passes<NC>()], dst[gather_passes<NC>()];   // a table entry's x, y, z (w is unused)
    int phase = -1;
};

// wave w of NW writes the envs w, w + NW, ... of the workgroup's El envs
template <int NC, int NW, int E>
FG_DEV void write_obs_gather(const float2* __restrict__ tables0, int env_stride, int w, const uint4* __restrict__ lut,
                             float2* __restrict__ out_env0, size_t unit0, size_t env_units, int El, GatherEntries<NC>& cache) {
    constexpr int PASSES = gather_passes<NC>(), P = PASSES * 64;
    const int lane = threadIdx.x & 63;
    for (int ee = w; ee < El; ee += NW) {
        const float2* __restrict__ T = tables0 + (size_t)ee * env_stride;
        const size_t first = unit0 + (size_t)ee * env_units;
        float2* __restrict__ out = out_env0 + (size_t)ee * env_units;
        const int phase = (int)(first & 1);
        if (phase != cache.phase) {                                  // wave-uniform
            const uint4* __restrict__ L = lut + phase * P + lane;
#pragma unroll
            for (int c = 0; c < PASSES; ++c) {
                const uint4 t = L[c * 64];
                cache.ab0[c] = t.x; cache.ab1[c] = t.y; cache.dst[c] = t.z;
            }
            cache.phase = phase;
        }
        f32x4 val[PASSES];
#pragma unroll
        for (int c = 0; c < PASSES; ++c) {
            const float2 A0 = T[cache.ab0[c] & 0xffffu], B0 = T[cache.ab0[c] >> 16], A1 = T[cache.ab1[c] & 0xffffu], B1 = T[cache.ab1[c] >> 16];
            val[c] = (f32x4){A0.x - B0.x, A0.y - B0.y, A1.x - B1.x, A1.y - B1.y};
        }
#pragma unroll
        for (int c = 0; c < PASSES; ++c) {
            const unsigned kind = cache.dst[c] >> 16, dst = cache.dst[c] & 0xffffu;
            if (kind == 2) *reinterpret_cast<f32x4*>(out + dst) = val[c];
            else if (kind == 1) out[dst] = make_float2(val[c].x, val[c].y);
        }
    }
}

// ---------------------------------------------------------------------------
// Span form of the gather writer: the E env blocks of a workgroup are ONE span of E * 3 N^2 units = E * 3 N^2 / 2 sixteen-byte
// pieces, and when that span starts on a 16-byte boundary (a full workgroup of a contiguous tensor: its base is then even a
// multiple of 64 bytes at 8 and 9 agents x 16 / 8 envs) store instruction p of the workgroup covers pieces [64 p, 64 p + 64):
// 1 KiB at a 1 KiB-aligned offset of the span, every 64-byte write request whole.  The per-env form above starts each env's
// instructions at the env's own base (1 944 bytes apart at 9 agents: 24 mod 64), so every instruction boundary splits a request
// in two - ~34.5 requests per env instead of 30.4, and the store stream is bound by requests in flight.
// Wave w of NW takes instructions w, w + NW, ...; which two units a lane stores in its t-th instruction, from which env's
// tables, never changes: the operand offsets (relative to env 0's tables, < 64 Ki units) are computed once per launch into an
// LDS table of one 8-byte entry per piece (it shares its place with the per-env form's table: a workgroup uses one of them).
// ---------------------------------------------------------------------------
template <int NC, int NW, int E, int CHMAX = 4> struct SpanGather {
    static constexpr int ENVU = 3 * NC * NC, PIECES = E * ENVU / 2, INSTR = (PIECES + 63) / 64, MINE = (INSTR + NW - 1) / NW;
    static_assert((E * ENVU) % 2 == 0, "an even number of units per workgroup span");
    // table in LDS, one entry per piece: x = a | b << 16 of its first unit, y = of its second (unit = T[a] - T[b], T = env 0's
    // tables); built once per launch by the writer waves themselves (each lane the entries it will read: no barrier needed)
    static FG_DEV void setup(uint2* __restrict__ lut, int w, int env_stride) {
        const int lane = threadIdx.x & 63;
        auto operands = [&](int unit) -> uint32_t {               // unit of the span = T[a] - T[b]
            const int e = unit / ENVU, q = unit - e * ENVU;
            const int r = q / (3 * NC), u = q - r * (3 * NC);
            uint32_t a, b;
            if (u == 0) { a = NC; b = 4 * NC + r; }                // 0 - (-v_r): the operation the other writers do
            else if (u < NC) { a = (u - 1 >= r) ? u : u - 1; b = r; }   // p_j - p_r, j != r in order
            else { a = u; b = NC; }                                // zeros | ideal shape | ideal velocity: entry - 0
            const uint32_t base = (uint32_t)(e * env_stride);
            return (a + base) | ((b + base) << 16);
        };
        for (int t = 0; t < MINE; ++t) {
            const int piece = (w + NW * t) * 64 + lane;
            if (piece < PIECES) lut[piece] = make_uint2(operands(2 * piece), operands(2 * piece + 1));
        }
    }
    // tables0: env 0's tables of this step's buffer; out: the workgroup's span (16-byte aligned)
    static FG_DEV void write(const uint2* __restrict__ lut, const float2* __restrict__ tables0, int w, float2* __restrict__ out) {
        const int lane = threadIdx.x & 63;
        f32x4* const out4 = reinterpret_cast<f32x4*>(out);
        constexpr int CH = MINE < CHMAX ? MINE : CHMAX;              // instructions whose operands are in flight together
#pragma unroll
        for (int t0 = 0; t0 < MINE; t0 += CH) {
            uint2 ent[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int piece = (w + NW * (t0 + c)) * 64 + lane;
                ent[c] = lut[(t0 + c < MINE && piece < PIECES) ? piece : 0];
            }
            f32x4 val[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const float2 A0 = tables0[ent[c].x & 0xffffu], B0 = tables0[ent[c].x >> 16];
                const float2 A1 = tables0[ent[c].y & 0xffffu], B1 = tables0[ent[c].y >> 16];
                val[c] = (f32x4){A0.x - B0.x, A0.y - B0.y, A1.x - B1.x, A1.y - B1.y};
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int piece = (w + NW * (t0 + c)) * 64 + lane;
                if (t0 + c < MINE && piece < PIECES) out4[piece] = val[c];
            }
        }
    }
};

}  // namespace fg

#endif  // FG_OBS_WRITERS_HPP_
