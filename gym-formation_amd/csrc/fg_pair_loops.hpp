// fg_pair_loops.hpp - All-pairs loops (contact force, reward pass) with packed fp32 math.
// Part of libformation_hip (gfx950); included by formation_hip.hip, one translation unit.
#ifndef FG_PAIR_LOOPS_HPP_
#define FG_PAIR_LOOPS_HPP_

#include "fg_common.hpp"

namespace fg {

// ---------------------------------------------------------------------------
// pair loops, two partner agents per iteration with packed fp32 math (v_pk_*_f32)
// ---------------------------------------------------------------------------
// World.step contact force on agent i (core.py:289-322): sum over j != i on PRE-step positions.
// NPC: the padded partner count when it is a small compile-time constant (<= 16), else 0.  With
// NPC every partner is fetched from LDS up front (one round trip instead of one per iteration:
// at small N the producer chain is latency-bound); the arithmetic and its order are the same.
template <int NPC = 0, int UNR = 2>
FG_DEV real2 contact_force_packed(const real* __restrict__ QX, const real* __restrict__ QY, int NP,
                                   int i, real2 p, real cf, real kmargin, real dmin, real cutoff2) {
    real fx = 0.0f, fy = 0.0f;
    const realx2 px = {p.x, p.x}, py = {p.y, p.y};
    const real inv_k = 1.0f / kmargin;
    auto add = [&](real dx, real dy, real d2) {
        // Hardware transcendentals (v_sqrt/v_exp/v_log/v_rcp, ~1 ulp).
        // Relative force error ~3e-7 (|f| <= 6) -> < 2e-8 on positions.
        // d2 == 0 for two distinct agents is kept: 0 * inf -> NaN as in core.py:312.
        const real d = hw_sqrt(d2);
        const real x = (dmin - d) * inv_k;
        const real pen = kmargin * (rmax(x, real(0)) + hw_log(1.0f + hw_exp(-rabs(x))));
        const real c = cf * pen * hw_rcp(d);
        fx += dx * c;
        fy += dy * c;
    };
    // Pairs inside the cutoff are rare (~1 % of all pairs at uniform density), but SOME lane of a wave has one in most
    // iterations, so a softplus evaluated inside the partner loop is walked through by the whole wave almost every
    // time.  The loop therefore only MARKS close partners (one bit each, 32 partners per chunk); the marked ones are
    // evaluated afterwards in ascending j - the summation order, and with it every bit of the result, is that of the
    // plain loop.  The wave now runs the softplus max-over-lanes(#contacts) times per chunk: ~2 instead of ~20 at 27 agents.
    // beyond the cutoff the softplus penetration is below fp32 resolution of the force: skipped
    // The mark word is built by shifting: m = 2 m + (close ? 1 : 0) is ONE v_addc behind the compare, so the first partner
    // of a chunk of n ends up in bit n - 1 and the marked partners are walked from the highest bit down (ascending j).
    auto mark = [&](uint32_t& m, realx2 qx, realx2 qy) {
        const realx2 dx = px - qx, dy = py - qy;
        const realx2 d2 = dx * dx + dy * dy;
        m = m + m + (d2.x < cutoff2 ? 1u : 0u);
        m = m + m + (d2.y < cutoff2 ? 1u : 0u);
    };
    auto flush = [&](int j0, int n, uint32_t m) {               // chunk of n partners starting at j0
        while (m) {
            const int hb = 31 - __builtin_clz(m);
            m &= ~(1u << hb);
            const int j = j0 + (n - 1 - hb);
            const realx2 qx = {QX[j], QX[j]}, qy = {QY[j], QY[j]};
            const realx2 dx = px - qx, dy = py - qy;
            const realx2 d2 = dx * dx + dy * dy;
            add(dx.x, dy.x, d2.x);
        }
    };
    if constexpr (NPC > 0 && NPC <= 16) {
        realx2 qx[NPC / 2], qy[NPC / 2];
#pragma unroll
        for (int h = 0; h < NPC / 2; ++h) {
            qx[h] = *reinterpret_cast<const realx2*>(QX + 2 * h);
            qy[h] = *reinterpret_cast<const realx2*>(QY + 2 * h);
        }
        uint32_t m = 0;
#pragma unroll
        for (int h = 0; h < NPC / 2; ++h) mark(m, qx[h], qy[h]);
        flush(0, NPC, m & ~(1u << (NPC - 1 - i)));
    } else {
        for (int j0 = 0; j0 < NP; j0 += 32) {
            const int jn = NP - j0 < 32 ? NP - j0 : 32;
            uint32_t m = 0;
#pragma unroll UNR
            for (int t = 0; t < jn; t += 2)
                mark(m, *reinterpret_cast<const realx2*>(QX + j0 + t), *reinterpret_cast<const realx2*>(QY + j0 + t));
            if ((unsigned)(i - j0) < (unsigned)jn) m &= ~(1u << (jn - 1 - (i - j0)));
            flush(j0, jn, m);
        }
    }
    return make_real2(fx, fy);
}

// The same sum for agents of different mass and size (FgParams.agent_props; no reference scenario has them): the
// contact distance of a pair is size_i + size_j (core.py:307) and agent i receives force_ratio = m_j / m_i times the
// pair's force (core.py:314-317: force_a = (m_b / m_a) f, force_b = -(m_a / m_b) f - for either role of i that is
// (m_j / m_i) * contact_force * (p_i - p_j) / d * penetration).  Plain loop, ascending j as the reference accumulates.
// FL[j] = partner j's flags (FG_AGENT_*): a pair needs both to collide (core.py:292-293); against an immovable partner
// the force is taken as it is, not scaled by the mass ratio (:319-321).  (An immovable agent's own force is never used.)
FG_DEV real2 contact_force_het(const real* __restrict__ QX, const real* __restrict__ QY, const real* __restrict__ MS,
                               const real* __restrict__ SZ, const real* __restrict__ FL, int N, int i, real2 p, real m_i,
                               real s_i, int flags_i, real cf, real kmargin) {
    real fx = 0.0f, fy = 0.0f;
    if (flags_i & (FG_AGENT_IMMOVABLE | FG_AGENT_NO_COLLIDE)) return make_real2(fx, fy);
    const real inv_k = 1.0f / kmargin;
    const real inv_m = 1.0f / m_i;
    const real far = (FG_F64 ? 40.0f : 18.0f) * kmargin;
    for (int j = 0; j < N; ++j) {
        const real dx = p.x - QX[j], dy = p.y - QY[j];
        const real d2 = dx * dx + dy * dy;
        const real dmin = s_i + SZ[j];
        const real cut = dmin + far;
        const int fj = (int)FL[j];
        if (j != i && d2 < cut * cut && !(fj & FG_AGENT_NO_COLLIDE)) {
            const real d = hw_sqrt(d2);
            const real x = (dmin - d) * inv_k;
            const real pen = kmargin * (rmax(x, real(0)) + hw_log(1.0f + hw_exp(-rabs(x))));
            const real ratio = (fj & FG_AGENT_IMMOVABLE) ? real(1) : MS[j] * inv_m;
            const real c = ratio * (cf * pen * hw_rcp(d));
            fx += dx * c;
            fy += dy * c;
        }
    }
    return make_real2(fx, fy);
}

// Scenario.is_collision with per-agent sizes (formation_hd_env.py:119-121): #{j != i : |p_j - p_i| < scale (size_i + size_j)}
FG_DEV int collision_count_het(const real* __restrict__ PX, const real* __restrict__ PY, const real* __restrict__ SZ,
                               int N, int i, real2 p, real s_i, real scale) {
    int c = 0;
    for (int j = 0; j < N; ++j) {
        const real dx = PX[j] - p.x, dy = PY[j] - p.y;
        const real t = scale * (s_i + SZ[j]);
        c += (j != i && dx * dx + dy * dy < t * t) ? 1 : 0;
    }
    return c;
}

// Scenario.reward inner pass for agent i / ideal point i (formation_hd_env.py:61-75):
//   rowmin = min_j |p~_i - s_j|^2,  colmin = min_j |p~_j - s_i|^2,  cnt = #{j != i : |p_j - p_i| < thr}
template <bool IDX, int NPC = 0, int UNR = 2>
FG_DEV void reward_pass_packed(const real* __restrict__ PX, const real* __restrict__ PY,
                               const real* __restrict__ SX, const real* __restrict__ SY, int NP,
                               real2 p, real ptx, real pty, real tx, real ty, real thr2,
                               real& rowmin, real& colmin, int& cnt, int& arg_lm, int& arg_ag) {
    const realx2 px = {p.x, p.x}, py = {p.y, p.y};
    const realx2 ptx2 = {ptx, ptx}, pty2 = {pty, pty}, tx2 = {tx, tx}, ty2 = {ty, ty};
    int c = -1;                                    // the self pair (distance 0) is counted below
    auto pair = [&](int j, realx2 qx, realx2 qy, realx2 sx, realx2 sy) {
        const realx2 cx = qx - px, cy = qy - py;
        const realx2 dc = cx * cx + cy * cy;
        c += (dc.x < thr2 ? 1 : 0) + (dc.y < thr2 ? 1 : 0);
        const realx2 rx = ptx2 - sx, ry = pty2 - sy;
        const realx2 dr = rx * rx + ry * ry;
        const realx2 ux = qx - tx2, uy = qy - ty2;
        const realx2 dq = ux * ux + uy * uy;
        if (IDX) {
            if (dr.x < rowmin) { rowmin = dr.x; arg_lm = j; }
            if (dr.y < rowmin) { rowmin = dr.y; arg_lm = j + 1; }
            if (dq.x < colmin) { colmin = dq.x; arg_ag = j; }
            if (dq.y < colmin) { colmin = dq.y; arg_ag = j + 1; }
        } else {
            rowmin = rmin(rmin(rowmin, dr.x), dr.y);
            colmin = rmin(rmin(colmin, dq.x), dq.y);
        }
    };
    if constexpr (NPC > 0 && NPC <= 16) {
        // up to 12 partners per fetch: 16 at once held 64 registers of operands and made every 16-agent rollout kernel spill
        // (128-VGPR budget); two fetches of 8 cost one more LDS round trip, same order of operations
        constexpr int CH = NPC <= 12 ? NPC / 2 : NPC / 4;            // realx2 pairs per fetch
        static_assert(NPC % 4 == 0, "padded partner counts are multiples of 4");
#pragma unroll
        for (int h0 = 0; h0 < NPC / 2; h0 += CH) {
            realx2 qx[CH], qy[CH], sx[CH], sy[CH];
#pragma unroll
            for (int h = 0; h < CH; ++h) {
                qx[h] = *reinterpret_cast<const realx2*>(PX + 2 * (h0 + h));
                qy[h] = *reinterpret_cast<const realx2*>(PY + 2 * (h0 + h));
                sx[h] = *reinterpret_cast<const realx2*>(SX + 2 * (h0 + h));
                sy[h] = *reinterpret_cast<const realx2*>(SY + 2 * (h0 + h));
            }
#pragma unroll
            for (int h = 0; h < CH; ++h) pair(2 * (h0 + h), qx[h], qy[h], sx[h], sy[h]);
        }
    } else {
#pragma unroll UNR
        for (int j = 0; j < NP; j += 2)
            pair(j, *reinterpret_cast<const realx2*>(PX + j), *reinterpret_cast<const realx2*>(PY + j),
                 *reinterpret_cast<const realx2*>(SX + j), *reinterpret_cast<const realx2*>(SY + j));
    }
    cnt = c + (thr2 > 0.0f ? 0 : 1);               // thr == 0: not even the self pair was counted
}

}  // namespace fg

#endif  // FG_PAIR_LOOPS_HPP_
