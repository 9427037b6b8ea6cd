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

}  // namespace fg

#endif  // FG_PAIR_LOOPS_HPP_
