// fg_policy_kernels.hpp - The reference's built-in demo controller on device: `ezpolicy`
// (formation_gym/__init__.py:19-47) expanded over a per-ary hierarchy by `get_action_BFS` (:49-99), the policy
// test.py:23 drives the env with.  Part of libformation_hip (gfx950); included by formation_hip.hip.
//
// The reference walks a Python queue of per-agent observation lists, level by level: at a level whose groups have
// n_cur members, every group is cut into `per` sub-groups of n_sub = n_cur / per members; the first agent of a
// sub-group is its leader (:64); the leader sees the sub-group CENTROIDS of its group (:68-71) and the centroids
// of the matching slices of the ideal shape (:73-74) as a `per`-agent formation problem, solves it with `ezpolicy`
// and hands the result x level (:78-79) down to its sub-group as that sub-group's target velocity (:91-94).
// Groups of one agent end the recursion: their target velocity is the action (:81-83).
//
// Everything the hierarchy reads is in observation row 0 of an env (relative positions p_j - p_0, the ideal shape,
// the ideal velocity): centroid DIFFERENCES do not depend on whose row they are taken from, and `ezpolicy` never
// uses the velocity it unpacks (:24).  Here one env is evaluated by a group of lanes:
//   bottom-up   per-level sums of relative positions and ideal points (each level adds `per` child sums),
//   all levels  one lane per (level, sub-group) PROBLEM - N + N/per + ... + per of them, dealt over the lanes in one
//               flattened pass - runs `ezpolicy` on its group's `per` child sums (registers, PER is a compile-time
//               constant).  Only the last line of `ezpolicy` depends on the level above (act += w * target velocity,
//               :43-46), so the pass stores (act, w) and
//   top-down    a two-instruction combine per level hands the target velocities down (x level, :78-79).
// The same device function serves the stand-alone launch (`fg_policy_bfs`, tables filled from the observation) and
// the closed-loop rollout kernels (tables filled from the workgroup's LDS state), so both give the same bits.
#ifndef FG_POLICY_KERNELS_HPP_
#define FG_POLICY_KERNELS_HPP_

#include "fg_common.hpp"

namespace fg {

// LDS of one env for the controller, in float2 units:
//   R[N] relative positions | S[N] ideal shape | SR[N] SS[N] level sums (levels 1 .. L-1 back to back) |
//   BA[2N] per problem: act, then in place the sub-group's target velocity (levels top .. leaf back to back; the leaf
//   level's N entries are the actions) | BW[2N floats] per problem: the weight of the level above (1 or 0.3)
__host__ __device__ constexpr int policy_block_units(int n) { return 7 * n; }

struct BlockSync { FG_DEV void operator()() const { __syncthreads(); } };
// the lanes of one env sit in ONE wave: its LDS operations complete in order, only the compiler has to be held back
struct WaveSync {
    FG_DEV void operator()() const {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
};

// `ezpolicy` (__init__.py:19-47) for sub-group `i` of the group whose `per` child sums start at cR/cS[base].
//   cen[k]   centroid of sub-group k (relative positions), tgt[k] centroid of its slice of the ideal shape
//   ideal    = tgt - mean(tgt)                                                            (:26-28)
//   cur      = [cen[k] - cen[i] for k != i, index order] + [(0, 0)], minus its mean      (:31-33; me = last row)
//   marks sorted by distance to me (:35); the first mark whose closest agent (np.argmin: first minimum, me is
//   LAST, so me wins only strictly) is me, else the last mark of the order (:36-40); act = clip(0.5 (mark - me)) (:39)
//   done = ||ideal - cur||_F < 0.01 (:42) -> act += ideal_vel (x 0.3 while not done) (:43-46)
//   Returns act BEFORE the ideal-velocity term and, in `w`, that term's weight.
template <int PER>
FG_DEV float2 ez_policy(const float2* __restrict__ cR, const float2* __restrict__ cS, int base, int i,
                        float inv_sub, float inv_per, float& w) {
    float2 cen[PER], ideal[PER], cur[PER];
    float2 tsum = make_float2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const float2 r = cR[base + k], s = cS[base + k];
        cen[k] = make_float2(r.x * inv_sub, r.y * inv_sub);
        ideal[k] = make_float2(s.x * inv_sub, s.y * inv_sub);
        tsum.x += ideal[k].x; tsum.y += ideal[k].y;
    }
    const float2 tmean = make_float2(tsum.x * inv_per, tsum.y * inv_per);
    float2 ci = cen[0];                                    // value selects: no dynamic register indexing
#pragma unroll
    for (int k = 1; k < PER; ++k) if (k == i) ci = cen[k];
    float2 msum = make_float2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        ideal[k].x -= tmean.x; ideal[k].y -= tmean.y;
        cur[k] = make_float2(cen[k].x - ci.x, cen[k].y - ci.y);
        if (k != i) { msum.x += cur[k].x; msum.y += cur[k].y; }
    }
    const float2 m = make_float2(msum.x * inv_per, msum.y * inv_per);      // mean over the others and (0, 0)
    const float2 me = make_float2(0.f - m.x, 0.f - m.y);
#pragma unroll
    for (int k = 0; k < PER; ++k) { cur[k].x -= m.x; cur[k].y -= m.y; }    // entry i is not an "other": unused
    int best = -1, far = 0;
    float best_d = INFINITY, far_d = -1.0f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const float ex = me.x - ideal[k].x, ey = me.y - ideal[k].y;
        const float dme = ex * ex + ey * ey;
        float dmin = INFINITY;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const float fx = cur[j].x - ideal[k].x, fy = cur[j].y - ideal[k].y;
            const float d = fx * fx + fy * fy;
            if (j != i) dmin = fminf(dmin, d);
        }
        if (dme < dmin && dme < best_d) { best = k; best_d = dme; }        // stable order: lower index first
        if (dme >= far_d) { far = k; far_d = dme; }                        // last of the order: higher index last
    }
    const int pick = best >= 0 ? best : far;
    float2 mark = ideal[0];
#pragma unroll
    for (int k = 1; k < PER; ++k) if (k == pick) mark = ideal[k];
    float2 act = make_float2(fminf(fmaxf(0.5f * (mark.x - me.x), -1.0f), 1.0f),
                             fminf(fmaxf(0.5f * (mark.y - me.y), -1.0f), 1.0f));
    // row s of `cur` is the s-th OTHER sub-group (index order, i skipped) for s < PER-1 and me for s = PER-1,
    // compared with ideal row s (:42 subtracts the arrays as they are)
    float nsq;
    {
        const float dx = ideal[PER - 1].x - me.x, dy = ideal[PER - 1].y - me.y;
        nsq = dx * dx + dy * dy;
    }
#pragma unroll
    for (int s = 0; s < PER - 1; ++s) {
        const float2 c = (s >= i) ? cur[s + 1] : cur[s];
        const float dx = ideal[s].x - c.x, dy = ideal[s].y - c.y;
        nsq += dx * dx + dy * dy;
    }
    w = (nsq < 1.0e-4f) ? 1.0f : 0.3f;
    return act;
}

// The hierarchy for one env.  `tab` = this env's policy_block_units(N) float2 of LDS with R and S filled and
// published; `lane` / `lanes`: this thread's index among the threads that share the env.  Every one of them must
// call (the level loops synchronise).  Returns the N actions (LDS, published).
// LCT: the number of levels when the caller knows it at compile time (the pipelined rollout kernels: N = PER^LCT is a template
// parameter) - the level loops unroll and the per-lane search for a problem's level becomes a select chain: closed loop 9 x 4096
// 2.52 -> 2.01 us/step, 4 x 16384 2.39 -> 1.77, 8 x 65536 21.7 -> 18.4, 9 x 65536 29.8 -> 27.7 (profiles/r05_policy_ct_ab.txt);
// 0: pl.L at run time (the stand-alone controller launches).  The operations and their order are the same.
template <int PER, int LCT = 0, class Sync>
FG_DEV const float2* bfs_policy_env(float2* __restrict__ tab, int N, const FgPolicyLevels& pl, float2 iv,
                                    int lane, int lanes, Sync sync) {
    const int L = LCT > 0 ? LCT : pl.L;
    float2* const R = tab;
    float2* const S = tab + N;
    float2* const SR = tab + 2 * N;
    float2* const SS = tab + 3 * N;
    // ---- bottom-up: level l holds the sums over per^l consecutive agents ------------
    const float2* srcR = R;
    const float2* srcS = S;
    int n_l = N, off = 0;
    for (int l = 1; l < L; ++l) {
        n_l /= PER;
        for (int j = lane; j < n_l; j += lanes) {
            float2 a = srcR[j * PER], b = srcS[j * PER];
#pragma unroll
            for (int c = 1; c < PER; ++c) {
                const float2 x = srcR[j * PER + c], y = srcS[j * PER + c];
                a.x += x.x; a.y += x.y; b.x += y.x; b.y += y.y;
            }
            SR[off + j] = a; SS[off + j] = b;
        }
        sync();
        srcR = SR + off; srcS = SS + off;
        off += n_l;
    }
    // ---- every (level, sub-group) problem, flattened: levels top .. leaf back to back -------------------
    float2* const BA = tab + 4 * N;
    float* const BW = reinterpret_cast<float*>(tab + 6 * N);
    int P = 0;
    for (int lev = L, n = PER; lev >= 1; --lev, n *= PER) P += n;
    for (int p = lane; p < P; p += lanes) {
        int lev = L, start = 0, n = PER, off_l = off - PER;   // level of problem p: its first index, its size,
        while (p >= start + n) {                                  // where its child level starts in SR / SS
            start += n; n *= PER; off_l -= n; --lev;
        }
        const int sg = p - start, l = lev - 1;
        const int g = sg / PER, i = sg - g * PER;
        float w;
        // pl.inv_sub[l] by value selects: `l` differs between lanes, and a per-lane index into the kernel-argument struct
        // would make the compiler copy the table into scratch memory (every policy kernel carried a 16-byte private segment)
        float inv_sub = pl.inv_sub[0];
#pragma unroll
        for (int t = 1; t < FG_POLICY_MAX_LEVELS; ++t) inv_sub = (t == l) ? pl.inv_sub[t] : inv_sub;
        BA[p] = ez_policy<PER>(l ? SR + off_l : R, l ? SS + off_l : S, g * PER, i, inv_sub, pl.inv_per, w);
        BW[p] = w;
    }
    sync();
    // ---- top-down: target velocity of a sub-group = (act + w * target velocity of its group) x level (:43-46, :78-79)
    int start = 0, n = PER, pstart = 0;
    for (int lev = L; lev >= 1; --lev) {
        const float flev = (float)lev;
        for (int sg = lane; sg < n; sg += lanes) {
            // value select, never a pointer select (that puts `iv` into scratch memory); at the top level the entry read is
            // a valid one of this table whose value is not used
            const float2 up = BA[pstart + sg / PER];
            const bool top = lev == L;
            const float2 tv = make_float2(top ? iv.x : up.x, top ? iv.y : up.y);
            float2 a = BA[start + sg];
            const float w = BW[start + sg];
            a.x += tv.x * w; a.y += tv.y * w;
            BA[start + sg] = make_float2(a.x * flev, a.y * flev);
        }
        sync();
        pstart = start; start += n; n *= PER;
    }
    return BA + pstart;                       // the leaf level: N actions
}

// The same hierarchy for ONE LANE that holds a whole env in registers (fg_hd_lane_kernel.hpp: 3 and 4 agents): N, PER and with
// them the number of levels are compile-time constants, every table below is a register array with static indices.  The
// operations and their order are bfs_policy_env's - level sums child by child, `ez_policy` per (level, sub-group), the
// top-down combine `(act + w * target velocity) * level` - so the two give the same bits.
template <int N, int PER> constexpr int policy_levels_ct() {
    int l = 0, n = 1;
    while (n < N) { n *= PER; ++l; }
    return l;
}
template <int N, int PER>
FG_DEV void bfs_policy_lane(const float2 (&R)[N], const float2 (&S)[N], const FgPolicyLevels& pl, float2 iv, float2 (&act)[N]) {
    constexpr int L = policy_levels_ct<N, PER>();
    static_assert(L >= 1 && L <= 3, "one env per lane: a handful of agents");
    float2 SR[L][N], SS[L][N];                          // level l: sums over PER^l consecutive agents (level 0: R, S)
#pragma unroll
    for (int j = 0; j < N; ++j) { SR[0][j] = R[j]; SS[0][j] = S[j]; }
    int n_l = N;
#pragma unroll
    for (int l = 1; l < L; ++l) {
        n_l /= PER;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (j < n_l) {
                float2 a = SR[l - 1][j * PER < N ? j * PER : 0], b = SS[l - 1][j * PER < N ? j * PER : 0];
#pragma unroll
                for (int c = 1; c < PER; ++c) {
                    const float2 x = SR[l - 1][j * PER + c < N ? j * PER + c : 0], y = SS[l - 1][j * PER + c < N ? j * PER + c : 0];
                    a.x += x.x; a.y += x.y; b.x += y.x; b.y += y.y;
                }
                SR[l][j] = a; SS[l][j] = b;
            }
        }
    }
    float2 BA[L][N];                                    // per level (index lev - 1) and sub-group: act, then the target velocity
    float BW[L][N];
    int n = N;
#pragma unroll
    for (int l = 0; l < L; ++l) {                       // the problems of level lev = l + 1 read the child sums of level l
#pragma unroll
        for (int sg = 0; sg < N; ++sg) {
            if (sg < n) {
                const int g = sg / PER, i = sg - g * PER;
                float w;
                BA[l][sg] = ez_policy<PER>(SR[l], SS[l], g * PER, i, pl.inv_sub[l], pl.inv_per, w);
                BW[l][sg] = w;
            }
        }
        n /= PER;
    }
    n = PER;
#pragma unroll
    for (int lev = L; lev >= 1; --lev) {                // top-down (:43-46, :78-79)
        const float flev = (float)lev;
#pragma unroll
        for (int sg = 0; sg < N; ++sg) {
            if (sg < n) {
                const float2 up = BA[lev < L ? lev : L - 1][sg / PER];
                const float2 tv = make_float2(lev == L ? iv.x : up.x, lev == L ? iv.y : up.y);
                float2 a = BA[lev - 1][sg];
                const float w = BW[lev - 1][sg];
                a.x += tv.x * w; a.y += tv.y * w;
                BA[lev - 1][sg] = make_float2(a.x * flev, a.y * flev);
            }
        }
        n *= PER;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) act[j] = BA[0][j];
}

// Stand-alone launch: one env per `lpe` lanes (power of two >= min(N, 256)), tables from observation row 0.
template <int PER>
__global__ __launch_bounds__(256) void policy_bfs_kernel(const int B, const int N, const int lpe, const FgPolicyLevels pl,
                                                         const float* __restrict__ obs, const long long env_stride,
                                                         float* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int tid = threadIdx.x;
    const int E = 256 / lpe;
    const int e = tid / lpe, lane = tid - e * lpe;
    const int b = blockIdx.x * E + e;
    const bool ok = b < B;
    float2* const tab = smem + e * policy_block_units(N);
    float2 iv = make_float2(0.f, 0.f);
    const float2* row0 = reinterpret_cast<const float2*>(obs + (size_t)(ok ? b : 0) * (size_t)env_stride);
    for (int a = lane; a < N; a += lpe) {
        // unit a of row 0 is p_a - p_0 (unit 0 is agent 0's velocity); units 2N-1 .. 3N-2 are the ideal shape
        tab[a] = (ok && a) ? row0[a] : make_float2(0.f, 0.f);
        tab[N + a] = ok ? row0[2 * N - 1 + a] : make_float2(0.f, 0.f);
    }
    if (ok) iv = row0[3 * N - 1];
    __syncthreads();
    const float2* res = bfs_policy_env<PER>(tab, N, pl, iv, lane, lpe, BlockSync());
    if (ok)
        for (int a = lane; a < N; a += lpe) reinterpret_cast<float2*>(act)[(size_t)b * N + a] = res[a];
}

// The same controller straight from the simulator state (no observation needed): R[a] = p_a - p_0 is the very
// subtraction the observation writers perform, so the actions equal `policy_bfs_kernel` on the written rows bit for bit.
template <int PER>
__global__ __launch_bounds__(256) void policy_state_kernel(const int B, const int N, const int lpe, const FgPolicyLevels pl,
                                                           const float* __restrict__ px, const float* __restrict__ py,
                                                           const float* __restrict__ shape, const float* __restrict__ ivel,
                                                           float* __restrict__ act) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int tid = threadIdx.x;
    const int E = 256 / lpe;
    const int e = tid / lpe, lane = tid - e * lpe;
    const int b = blockIdx.x * E + e;
    const bool ok = b < B;
    float2* const tab = smem + e * policy_block_units(N);
    float2 iv = make_float2(0.f, 0.f);
    const size_t s0 = (size_t)(ok ? b : 0) * N;
    const float x0 = ok ? px[s0] : 0.f, y0 = ok ? py[s0] : 0.f;
    for (int a = lane; a < N; a += lpe) {
        tab[a] = ok ? make_float2(px[s0 + a] - x0, py[s0 + a] - y0) : make_float2(0.f, 0.f);
        tab[N + a] = ok ? reinterpret_cast<const float2*>(shape)[s0 + a] : make_float2(0.f, 0.f);
    }
    if (ok) iv = reinterpret_cast<const float2*>(ivel)[b];
    __syncthreads();
    const float2* res = bfs_policy_env<PER>(tab, N, pl, iv, lane, lpe, BlockSync());
    if (ok)
        for (int a = lane; a < N; a += lpe) reinterpret_cast<float2*>(act)[(size_t)b * N + a] = res[a];
}

}  // namespace fg

#endif  // FG_POLICY_KERNELS_HPP_
