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
//            memory, streamed by register-cached row writers or LDS tiles
//            (fg_obs_writers.hpp).
// Rollout launches and single steps at N = 81 / 243 run the same phases split over
// producer and writer waves (fg_rollout_kernels.hpp).
// Observation bytes (24 N^2 per env) dominate traffic; everything else is 53 N + 16.
// There is no dense contraction here, hence no MFMA: the kernel is HBM-store bound.
//
// Source layout: fg_common.hpp (arguments, LDS layout, reductions, RNG, World options),
// fg_pair_loops.hpp, fg_obs_writers.hpp, fg_step_kernel.hpp, fg_rollout_kernels.hpp,
// fg_aux_kernels.hpp (resets, landmark scenarios); this file holds the host side: variant
// tables, dispatch and the extern "C" entry points.

#include <climits>
#include <cstdlib>
#include "fg_common.hpp"
#include "fg_pair_loops.hpp"
#include "fg_obs_writers.hpp"
#include "fg_step_kernel.hpp"
#include "fg_rollout_kernels.hpp"
#include "fg_aux_kernels.hpp"

namespace fg {

// ---------------------------------------------------------------------------
// host side: geometry selection, validation, launches
// ---------------------------------------------------------------------------
static thread_local char g_err[256] = "";

static int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

// Tuning switches (profiles/README.md) are environment variables read ONCE per call site: a launch
// costs no getenv, and a process sees one consistent configuration.
static int read_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#define FG_OVERRIDE(var, name) \
    do { static const int fg_o_ = read_env_int(name, INT_MIN); if (fg_o_ != INT_MIN) (var) = fg_o_; } while (0)
struct GeomOverride { int t, e; bool set; };

struct Geometry { int G, T, E, lds; };

// Kernels that need more than the default 64 KiB of dynamic LDS opt in once per (kernel, device): the
// attribute belongs to the device the function is loaded on, and a process may drive several GPUs.
static void raise_lds_limit(const void* fn, int lds, unsigned long long* done_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!((*done_mask >> dev) & 1ull)) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        *done_mask |= 1ull << dev;
    }
}

static int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

template <int NC, int G, int T, int E, bool IDX, int WR, bool OPTS>
static hipError_t launch_v(const Args& a, int grid, int lds, hipStream_t st) {
    hipLaunchKernelGGL((step_kernel<NC, G, T, E, IDX, WR, OPTS>), dim3(grid), dim3(T), lds, st,
                       a.B, a.N, (const float*)a.px, (const float*)a.py, (const float*)a.vx, (const float*)a.vy,
                       (const float*)a.shape, (const float*)a.ivel, (const int32_t*)a.step, a);
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
    FG_VARIANT_W(9, 16, 128, 4, 10), FG_VARIANT_W(9, 16, 64, 4, 10), FG_VARIANT_W(9, 16, 128, 8, 10), FG_VARIANT_W(9, 16, 256, 16, 10),
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
    if (N == 9 && B >= 16384) { want_t = 128; want_e = 8; }
    {   // "T,E", parsed once
        static const GeomOverride go = [] { GeomOverride g = {0, 0, false};
                                            if (const char* s = getenv("FG_GEOM")) g.set = sscanf(s, "%d,%d", &g.t, &g.e) == 2;
                                            return g; }();
        if (go.set) { want_t = go.t; want_e = go.e; }
    }
    FG_OVERRIDE(want_flat, "FG_FLAT");
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
    FG_OVERRIDE(tw, "FG_TW");
    if (a.K == 1) {
        // env batches per workgroup: enough to overlap batch g+1's pair loops with batch g's store
        // stream, few enough to keep every CU busy (MI355X sweep, profiles/README.md)
        const int E = (N == 243 && tw == 128) ? 2 : ((N == 81 && tw == 512) ? 8 : 4);
        const int batches = (B + E - 1) / E;
        int target = 256;                                  // workgroups in the grid: one per CU
        FG_OVERRIDE(target, "FG_STEPWG");
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
            static unsigned long long raised = 0;                                                        \
            raise_lds_limit((const void*)&rollout_kernel_wide<NCV, AV, EV, TWV>, lds, &raised); }         \
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
    if (B == 0) return FG_OK;                         // an empty batch is a no-op (e.g. a rank that owns no envs)
    if (B < 0) return fail(FG_ERR_BAD_ARG, "B must be >= 0%s");
    if (N < 3 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "formation_hd_env needs 3 <= N <= 1024%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !act || !ideal_shape || !ideal_vel || !step || !obs || !reward)
        return fail(FG_ERR_BAD_ARG, "fg_step_hd: a required pointer is NULL%s");
    if (((uintptr_t)obs & 15u) || ((uintptr_t)act & 7u) || ((uintptr_t)ideal_shape & 7u) || ((uintptr_t)ideal_vel & 7u))
        return fail(FG_ERR_ALIGNMENT, "obs must be 16-byte, act/ideal_shape/ideal_vel 8-byte aligned%s");
    Args a; memset(&a, 0, sizeof(a));
    a.p = *params; a.B = B; a.N = N; a.inv_n = 1.0f / (float)N; a.K = 1; a.obs_every = 1; a.do_phys = 1; a.do_post = 1;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y; a.act = act;
    a.shape = ideal_shape; a.ivel = ideal_vel; a.step = step;
    a.obs = obs; a.rew = reward; a.indiv = indiv_reward; a.done = done;
    a.near_lm = near_lm; a.near_ag = near_ag; a.hd_idx = hd_idx;
    {   // 243 agents: pipeline over env batches inside the launch (no index outputs, no World options):
        // 1.85-2.2 ms vs 2.1-2.3 ms at 243 x 8192.  At 81 agents the plain kernel is as fast or faster
        // (74 vs 80 us at 81 x 2048 on the same box); FG_PIPE81=1 selects the pipelined one.
        int nopipe = 0; FG_OVERRIDE(nopipe, "FG_NOPIPE");
        int pipe81 = 0; FG_OVERRIDE(pipe81, "FG_PIPE81");
        const bool opts = a.p.num_walls > 0 || a.p.u_noise > 0.f || a.p.max_speed > 0.f || a.p.accel > 0.f;
        if ((N == 243 || (N == 81 && pipe81)) && !nopipe && !opts && !near_lm && !near_ag && !hd_idx)
            return launch_wide(a, (hipStream_t)stream);
    }
    return launch_step(a, (hipStream_t)stream);
}

int fg_physics_step(const FgParams* params, int B, int N,
                    float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                    const float* act, void* stream) {
    int rc = check_params(params);
    if (rc) return rc;
    if (B == 0) return FG_OK;                         // an empty batch is a no-op (e.g. a rank that owns no envs)
    if (B < 0) return fail(FG_ERR_BAD_ARG, "B must be >= 0%s");
    if (N < 2 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "N must be in [2, 1024]%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !act)
        return fail(FG_ERR_BAD_ARG, "fg_physics_step: a required pointer is NULL%s");
    if ((uintptr_t)act & 7u) return fail(FG_ERR_ALIGNMENT, "act must be 8-byte aligned%s");
    Args a; memset(&a, 0, sizeof(a));
    a.p = *params; a.p.auto_reset = 0; a.B = B; a.N = N; a.inv_n = 1.0f / (float)N; a.K = 1; a.obs_every = 1; a.do_phys = 1; a.do_post = 0;
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
    if (B == 0) return FG_OK;                         // an empty batch is a no-op (e.g. a rank that owns no envs)
    if (B < 0) return fail(FG_ERR_BAD_ARG, "B must be >= 0%s");
    if (N < 3 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "formation_hd_env needs 3 <= N <= 1024%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !ideal_shape || !ideal_vel)
        return fail(FG_ERR_BAD_ARG, "fg_observe_hd: a required pointer is NULL%s");
    if (!obs && !reward) return fail(FG_ERR_BAD_ARG, "fg_observe_hd: obs and reward both NULL%s");
    if (((uintptr_t)obs & 15u) || ((uintptr_t)ideal_shape & 7u) || ((uintptr_t)ideal_vel & 7u))
        return fail(FG_ERR_ALIGNMENT, "obs must be 16-byte, ideal_shape/ideal_vel 8-byte aligned%s");
    Args a; memset(&a, 0, sizeof(a));
    a.p = *params; a.p.auto_reset = 0; a.B = B; a.N = N; a.inv_n = 1.0f / (float)N; a.K = 1; a.obs_every = 1; a.do_phys = 0; a.do_post = 1;
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
    if (B == 0 || K == 0) return FG_OK;               // empty batch / zero steps: nothing to do
    if (B < 0 || K < 0) return fail(FG_ERR_BAD_ARG, "B and K must be >= 0%s");
    if (N < 3 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "formation_hd_env needs 3 <= N <= 1024%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !act_seq || !ideal_shape || !ideal_vel || !step || !reward_seq)
        return fail(FG_ERR_BAD_ARG, "fg_rollout_hd: a required pointer is NULL%s");
    if (((uintptr_t)obs_seq & 15u) || ((uintptr_t)act_seq & 7u) || ((uintptr_t)ideal_shape & 7u) || ((uintptr_t)ideal_vel & 7u))
        return fail(FG_ERR_ALIGNMENT, "obs_seq must be 16-byte, act_seq/ideal_shape/ideal_vel 8-byte aligned%s");
    Args a; memset(&a, 0, sizeof(a));
    a.p = *params; a.B = B; a.N = N; a.inv_n = 1.0f / (float)N; a.K = K; a.obs_every = obs_every < 1 ? 1 : obs_every;
    a.do_phys = 1; a.do_post = 1;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y; a.act = act_seq;
    a.shape = ideal_shape; a.ivel = ideal_vel; a.step = step;
    a.obs = obs_seq; a.rew = reward_seq; a.indiv = indiv_seq; a.done = done_seq;
    // K >= 2 at the specialised small N: producer / writer pipelined kernel
    if (FG_PROBES) { FG_OVERRIDE(a.probe, "FG_PROBE"); }
    int nopipe = 0; FG_OVERRIDE(nopipe, "FG_NOPIPE");
    if (K >= 2 && !nopipe && (N == 81 || N == 243)) return launch_wide(a, (hipStream_t)stream);
    if (K >= 2 && !nopipe && (N == 27 || N == 9 || N == 3)) {
        int tw = 256;                      // defaults from the MI355X sweep (profiles/README.md)
        FG_OVERRIDE(tw, "FG_TW");
        hipStream_t st = (hipStream_t)stream;
        hipError_t err = hipSuccess;
        int wr = 10;
        FG_OVERRIDE(wr, "FG_ROLLWR");
#define FG_ROLL(NCV, GV, TPV, TWV, EV, WRV)                                                              \
        {   const int grid = (B + (EV) - 1) / (EV);                                                      \
            int lds = (EV) * roll_block_floats(NCV) * (int)sizeof(float);                                \
            if ((WRV) > 0) lds += 2 * ((TWV) / 64) * ((3 * (NCV) * ((WRV) - 1) + 3) & ~1) * (int)sizeof(float2); \
            hipLaunchKernelGGL((rollout_kernel<NCV, GV, TPV, TWV, EV, WRV>), dim3(grid), dim3((TPV) + (TWV)), lds, st, a); \
            err = hipGetLastError(); }
        if (N == 27) {
            int re = 16;                   // 16 envs per workgroup: 8 producer + 4 writer waves, one workgroup per CU
            FG_OVERRIDE(re, "FG_ROLLE");
            int share = 0;                 // tuning: producers join the observation stream through an LDS tile counter
            FG_OVERRIDE(share, "FG_SHARE");
            if (re == 2) { if (wr == 10) { if (tw == 64) FG_ROLL(27, 32, 64, 64, 2, 10) else FG_ROLL(27, 32, 64, 128, 2, 10) }
                           else { if (tw == 64) FG_ROLL(27, 32, 64, 64, 2, 0) else FG_ROLL(27, 32, 64, 128, 2, 0) } }
            else if (re == 16 && share) {
                const int grid = (B + 15) / 16;
                const int tunits = (3 * 27 * 9 + 3) & ~1;
                const int lds = 16 * roll_block_floats(27) * (int)sizeof(float) + 16 +
                                ((tw / 64) * 2 + 8) * tunits * (int)sizeof(float2);
                if (tw == 256) {
                    static unsigned long long raised = 0;
                    raise_lds_limit((const void*)&rollout_kernel<27, 32, 512, 256, 16, 10, true>, lds, &raised);
                    hipLaunchKernelGGL((rollout_kernel<27, 32, 512, 256, 16, 10, true>), dim3(grid), dim3(768), lds, st, a);
                } else {
                    static unsigned long long raised2 = 0;
                    raise_lds_limit((const void*)&rollout_kernel<27, 32, 512, 128, 16, 10, true>, lds, &raised2);
                    hipLaunchKernelGGL((rollout_kernel<27, 32, 512, 128, 16, 10, true>), dim3(grid), dim3(640), lds, st, a);
                }
                err = hipGetLastError();
            }
            else if (re == 16) { if (wr == 10) { if (tw == 256) FG_ROLL(27, 32, 512, 256, 16, 10) else FG_ROLL(27, 32, 512, 512, 16, 10) }
                                 else { if (tw == 256) FG_ROLL(27, 32, 512, 256, 16, 0) else FG_ROLL(27, 32, 512, 512, 16, 0) } }
            else if (re == 8 && tw == 512) { if (wr == 10) FG_ROLL(27, 32, 256, 512, 8, 10) else FG_ROLL(27, 32, 256, 512, 8, 0) }
            else if (re == 8) { if (wr == 10) { if (tw == 128) FG_ROLL(27, 32, 256, 128, 8, 10) else FG_ROLL(27, 32, 256, 256, 8, 10) }
                                else { if (tw == 128) FG_ROLL(27, 32, 256, 128, 8, 0) else FG_ROLL(27, 32, 256, 256, 8, 0) } }
            else
            if (wr == 10) { if (tw == 64) FG_ROLL(27, 32, 128, 64, 4, 10) else if (tw == 256) FG_ROLL(27, 32, 128, 256, 4, 10) else FG_ROLL(27, 32, 128, 128, 4, 10) }
            else { if (tw == 64) FG_ROLL(27, 32, 128, 64, 4, 0) else if (tw == 256) FG_ROLL(27, 32, 128, 256, 4, 0) else FG_ROLL(27, 32, 128, 128, 4, 0) }
        }
        else if (N == 9) {
            // MI355X sweep (profiles/README.md): a batch of <= 4096 envs is bound by the producers'
            // dependent chain (1.8 us/step) and wants many small workgroups; larger batches are
            // store-bound and want 16-env workgroups (whole 128-byte lines per workgroup) with the
            // LDS-tiled writer.  FG_ROLL9 overrides (tuning aid).
            int v9 = B >= 8192 ? 4 : B > 4096 ? 6 : 5;
            FG_OVERRIDE(v9, "FG_ROLL9");
            if (v9 == 1) FG_ROLL(9, 16, 64, 64, 4, 10)
            else if (v9 == 4) FG_ROLL(9, 16, 256, 256, 16, 10)
            else if (v9 == 6) FG_ROLL(9, 16, 128, 128, 8, 10)
            else if (v9 == 0) FG_ROLL(9, 16, 64, 64, 4, 0)
            else FG_ROLL(9, 16, 64, 128, 4, 0)
        }
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
    if (B == 0) return FG_OK;                         // an empty batch is a no-op (e.g. a rank that owns no envs)
    if (B < 0) return fail(FG_ERR_BAD_ARG, "B must be >= 0%s");
    Geometry g;
    if (!geometry_for(N, &g)) return fail(FG_ERR_UNSUPPORTED_N, "N must be in [2, 1024]%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !ideal_shape || !ideal_vel)
        return fail(FG_ERR_BAD_ARG, "fg_reset_hd: a required pointer is NULL%s");
    Args a; memset(&a, 0, sizeof(a));
    a.p = *params; a.B = B; a.N = N; a.inv_n = 1.0f / (float)N;
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
    if (B == 0) return FG_OK;
    if (B < 0 || L <= 0 || M < 0) return fail(FG_ERR_BAD_ARG, "B >= 0, L > 0, M >= 0 required%s");
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
    if (B == 0) return FG_OK;                         // an empty batch is a no-op (e.g. a rank that owns no envs)
    if (B < 0) return fail(FG_ERR_BAD_ARG, "B must be >= 0%s");
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

int fg_decode_actions(int mode, int64_t count, void* action, float* u_out, void* stream) {
    if (mode != FG_ACT_ONEHOT5 && mode != FG_ACT_INDEX && mode != FG_ACT_ARGMAX)
        return fail(FG_ERR_BAD_ARG, "fg_decode_actions: unknown mode%s");
    if (count == 0) return FG_OK;
    if (count < 0 || count > ((int64_t)1 << 38)) return fail(FG_ERR_BAD_ARG, "fg_decode_actions: count out of range%s");
    if (!action || !u_out) return fail(FG_ERR_BAD_ARG, "fg_decode_actions: a required pointer is NULL%s");
    if (((uintptr_t)u_out & 7) || (mode == FG_ACT_ARGMAX && ((uintptr_t)action & 7)))
        return fail(FG_ERR_ALIGNMENT, "fg_decode_actions: buffers must be 8-byte aligned%s");
    const unsigned grid = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(decode_actions_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mode, count, action,
                       reinterpret_cast<float2*>(u_out));
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "decode launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

}  // extern "C"
