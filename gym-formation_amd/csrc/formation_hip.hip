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
// fg_policy_kernels.hpp (the reference's demo controller), fg_aux_kernels.hpp (resets, landmark
// scenarios); this file holds the host side: variant tables, dispatch and the extern "C" entry
// points.  formation_hip_f64.hip builds the step kernel's source in double (tests only).

#include <atomic>
#include <cstdarg>
#include <cstdlib>
#include <vector>
#include "fg_common.hpp"
#include "fg_pair_loops.hpp"
#include "fg_obs_writers.hpp"
#include "fg_step_kernel.hpp"
#include "fg_rollout_kernels.hpp"
#include "fg_aux_kernels.hpp"
#include "fg_scn_lane_kernel.hpp"
#include "fg_hd_lane_kernel.hpp"
#include "fg_policy_kernels.hpp"

namespace fg {

// ---------------------------------------------------------------------------
// host side: geometry selection, validation, launches
// ---------------------------------------------------------------------------
static thread_local char g_err[256] = "";

static int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

struct Geometry { int G, T, E, lds; };

// Dry run of the dispatch (fg_describe_launch): while `g_describe` points at a caller's buffer the launchers below append the
// kernel instantiation and launch geometry they WOULD use and return without touching the device.  A CPU-side test holds a
// committed snapshot of these choices over a grid of shapes (tests/golden/dispatch.json): a threshold that moves changes the
// snapshot, not just a timing somebody may or may not look at.
static thread_local char* g_describe = nullptr;
static thread_local int g_describe_cap = 0;
static bool describe(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
static bool describe(const char* fmt, ...) {
    if (!g_describe) return false;
    const int used = (int)strlen(g_describe);
    if (used < g_describe_cap - 1) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(g_describe + used, (size_t)(g_describe_cap - used), fmt, ap);
        va_end(ap);
    }
    return true;
}

// A launch goes to the device its DATA lives on: if that is not the calling thread's current device (an env living
// on cuda:1 driven from a thread whose current device is cuda:0), switch for the duration of the call.  A non-NULL
// stream names its device; the NULL stream is "the default stream of whatever device is current", which says nothing
// (torch hands out 0 for the default stream of EVERY device), so there the device is read off the first state pointer
// (hipPointerGetAttributes).  A process that sees one GPU never pays for either query.
static int visible_devices() {
    static const int n = [] { int c = 0; return hipGetDeviceCount(&c) == hipSuccess ? c : 1; }();
    return n;
}
// the device a launch with this stream and this first data pointer belongs to; -1 = cannot tell (stay on the current device)
static int resolve_device(void* stream, const void* data) {
    if (stream) {
        hipDevice_t sdev = -1;
        if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess) return (int)sdev;
        (void)hipGetLastError();
        return -1;
    }
    if (data) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, data) == hipSuccess && attr.type == hipMemoryTypeDevice) return attr.device;
        (void)hipGetLastError();                                 // a host / unknown pointer: leave the error state clean
    }
    return -1;
}
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    DeviceGuard(void* stream, const void* data) {
        if (visible_devices() < 2) return;
        const int target = resolve_device(stream, data);
        if (target < 0) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != target) switched = hipSetDevice(target) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// Kernels that need more than the default 64 KiB of dynamic LDS opt in once per (kernel, device): the
// attribute belongs to the device the function is loaded on, and a process may drive several GPUs (one
// thread each: the bookkeeping is atomic).
static hipError_t raise_lds_limit(const void* fn, int lds, std::atomic<unsigned long long>* done_mask) {
    if (lds <= 64 * 1024) return hipSuccess;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if ((done_mask->load(std::memory_order_acquire) >> dev) & 1ull) return hipSuccess;
    const hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err == hipSuccess) done_mask->fetch_or(1ull << dev, std::memory_order_release);
    return err;
}

static int pow2ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

template <int NC, int G, int T, int E, bool IDX, bool OPTS>
static hipError_t launch_v(const Args& a, int grid, int lds, hipStream_t st) {
    const int n_split = a.N | ((E == 1 && a.split > 1 ? a.split : 0) << 16);
    if (describe("step_kernel<%d,%d,%d,%d,%d,%d,%d> grid %d lds %d%s; ", NC, G, T, E, (int)IDX, (int)OPTS,
                 (int)(!IDX && !OPTS && NC >= 27 && a.K == 1 && a.obs_every == 1), grid, lds, a.obs_only ? " obs-only" : ""))
        return hipSuccess;
    if (lds > 64 * 1024) {                 // run-time N close to 1024 (+ the per-agent tables of the OPTS instantiation)
        static std::atomic<unsigned long long> raised{0};
        const hipError_t err = raise_lds_limit((const void*)&step_kernel<NC, G, T, E, IDX, OPTS>, lds, &raised);
        if (err != hipSuccess) return err;
    }
    // the single-step instantiation (no step loop, no slot arithmetic) of the plain variants from 27 agents up: 27 x 4096
    // 15.15 -> 15.03 us, 81 x 2048 -1 %; at 3 and 9 agents it is slower (profiles/r02_step/kone.txt)
    if (!IDX && !OPTS && NC >= 27 && a.K == 1 && a.obs_every == 1) {
        hipLaunchKernelGGL((step_kernel<NC, G, T, E, false, false, (!IDX && !OPTS && NC >= 27)>), dim3(grid), dim3(T), lds, st,
                           a.B, n_split, (const float*)a.px, (const float*)a.py, (const float*)a.vx, (const float*)a.vy,
                           (const float*)a.shape, (const float*)a.ivel, (const int32_t*)a.step, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((step_kernel<NC, G, T, E, IDX, OPTS>), dim3(grid), dim3(T), lds, st,
                       a.B, n_split, (const float*)a.px, (const float*)a.py, (const float*)a.vx, (const float*)a.vy,
                       (const float*)a.shape, (const float*)a.ivel, (const int32_t*)a.step, a);
    return hipGetLastError();
}

using LaunchFn = hipError_t (*)(const Args&, int, int, hipStream_t);
// One entry per (agent count, workgroup shape).  plain / idx (+ landmark-index outputs) / opts (idx + World options
// no reference scenario enables).  min_B: the entry is the default from that batch size up (MI355X sweeps,
// profiles/README.md: a batch that fills the chip many times over streams best with 8 envs per workgroup, a
// single-generation batch prefers one env per wave with spare writer waves).  NC = 0 entries take N at run time.
struct Variant { int NC, G, T, E, min_B; LaunchFn plain, idx, opts; };
#define FG_VARIANT(NC, G, T, E, MINB) {NC, G, T, E, MINB, &launch_v<NC, G, T, E, false, false>, \
                                       &launch_v<NC, G, T, E, true, false>, &launch_v<NC, G, T, E, true, true>}
#define FG_VARIANT_BIG(NC, G, T, E, MINB) {NC, G, T, E, MINB, &launch_v<NC, G, T, E, false, false>, nullptr, nullptr}

// batch-size thresholds of the dispatch below (MI355X sweeps; the profile behind each is named where it is used)
constexpr int FG_SPLIT_MAX_B = 128;        // split step (launch_step) up to this many envs
constexpr int FG_WIDE81_MIN_B = 16384;     // single-step launches at 81 agents take the pipelined kernel from this batch size up
constexpr int FG_WIDE243_MIN_B = 4096;     // single-step launches at 243 agents: below this the pipelined kernel's 4-env batches leave CUs
                                           // idle and one env per workgroup (step_kernel) is up to 3x faster (profiles/r02_step/wide243_min_b.txt)
static const Variant kVariants[] = {
    FG_VARIANT(3, 4, 128, 16, 0), FG_VARIANT_BIG(3, 4, 128, 32, 65536),   // every lane owns an agent: 3 x 262144 35.3 -> 27.8 us (profiles/r02_step/n3_big_batches.txt)
    FG_VARIANT(9, 16, 128, 4, 0), FG_VARIANT_BIG(9, 16, 128, 8, 16384),
    FG_VARIANT(27, 32, 256, 4, 0), FG_VARIANT_BIG(27, 32, 256, 8, 32768),
    FG_VARIANT(81, 128, 128, 1, 0),
    FG_VARIANT(243, 256, 256, 1, 0),
    // the agent counts of the other hierarchies get_action_BFS takes (per_layer 2, 4, 5, 8: __init__.py:49-56): compile-time
    // N with the register-cached rows writer for the plain launch; index outputs / World options take the run-time-N kernels
    FG_VARIANT_BIG(4, 4, 128, 16, 0), FG_VARIANT_BIG(8, 8, 128, 8, 0), FG_VARIANT_BIG(16, 16, 128, 4, 0),
    FG_VARIANT_BIG(25, 32, 256, 4, 0), FG_VARIANT_BIG(32, 32, 256, 4, 0), FG_VARIANT_BIG(64, 64, 256, 2, 0),
    FG_VARIANT_BIG(125, 128, 128, 1, 0),
    FG_VARIANT(0, 4, 64, 16, 0), FG_VARIANT(0, 8, 64, 8, 0), FG_VARIANT(0, 16, 64, 4, 0), FG_VARIANT(0, 32, 128, 4, 0),
    FG_VARIANT(0, 64, 128, 2, 0), FG_VARIANT(0, 128, 128, 1, 0), FG_VARIANT(0, 256, 256, 1, 0),
    FG_VARIANT(0, 512, 512, 1, 0), FG_VARIANT(0, 1024, 1024, 1, 0),
};

// need_full: the launch wants the idx / opts instantiation (only the first entry of an agent count has them)
// generic: the run-time-N instantiation whatever N is (its flat observation writer is the one that knows the
//          communication block of non-silent agents, FgParams.comm_state)
static const Variant* variant_for(int N, int B = 0, bool need_full = false, bool generic = false) {
    if (N < 2 || N > FG_MAX_AGENTS) return nullptr;
    const Variant* best = nullptr;
    for (const Variant& v : kVariants) {
        if (generic || v.NC != N || B < v.min_B || (need_full && !v.opts)) continue;
        if (!best || v.min_B > best->min_B) best = &v;
    }
    if (best) return best;
    const int G = N <= 64 ? (pow2ceil(N) < 4 ? 4 : pow2ceil(N)) : (pow2ceil(N) < 128 ? 128 : pow2ceil(N));
    for (const Variant& v : kVariants)
        if (v.NC == 0 && v.G == G) return &v;
    return nullptr;
}

static bool geometry_for(int N, Geometry* g, int B = 0, bool need_full = false, bool generic = false) {
    const Variant* v = variant_for(N, B, need_full, generic);
    if (!v) return false;
    g->G = v->G; g->T = v->T; g->E = v->E;
    g->lds = v->E * env_block_floats(N) * (int)sizeof(float) + 72 * (int)sizeof(float);
    return true;
}

static bool world_options_set(const FgParams& p) {
    return p.num_walls > 0 || p.u_noise > 0.f || p.max_speed > 0.f || p.accel > 0.f || p.agent_props || p.comm_state;
}

static int launch_step(Args a, hipStream_t st) {
    Geometry g;
    const bool opts = world_options_set(a.p);
    const bool idx = a.near_lm || a.near_ag || a.hd_idx;
    const bool generic = a.p.comm_state != nullptr;
    const Variant* v = variant_for(a.N, a.B, opts || idx, generic);
    if (!v || !geometry_for(a.N, &g, a.B, opts || idx, generic)) return fail(FG_ERR_UNSUPPORTED_N, "N must be in [2, 1024]%s");
    if (opts) g.lds += (3 * npad(a.N) + 2 * g.E * a.N) * (int)sizeof(float);    // per-agent mass / size / flags, per-env comm states
    a.coll_scale = (float)((double)a.p.collide_thresh / (double)a.p.dist_min);
    const int grid = (a.B + g.E - 1) / g.E;
    // Split step: more than 64 agents (one env per workgroup) and fewer envs than half the chip's CUs.  One workgroup per
    // env would leave most CUs idle during the observation stream, which is 99 % of the bytes; so the fused kernel runs
    // WITHOUT the observation (state, reward, done, reset) and a second launch of S workgroups per env streams the
    // observation from the final state - the very values the fused kernel holds in LDS, hence the same bits
    // (243 x 64: 43.4 -> 37.4 us, 256 x 64: 69 -> 41, 1024 x 64: 581 -> 394-448; neutral at 81 agents; 243 x 128 loses 8 %,
    // hence 96 there: profiles/r02_step/split_step.txt)
    if (g.E == 1 && a.K == 1 && a.do_phys && a.do_post && a.obs && a.B <= (a.N == 243 ? (FG_SPLIT_MAX_B * 3) / 4 : FG_SPLIT_MAX_B)) {
        int S = 256 / a.B;
        if (S > 8) S = 8;
        Args a1 = a; a1.obs = nullptr;
        hipError_t err = (opts ? v->opts : idx ? v->idx : v->plain)(a1, grid, g.lds, st);
        if (err != hipSuccess) return fail(FG_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(err));
        Args a2 = a;
        a2.do_phys = 0; a2.obs_only = 1; a2.split = S; a2.p.auto_reset = 0;
        a2.rew = nullptr; a2.indiv = nullptr; a2.done = nullptr; a2.near_lm = nullptr; a2.near_ag = nullptr; a2.hd_idx = nullptr;
        err = (generic ? v->opts : v->plain)(a2, grid * S, g.lds, st);   // the communication block lives in the OPTS instantiation
        if (err != hipSuccess) return fail(FG_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(err));
        return FG_OK;
    }
    const hipError_t err = (opts ? v->opts : idx ? v->idx : v->plain)(a, grid, g.lds, st);
    if (err != hipSuccess) return fail(FG_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

// 64 < N <= 256: producer / writer pipelined kernel (rollout: over steps; single step: over env batches)
// PER > 0: closed loop with the PER-ary demo controller inside the kernel (fg_rollout_hd_policy)
template <int NC, int A, int E, int TW, int PER, bool BATCHES = false>
static int launch_wide_v(Args a, hipStream_t st) {
    a.groups = 1;
    if constexpr (BATCHES) {
        // env batches per workgroup: enough to overlap batch g+1's pair loops with batch g's store
        // stream, few enough to keep every CU busy (one workgroup per CU; MI355X sweep, profiles/README.md)
        const int batches = (a.B + E - 1) / E;
        a.groups = batches / 256;
        if (a.groups < 1) a.groups = 1;
        if (a.groups > 64) a.groups = 64;
    }
    const int grid = (a.B + E * a.groups - 1) / (E * a.groups);
    const int lds = E * roll_block_floats(NC) * (int)sizeof(float) +
                    (PER > 0 ? E * policy_block_units(NC) * (int)sizeof(float2) : 0);
    if (describe("rollout_kernel_wide<%d,%d,%d,%d,%d,%d> grid %d lds %d groups %d; ", NC, A, E, TW, PER, (int)BATCHES, grid, lds, a.groups))
        return FG_OK;
    static std::atomic<unsigned long long> raised{0};
    hipError_t err = raise_lds_limit((const void*)&rollout_kernel_wide<NC, A, E, TW, PER, BATCHES>, lds, &raised);
    if (err == hipSuccess) {
        hipLaunchKernelGGL((rollout_kernel_wide<NC, A, E, TW, PER, BATCHES>), dim3(grid), dim3(E * 64 + TW), lds, st, a);
        err = hipGetLastError();
    }
    if (err != hipSuccess) return fail(FG_ERR_HIP, "pipelined launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}
// 4 envs (= producer waves) and 4 writer waves per workgroup (MI355X sweeps: profiles/README.md, profiles/r03_wide/)
template <int NC, int PER>
static int launch_wide(const Args& a, hipStream_t st) {
    static_assert(NC > 64 && NC <= 256, "one producer wave per env, up to 4 agents per lane");
    constexpr int A = (NC + 63) / 64;
    // 243 agents, at most 256 envs: one env per workgroup spreads the batch over more CUs (243 x 256 x 8: 103 -> 82 us/step,
    // 243 x 64: 90 -> 78; at 81 agents the producer wave's own chain, ~15 us per step, is the bound whatever the geometry:
    // profiles/r03_wide/ab_wide_small_batches.txt)
    // a single step (K == 1) of the open loop pipelines over env batches inside the workgroup (fg_step_hd at 81 / 243 agents)
    if constexpr (PER == 0 && (NC == 81 || NC == 243))
        if (a.K == 1) return launch_wide_v<NC, A, 4, 256, 0, true>(a, st);
    if constexpr (NC == 243)
        if (a.K > 1 && a.B <= 256) return launch_wide_v<243, 4, 1, 256, PER>(a, st);
    return launch_wide_v<NC, A, 4, 256, PER>(a, st);
}

// N <= 64, K >= 2: producer / writer pipelined rollout kernel
template <int NC, int G, int TP, int TW, int E, int WR, int PER, bool STREAM = false>
static int launch_roll_v(const Args& a, hipStream_t st) {
    const int grid = (a.B + E - 1) / E;
    constexpr int lds = roll_lds_bytes<NC, TW, E, WR, PER>();     // env blocks, writer tiles, controller tables, reward hand-over
    if (describe("rollout_kernel<%d,%d,%d,%d,%d,%d,%d,%d> grid %d lds %d; ", NC, G, TP, TW, E, WR, PER, (int)STREAM, grid, lds)) return FG_OK;
    static std::atomic<unsigned long long> raised{0};
    hipError_t err = raise_lds_limit((const void*)&rollout_kernel<NC, G, TP, TW, E, WR, PER, STREAM>, lds, &raised);
    if (err == hipSuccess) {
        hipLaunchKernelGGL((rollout_kernel<NC, G, TP, TW, E, WR, PER, STREAM>), dim3(grid), dim3(TP + TW), lds, st, a);
        err = hipGetLastError();
    }
    if (err != hipSuccess) return fail(FG_ERR_HIP, "rollout launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}
// writer of the pipelined kernels per agent count: 1 + rows per LDS tile (the rows must divide N: 9 of 27 and 9, 5 of 25,
// 8 of 16 and 32), 0 = register-cached rows (few agents: a tile is too small to pay; 64 agents: a row is a wave)
constexpr int roll_writer(int n) { return n == 9 ? FG_WR_GATHER : n == 27 ? 10 : n == 25 ? 6 : (n == 16 || n == 32) ? 9 : 0; }

// ---- which instantiation a (agents, batch, buffer) gets: ONE rule table per agent-count class --------------------------------
// A rule = a batch range, conditions on the observation buffer, the instantiation, and the measurement that put the threshold
// where it is.  The FIRST rule that applies is taken; a row whose instantiation does not exist for this (agents, per) - `on`
// false - is skipped, every table ends with a catch-all.  tests/golden/dispatch.json holds what these tables answer over a grid
// of shapes (tests/test_dispatch_snapshot.py): a moved threshold is a diff of that file.
//   rollout_kernel<NC, G, TP, TW, E, WR, PER, STREAM>: G lanes per env, TP producer + TW writer threads, E envs per workgroup,
//   WR the observation writer (0 rows, FG_WR_GATHER, else 1 + rows per LDS tile), STREAM the HBM-streaming form of the writer
using RollFn = int (*)(const Args&, hipStream_t);
enum : unsigned {
    R_HBM = 1u,          // the launch's observation buffer does not fit the 256 MiB Infinity Cache (> 400 MB)
    R_CACHED = 2u,       // ... it does
    R_PLACED = 4u,       // FgParams.obs_placed: the buffer was composed of chunks spread over the device's memory (fg_arena_*)
    R_BEYOND_IC = 16u,   // ... it exceeds the Infinity Cache at all (> 260 MB): the late-round-5 geometry rules, which win from there on
                         // (profiles/r05_hbm_threshold_ab.txt: 9 x 8192 x 20 = 318 MB 3.30 -> 3.21 us/step, 8 x 16384 x 10 5.01 -> 4.59;
                         // at 100-150 MB they lose 3-4 %)
    R_FILL32 = 8u,       // workgroups of 32 envs, one per CU at a time (1024 threads), fill their generations of 256 well enough
                         // (run_roll_rules)
    R_FILL32B = 32u,     // the same for the 512- / 768-thread 32-env workgroups of 8 agents: whole generations of 256, a last one at least
                         // half full, or six and more (profiles/r05_fill_rule_ab.txt: 1.12 / 1.25 / 2.25 generations lose 0-11 %)
};
struct RollRule {
    int b_lo, b_hi;      // batch sizes the rule covers (inclusive)
    unsigned need;       // R_* conditions, all of them
    RollFn fn;           // nullptr: no such instantiation for this (agents, per)
    const char* why;     // the measurement behind the threshold
};
constexpr int B_ANY = 0x7fffffff;
template <bool ON, int NC, int G, int TP, int TW, int E, int WR, int PER, bool STREAM = false>
constexpr RollFn roll_fn() {
    if constexpr (ON) return &launch_roll_v<NC, G, TP, TW, E, WR, PER, STREAM>;
    else return nullptr;
}
template <size_t R>
static int run_roll_rules(const RollRule (&rules)[R], const Args& a, hipStream_t st) {
    const double obs_bytes = (double)(a.K / a.obs_every) * a.B * (double)a.obs_pitch * 8.0;
    const bool hbm = obs_bytes > 400e6, beyond_ic = obs_bytes > 260e6;
    for (const RollRule& r : rules) {
        if (!r.fn || a.B < r.b_lo || a.B > r.b_hi) continue;
        if (((r.need & R_HBM) && !hbm) || ((r.need & R_CACHED) && hbm) || ((r.need & R_PLACED) && !a.p.obs_placed)) continue;
        if ((r.need & R_BEYOND_IC) && !beyond_ic) continue;
        if (r.need & R_FILL32B) {
            const int wgs = (a.B + 31) / 32, full = wgs / 256, rest = wgs % 256;
            if (!(rest == 0 || rest >= 128 || full >= 6)) continue;
        }
        if (r.need & R_FILL32) {
            // generations of 256 workgroups (one 1024-thread workgroup per CU): whole ones, a last one at least three quarters full,
            // or six and more - profiles/r05_fill_rule_ab.txt: 1.00 +8 %, 1.12 -10 %, 1.25 -7 %, 1.50 -2 %, 1.75 +12 % (placed: +7 %),
            // 2.25 -3 %, 3.75 +9 % (placed: +2 %), 6.00 +7 %; half-full last generations (2.50 +1 %, 3.50 +2.5 %, 4.50 +3 % on ordinary
            // allocations) lose on placed buffers (3.50: -6.5 % on two boxes) and are left out
            const int wgs = (a.B + 31) / 32, full = wgs / 256, rest = wgs % 256;
            if (!(rest == 0 || rest >= 192 || full >= 6)) continue;
        }
        return r.fn(a, st);
    }
    return fail(FG_ERR_UNSUPPORTED_N, "no rollout rule for this shape%s");       // (unreachable: every table ends with a catch-all)
}

// up to 8 agents (3, 4, 8): one wave of producers, two writer waves with the rows writer (rows of 9 units keep 9 of a wave's
// 64 lanes busy; one writer wave was the bottleneck: 3 x 1024 x 20 2.28 -> 1.36 us/step, 3 x 16384 2.48 -> 1.63, 3 x 65536
// 7.46 -> 7.05; four waves lose from 16384 envs up)
template <int NC, int PER>
static int launch_roll_8(const Args& a, hipStream_t st) {
    constexpr int G = NC <= 4 ? 4 : 8, E = 64 / G;
    constexpr bool N8 = NC == 8, OPEN = PER == 0;
    static constexpr RollRule rules[] = {
        {4096, 5119, R_HBM, roll_fn<N8 && OPEN, NC, G, 128, 512, 128 / G, FG_WR_GATHER, PER>(),
         "profiles/r04_writers_ab.txt: one workgroup per CU into a buffer beyond the Infinity Cache, 16 envs and eight writer waves "
         "per workgroup - 8 x 4096 x 120 1.69 -> 1.36 us/step (the writer waves bound it); profiles/r05_8x32_ab.txt: with the span form of "
         "the gather writer instead of LDS tiles 8 x 4096 x 160 1.152 -> 1.057 (0.74 -> 0.81 real), 8 x 4500 2.18 -> 1.98"},
        {8192, 49152, R_BEYOND_IC | R_FILL32B, roll_fn<N8 && OPEN, NC, G, 256, 512, 32, FG_WR_GATHER, PER>(),
         "profiles/r05_8x32_ab.txt: 32 envs, 4 producer and 8 writer waves per workgroup - 8 x 8192 x 80 2.36 -> 2.22 us/step, 8 x 16384 "
         "4.74-4.97 -> 4.28-4.39, 8 x 24576 6.96 -> 6.48, 8 x 32768 8.99 -> 8.61, 8 x 49152 13.36 -> 13.10 (0.80 of 8 TB/s in real bytes); "
         "8 x 65536 stays with the small workgroups (17.6-17.8 vs 17.9-18.1)"},
        {12288, B_ANY, R_FILL32B, roll_fn<N8 && PER == 2, NC, G, 256, 256, 32, FG_WR_GATHER, PER>(),
         "profiles/r05_8x32_ab.txt: closed loop, 32 envs, 4 + 4 waves per workgroup - 8 x 12288 6.18 -> 3.90 us/step, 8 x 16384 6.88 -> 4.65, "
         "8 x 24576 9.54 -> 7.39, 8 x 32768 10.64 -> 9.42, 8 x 65536 20.1-20.4 -> 18.95-19.4 (0.50-0.69 -> 0.72-0.74 real); 8 x 8192 keeps the "
         "small workgroups (2.92 vs 3.03)"},
        {12288, 49152, R_FILL32B, roll_fn<N8 && PER == 8, NC, G, 256, 256, 32, FG_WR_GATHER, PER>(),
         "profiles/r05_8x32_ab.txt: the one-level 8-ary controller - 8 x 16384 9.93 -> 6.52 us/step, 8 x 24576 13.49 -> 9.87, 8 x 32768 14.65 -> "
         "13.28, 8 x 49152 21.25 -> 20.35; equal at 65536 (27.9 vs 28.2)"},
        {32768, B_ANY, 0, roll_fn<N8 && PER == 8, NC, G, 64, 64, E, FG_WR_GATHER, PER>(),
         "profiles/r05_r8_ab.txt: closed loop over many workgroup generations is bound by the producers' chain (controller + physics, "
         "~3.6 us per step); ONE writer wave leaves room for more resident producer waves - 8 x 65536 22.7 -> 20.9 us/step; at 8192 "
         "envs two writer waves stay ahead (3.17 vs 3.50)"},
        {4096, B_ANY, 0, roll_fn<N8, NC, G, 64, 128, E, FG_WR_GATHER, PER>(),
         "profiles/r04_gather_ab.txt: an env's 1536 bytes as one contiguous span of 16-byte stores instead of rows in 64- / 128-byte "
         "pieces (LDS tiles: 8 x 65536 23.3 -> 20.2 us/step, 8 x 8192 3.14 -> 2.69; the gather writer since: 18.9 -> 18.6, equal at "
         "8192; closed loop 23.2 -> 22.7); 8 x 1024 is bound by the producers' chain and keeps the rows writer (1.09 vs 1.48)"},
        {0, B_ANY, 0, roll_fn<true, NC, G, 64, 128, E, 0, PER>(), "the default of this class (comment above)"},
    };
    return run_roll_rules(rules, a, st);
}
// 9 ... 16 agents: a batch of <= 4096 envs is bound by the producers' dependent chain and wants many small workgroups with
// the row writer; larger batches are store-bound and want whole 128-byte lines per workgroup with the LDS-tile writer.
template <int NC, int PER>
static int launch_roll_16(const Args& a, hipStream_t st) {
    constexpr int WR = roll_writer(NC);
    constexpr bool N9 = NC == 9, OPEN = PER == 0;
    // 9 agents into a rollout buffer beyond the Infinity Cache (128 steps of 4096 envs: 1 GB).  Cycle stamps inside the kernel
    // (profiles/r04_trace_ab.txt) showed the writer waves, not the producers' chain and not the memory, bounding it: 5400-5700
    // cycles per step for four waves with the tile writer against 2800 for the producers.  Hence the gather writer
    // (fg_obs_writers.hpp; 9 x 8192 3.40 -> 3.10 us/step, 9 x 16384 6.8 -> 6.3, profiles/r04_gather_ab.txt) and, while a
    // workgroup has a CU to itself, more writer waves per env
    static constexpr RollRule rules[] = {
        {1025, 2048, R_HBM, roll_fn<N9 && OPEN, NC, 16, 128, 256, 8, WR, PER>(),
         "profiles/r04_writers_ab.txt, r04_gather_geom.txt: 9 x 2048 x 250 1.47 -> 1.31 us/step"},
        {2049, 4096, R_HBM, roll_fn<N9 && OPEN, NC, 16, 256, 512, 16, WR, PER>(),
         "profiles/r04_writers_ab.txt, r04_gather_geom.txt: 9 x 4096 x 128 2.42 -> 1.83 us/step (1.43 since: profiles/r05_9x4096_rollout.md)"},
        {2049, 4096, R_HBM, roll_fn<N9 && !OPEN, NC, 16, 256, 256, 16, WR, PER>(),
         "profiles/r04_writers_ab.txt: the closed loop's controller tables leave LDS for four writer waves"},
        {8192, B_ANY, R_BEYOND_IC | R_FILL32, roll_fn<N9 && OPEN, NC, 16, 512, 512, 32, WR, PER>(),
         "profiles/r05_9x32_ab.txt: 32 envs, 8 producer and 8 writer waves per workgroup - 9 x 8192 x 64 2.97 -> 2.84 us/step, 9 x 16384 5.69 -> "
         "5.44, 9 x 32768 11.49 -> 11.10, 9 x 65536 23.9 -> 23.3 (0.76 -> 0.78-0.795 of 8 TB/s in real bytes); a batch that leaves the last "
         "generation half empty loses (9 x 12288: 4.42 -> 4.97)"},
        {8192, 32768, R_BEYOND_IC, roll_fn<NC == 16 && OPEN, NC, 16, 256, 512, 16, WR, PER>(),
         "profiles/r05_16x_ab.txt: eight writer waves per 16 envs - 16 x 8192 x 24 8.61 -> 8.10 us/step, 16 x 16384 16.91 -> 16.03, 16 x 32768 "
         "33.8 -> 33.55 (0.77-0.79 -> 0.80-0.83 of 8 TB/s in real bytes); 16 x 65536 and the closed loop keep four (69.8 vs 73.7; 16.5 vs 17.25)"},
        {8192, B_ANY, 0, roll_fn<true, NC, 16, 256, 256, 16, WR, PER>(), "profiles/README.md (round 2 sweeps): store-bound, whole lines per workgroup"},
        {4097, B_ANY, 0, roll_fn<true, NC, 16, 128, 128, 8, WR, PER>(), "profiles/README.md (round 2 sweeps)"},
        {0, B_ANY, 0, roll_fn<true, NC, 16, 64, 128, 4, 0, PER>(), "the producers' chain bounds it: many small workgroups, rows writer"},
    };
    return run_roll_rules(rules, a, st);
}
// 17 ... 32 agents (25, 27, 32).  Defaults from the MI355X sweeps at 27 agents (profiles/README.md): 16 envs per workgroup =
// 8 producer + 4 writer waves, one workgroup per CU at 4096 envs, LDS-tile writer.
template <int NC, int PER>
static int launch_roll_32(const Args& a, hipStream_t st) {
    constexpr int WR = roll_writer(NC);
    constexpr bool POLICY = PER > 0, N27 = NC == 27;
    static constexpr RollRule rules[] = {
        // Batches that do not fill the chip with 16-env workgroups (4096 envs = 256 workgroups = one per CU): fewer envs per
        // workgroup, so that the batch still spreads over the CUs
        {0, 512, 0, roll_fn<N27, NC, 32, 64, 256, 2, WR, PER>(),
         "profiles/r03_wide/ab_27_small_batches_*.txt: 27 x 512 x 20 9.1 -> 2.6 us/step, 27 x 256 9.1 -> 2.6"},
        {0, 1024, 0, roll_fn<true, NC, 32, 128, 256, 4, WR, PER>(), "profiles/r03_wide/ab_27_small_batches_*.txt: 27 x 1024 x 20 9.2 -> 3.4 us/step"},
        {0, 2048, 0, roll_fn<N27 && POLICY, NC, 32, 256, 256, 8, WR, PER>(), "profiles/r03_wide/ab_27_small_batches_*.txt: 27 x 2048 8.0 -> 6.0-6.5 us/step"},
        {0, 2048, R_HBM, roll_fn<N27 && !POLICY, NC, 32, 256, 512, 8, WR, 0, true>(),
         "profiles/r03_wide/ab_27_small_batches_*.txt: 27 x 2048 8.0 -> 6.0-6.5 us/step, placed or not; the HBM-streaming form of the tile writer"},
        {0, 2048, R_CACHED, roll_fn<N27 && !POLICY, NC, 32, 256, 512, 8, WR, 0, false>(), "... and its plain form while the buffer stays in the Infinity Cache"},
        // (27 x 2560 and up are faster with 16 envs per workgroup)
        // 25 and 32 agents into buffers beyond the Infinity Cache: the rows writer with four writer waves (their 5- and 8-row tiles
        // are smaller than 27's 9-row ones, and the streaming form's pacing was tuned on those)
        {4096, B_ANY, R_HBM | R_PLACED, roll_fn<!POLICY && !N27, NC, 32, 512, 256, 16, 0, 0>(),
         "profiles/r05_25_32_ab.txt: placed buffers - 25 x 4096 x 20 10.71 -> 10.44 us/step, 25 x 8192 21.29 -> 20.23, 32 x 4096 15.50 -> 15.20"},
        {4096, B_ANY, R_BEYOND_IC, roll_fn<!POLICY && NC == 32, NC, 32, 512, 256, 16, 0, 0>(),
         "profiles/r05_25_32_ab.txt: ordinary allocations - 32 x 4096 x 20 18.56 -> 18.00 us/step"},
        {0, B_ANY, R_HBM, roll_fn<POLICY, NC, 32, 512, 512, 16, 0, PER>(),
         "profiles/r03_wide/ab_closed_loop_*: the closed-loop instantiation cannot hold 16 tiles beside its controller tables; it takes its "
         "8 writer waves with the rows writer - 11.7 vs 13.35 us/step on a placed buffer; profiles/r05_hint_audit.txt: on an ordinary "
         "allocation as well (27 x 2560 x 32 13.05 -> 8.44, 27 x 8192 27.6 -> 26.0, 25 x 4096 12.8 -> 11.9; 27 x 4096 equal)"},
        {0, B_ANY, R_HBM | R_PLACED, roll_fn<!POLICY, NC, 32, 512, 512, 16, WR, 0, true>(),
         "profiles/r03_wide/ab_27_writers_*.txt: a buffer composed of chunks spread over the device's memory takes the stream of 8 paced writer "
         "waves - 27 x 4096 x 20 11.6-11.8 us/step against 12.9 with 4, 27 x 8192 23.3-23.5 against 25.9-26.4, 27 x 16384 47.6-48.4 against "
         "48.7-49.4; on an ordinary allocation 8 waves lose (13.2-14.4 against 12.75, round 2)"},
        {0, 3072, R_HBM, roll_fn<!POLICY, NC, 32, 512, 512, 16, WR, 0, true>(),
         "profiles/r03_wide/ab_27_mid_batches.txt: ... and so does any buffer while fewer than ~200 of the 256 CUs have a workgroup (a "
         "workgroup's own store rate is the bound then): 27 x 2560 x 37 on an ordinary allocation 12.4 -> 8.1 us/step, 27 x 3072 12.5 -> 10.7 "
         "(27 x 3584 equal, 27 x 4000 12.8 vs 14.0 stays with 4)"},
        {8192, B_ANY, R_HBM, roll_fn<!POLICY && N27, NC, 32, 512, 512, 16, WR, 0, true>(),
         "profiles/r05_hint_audit.txt, r05_25_32_ab.txt: from 8192 envs the eight paced writer waves win on an ordinary allocation too - "
         "27 x 8192 x 10 26.0-26.2 -> 25.2-25.5 us/step, 27 x 16384 49.6 -> 48.8-49.1 (27 x 4096 keeps four: 12.8 vs 13.1-13.2)"},
        {0, 16383, R_HBM, roll_fn<!POLICY && N27, NC, 32, 512, 256, 16, WR, 0, true>(),
         "profiles/README.md (round 2): line ownership + paced stores for batches of a few workgroup generations whose buffer does not fit the "
         "Infinity Cache; 27 agents only - profiles/r05_25_32_ab.txt: at 25 agents on an ordinary allocation this form ran 15.1 us/step (4096 "
         "envs) and 30.8 (8192) where the plain tile writer below runs 11.5 and 23.5"},
        {0, 16383, R_HBM, roll_fn<POLICY && N27, NC, 32, 512, 256, 16, WR, PER, true>(), "as above, closed loop"},
        {0, B_ANY, 0, roll_fn<true, NC, 32, 512, 256, 16, WR, PER>(), "the default of this class (comment above)"},
    };
    return run_roll_rules(rules, a, st);
}
// 64 agents: an env is a wave of producers, a row of its observation 3 stores of a writer wave
template <int NC, int PER>
static int launch_roll_64(const Args& a, hipStream_t st) {
    static constexpr RollRule rules[] = {
        {0, 1024, 0, roll_fn<PER == 0, NC, 64, 128, 256, 2, 0, 0>(), "profiles/r04_generic_n.md: a batch that does not fill the chip with 8-env workgroups"},
        {0, B_ANY, 0, roll_fn<true, NC, 64, 512, 256, 8, 0, PER>(), "profiles/r04_generic_n.md"},
    };
    return run_roll_rules(rules, a, st);
}

// The pipelined K-step kernels exist for the agent counts of the reference's hierarchies, N = per^L: the reference's own
// 3^L (README.md:34-36) and the other `per_layer` values get_action_BFS takes (__init__.py:49-56: 2, 4, 5, 8).  per = 0: open
// loop (actions from act_seq); else the closed loop of fg_rollout_hd_policy.  Returns false when (N, per) has no instantiation
// (the caller then runs step_kernel's K-loop / chained launches).
// 3 and 4 agents, open loop, contiguous observations, a batch that fills the chip: one env per lane (fg_hd_lane_kernel.hpp)
template <int NC, int PER = 0, int PW = 1>
static int launch_hd_lane(const Args& a, hipStream_t st) {
    const int grid = 8 * (((a.B + 64 * PW - 1) / (64 * PW) + 7) / 8);
    constexpr int lds = hd_lane_lds_bytes(NC, PW);
    if (describe("hd_lane_kernel<%d,%d,%d> grid %d lds %d; ", NC, PER, PW, grid, lds)) return FG_OK;
    static std::atomic<unsigned long long> raised{0};
    hipError_t err = raise_lds_limit((const void*)&hd_lane_kernel<NC, PER, PW>, lds, &raised);
    if (err != hipSuccess) return fail(FG_ERR_HIP, "rollout launch failed: %s", hipGetErrorString(err));
    hipLaunchKernelGGL((hd_lane_kernel<NC, PER, PW>), dim3(grid), dim3(128 * PW), lds, st, a);
    err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "rollout launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}
// One-env-per-lane kernels with four producer waves (256-env workgroups): how full the generations of 256 such workgroups (one per CU) must be - profiles/r05_fill_rule_ab.txt, eleven batch sizes per kernel:
// whole generations or a last one at least three quarters full (a single one: seven eighths: 57344 envs +12 %, 49152 -11 %); `half`:
// a last generation at least half full is enough (basic and formation_hd_env at 3 agents: 98304 envs +5 ... +8 %).
static bool lane_wide_fill(int B, bool half) {
    const int wgs = (B + 255) / 256, full = wgs / 256, rest = wgs % 256;
    return rest == 0 || (full >= 1 && rest >= (half ? 128 : 192)) || (full == 0 && rest >= 224) || full >= 6;
}
// from 32768 envs (512 workgroups of 64 envs: two per CU) the one-env-per-lane kernel wins - 3 x 65536 6.3 -> 3.4 us/step, 4 x 65536
// 9.2 -> 5.0; below, the lane-per-agent kernel spreads a batch over more waves (3 x 16384: 1.62 vs 1.89): profiles/r04_hd_lane_ab.txt
constexpr int FG_HD_LANE_MIN_B = 32768;

static bool launch_pipelined(const Args& a, int per, hipStream_t st, int* rc) {
#define FG_ROLL(FN, NN, PP) if (a.N == NN && per == PP) { *rc = FN<NN, PP>(a, st); return true; }
    if (a.B >= FG_HD_LANE_MIN_B && a.obs_pitch == 3LL * a.N * a.N) {
        // Several producer waves per workgroup - one span of 256 (128) envs per workgroup and step - into a buffer beyond the
        // Infinity Cache (profiles/r05_lane_pw_ab.txt): open loop from 65536 envs with four (3 x 65536 3.02 -> 2.79 us/step, 0.73 ->
        // 0.79 of 8 TB/s in real bytes; 3 x 131072 6.13 -> 5.60, 4 x 131072 10.44 -> 9.61; 4 x 49152 would lose, 3.84 -> 4.47); the
        // closed loop from 98304 envs with two (3 x 98304 5.02 -> 4.28, 4 x 131072 10.18 -> 9.32; at 65536 one stays ahead, 2.70 vs 2.82)
        const double obs_bytes = (double)(a.K / a.obs_every) * a.B * (double)a.obs_pitch * 8.0;
        const bool hbm = obs_bytes > 400e6;
        if (per == 0) {                                 // (3 agents from 260 MB: 3 x 65536 x 20 = 283 MB 3.32 -> 3.06 us/step; 4 agents at 251 MB lose)
            if (a.N == 3 && obs_bytes > 260e6 && lane_wide_fill(a.B, true)) { *rc = launch_hd_lane<3, 0, 4>(a, st); return true; }
            if (a.N == 4 && hbm && lane_wide_fill(a.B, false)) { *rc = launch_hd_lane<4, 0, 4>(a, st); return true; }
        }
        if (hbm && per > 0 && a.B >= 98304) {
            if (a.N == 3 && per == 3) { *rc = launch_hd_lane<3, 3, 2>(a, st); return true; }
            if (a.N == 4 && per == 2) { *rc = launch_hd_lane<4, 2, 2>(a, st); return true; }
        }
        if (a.N == 3 && per == 0) { *rc = launch_hd_lane<3>(a, st); return true; }
        if (a.N == 4 && per == 0) { *rc = launch_hd_lane<4>(a, st); return true; }
        // ... and the closed loop with the controller on the lane's registers (bfs_policy_lane)
        if (a.N == 3 && per == 3) { *rc = launch_hd_lane<3, 3>(a, st); return true; }
        if (a.N == 4 && per == 2) { *rc = launch_hd_lane<4, 2>(a, st); return true; }
        if (a.N == 4 && per == 4) { *rc = launch_hd_lane<4, 4>(a, st); return true; }
    }
    FG_ROLL(launch_roll_8, 3, 0) FG_ROLL(launch_roll_8, 3, 3)
    FG_ROLL(launch_roll_8, 4, 0) FG_ROLL(launch_roll_8, 4, 2) FG_ROLL(launch_roll_8, 4, 4)
    FG_ROLL(launch_roll_8, 8, 0) FG_ROLL(launch_roll_8, 8, 2) FG_ROLL(launch_roll_8, 8, 8)
    FG_ROLL(launch_roll_16, 9, 0) FG_ROLL(launch_roll_16, 9, 3)
    FG_ROLL(launch_roll_16, 16, 0) FG_ROLL(launch_roll_16, 16, 2) FG_ROLL(launch_roll_16, 16, 4)
    FG_ROLL(launch_roll_32, 25, 0) FG_ROLL(launch_roll_32, 25, 5)
    FG_ROLL(launch_roll_32, 27, 0) FG_ROLL(launch_roll_32, 27, 3)
    FG_ROLL(launch_roll_32, 32, 0) FG_ROLL(launch_roll_32, 32, 2)
    FG_ROLL(launch_roll_64, 64, 0) FG_ROLL(launch_roll_64, 64, 2) FG_ROLL(launch_roll_64, 64, 4) FG_ROLL(launch_roll_64, 64, 8)
    // more than 64 agents: the pipelined kernels keep one env on ONE producer wave (~15 / ~78 us of dependent work per step
    // at 81 / 243 agents); a batch too small to hide that under other envs' stores is faster in step_kernel's K-loop, which
    // spreads an env over 2 / 4 waves: 81 x 256 x 20 15.6 -> 12.5 us/step, 81 x 512 18.1 -> 16.1 (81 x 1024: 31.8 vs 33.2, stays),
    // 243 x 64 x 8 77.8 -> 37.0, 243 x 256 81.8 -> 67.8 (243 x 257 on the pipelined kernel: 103; 243 x 512 equal) -
    // profiles/r03_wide/ab_wide_small_kloop.txt.  The closed loop always runs in the pipelined kernel (one launch).
    if (per == 0 && a.B <= 512) return false;
    FG_ROLL(launch_wide, 81, 0) FG_ROLL(launch_wide, 81, 3)
    FG_ROLL(launch_wide, 125, 0) FG_ROLL(launch_wide, 125, 5)
    FG_ROLL(launch_wide, 243, 0) FG_ROLL(launch_wide, 243, 3)
#undef FG_ROLL
    return false;
}

// N = per^L with 2 <= per <= 8: fills the host-rounded constants of the hierarchy
static bool policy_levels_for(int N, int per, FgPolicyLevels* pl) {
    if (per < 2 || per > 8 || N < per) return false;
    memset(pl, 0, sizeof(*pl));
    long long n = 1;
    int L = 0;
    while (n < N && L < FG_POLICY_MAX_LEVELS) { pl->inv_sub[L] = (float)(1.0 / (double)n); n *= per; ++L; }
    if (n != N) return false;
    pl->per = per; pl->L = L; pl->inv_per = (float)(1.0 / (double)per);
    return true;
}

// Args.obs_pitch (float2 units) from FgParams.obs_env_pitch (floats; 0 = contiguous)
static int set_obs_pitch(Args* a) {
    const long long nenv = 6LL * a->N * a->N;
    const long long pitch = a->p.obs_env_pitch ? (long long)a->p.obs_env_pitch : nenv;
    if (pitch < nenv || (pitch & 1))
        return fail(FG_ERR_BAD_ARG, "params: obs_env_pitch must be 0 or an even number of floats >= 6 N^2%s");
    a->obs_pitch = pitch / 2;
    return FG_OK;
}

static int check_params(const FgParams* p) {
    if (!p) return fail(FG_ERR_BAD_ARG, "params is NULL%s");
    if (!(p->mass > 0.f) || !(p->contact_margin > 0.f) || !(p->dt > 0.f))
        return fail(FG_ERR_BAD_ARG, "params: mass, contact_margin and dt must be > 0%s");
    if (p->num_walls < 0 || p->num_walls > FG_MAX_WALLS || p->accel < 0.f || p->max_speed < 0.f || p->u_noise < 0.f)
        return fail(FG_ERR_BAD_ARG, "params: 0 <= num_walls <= 4, accel/max_speed/u_noise >= 0%s");
    if (!(p->dist_min > 0.f)) return fail(FG_ERR_BAD_ARG, "params: dist_min must be > 0%s");
    if (((uintptr_t)p->agent_props & 3u) || ((uintptr_t)p->comm_state & 7u))
        return fail(FG_ERR_ALIGNMENT, "params: agent_props must be 4-byte, comm_state 8-byte aligned%s");
    return FG_OK;
}

// one env per lane, PW producer waves (64 PW envs) and PW * scn_lane_writers(KIND) writer waves per workgroup
// Four producer waves - 256 envs, one span of 55 ... 133 KB per workgroup and step instead of four - where such workgroups
// come in whole generations of 256 (one per CU; or in six generations and more) - profiles/r05_lane_pw_ab.txt:
//   basic 3 x 65536 3.04 -> 2.83 us/step (0.72 -> 0.77 of 8 TB/s, all bytes), 3 x 131072 6.26 -> 5.67 (0.70 -> 0.77);
//   partial 5 x 131072 12.85 -> 12.51, range 4 x 131072 10.20 -> 9.33 (equal or slower at 65536: 6.44 / 6.40, 4.60 / 4.80);
//   obstacle never (5.47 / 5.65, 11.25 / 11.24); half-filled generations lose everywhere (32768 envs: 1.70 -> 2.40).
// Measured on rollouts whose observation buffer does not fit the Infinity Cache (> 260 MB: R_BEYOND_IC of the formation_hd_env rules): only those.
template <int KIND> static bool scn_lane_wide(int B, double obs_bytes) {
    if (obs_bytes <= 260e6) return false;
    if (KIND == FG_SCN_BASIC) return lane_wide_fill(B, true);
    // partial / range: whole generations from the second on only - on boxes whose placed buffers run fast the 256-env form loses 1-5 % at
    // 57344 / 65536 / 114688 envs (it wins 7-13 % there on slow compositions and ordinary allocations), at 131072 it wins on both
    const int wgs = (B + 255) / 256;
    return (KIND == FG_SCN_PARTIAL || KIND == FG_SCN_RANGE) && B >= 131072 && (wgs % 256 == 0 || wgs >= 6 * 256);
}
template <int KIND, int NN, int LL, int MM, int NBR, int PW>
static int launch_scn_lane(const ScnArgs& a, hipStream_t st) {
    constexpr int lds = scn_lane_lds_bytes(KIND, NN, LL, MM, NBR, PW);
    constexpr int threads = 64 * PW * (1 + scn_lane_writers(KIND));
    const int grid = 8 * (((a.B + 64 * PW - 1) / (64 * PW) + 7) / 8);
    if (describe("scn_lane_kernel<%d,%d,%d,%d,%d,%d> grid %d threads %d lds %d; ", KIND, NN, LL, MM, NBR, PW, grid, threads, lds)) return FG_OK;
    static std::atomic<unsigned long long> raised{0};
    const hipError_t err = raise_lds_limit((const void*)&scn_lane_kernel<KIND, NN, LL, MM, NBR, PW>, lds, &raised);
    if (err != hipSuccess) return fail(FG_ERR_HIP, "scenario launch failed: %s", hipGetErrorString(err));
    hipLaunchKernelGGL((scn_lane_kernel<KIND, NN, LL, MM, NBR, PW>), dim3(grid), dim3(threads), lds, st, a);
    return FG_OK;
}

}  // namespace fg

using namespace fg;

extern "C" {

int fg_abi_version(void) { return FG_ABI_VERSION; }

// ---- placed device memory: separately created physical chunks, mapped into fresh address ranges (HIP virtual memory
// management).  Discipline: (1) a chunk is mapped at ONE address at a time (no aliases); (2) the device is drained before
// any mapping goes away; (3) an address range that has held a mapping is NEVER used again in this process: on this stack
// (ROCm 7.2, MI355X) hipMemUnmap leaves the GPU's translations of the range behind, and a later mapping at the same
// address has most of its accesses land in the chunks that were mapped there before (profiles/r03_place/va_reuse_check.txt:
// 73-100 % of the old chunks overwritten).  So reservations are not handed back (hipMemAddressFree would let the very
// next reservation land on them); they cost address space only - the physical chunks are released as soon as they are
// not needed.
namespace {
struct Arena {
    int dev = 0;
    size_t chunk = 0, n = 0;
    std::vector<hipMemGenericAllocationHandle_t> handle;
    std::vector<char> live;                             // handle exists
    std::vector<int> mapped_in;                         // index of the mapping that holds the chunk, -1 = unmapped
    struct Mapping { char* base; std::vector<uint32_t> chunks; };
    std::vector<Mapping> maps;                          // slots; base == nullptr: free slot
};
}  // namespace

// Before any chunk is unmapped or released the whole device is drained: work on ANY stream may still be using the
// addresses that are about to go away.
static void drain_device(int dev) {
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
    (void)hipDeviceSynchronize();
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
}

static std::atomic<unsigned long long> g_retired_address_bytes{0};   // reservations retired for good (discipline 3)

static void arena_unmap_slot(Arena* a, size_t slot) {
    Arena::Mapping& m = a->maps[slot];
    if (!m.base) return;
    for (size_t j = 0; j < m.chunks.size(); ++j) {
        (void)hipMemUnmap(m.base + j * a->chunk, a->chunk);
        a->mapped_in[m.chunks[j]] = -1;
    }
    g_retired_address_bytes.fetch_add((unsigned long long)(m.chunks.size() * a->chunk));   // the reservation stays: never reused
    m.base = nullptr; m.chunks.clear();
}

// An address range that has held a mapping is retired, not freed (discipline 3): a process-wide budget keeps a long-lived
// caller that places buffers again and again from running the 128 TiB address space dry (then arenas are refused and the
// caller falls back to ordinary allocations, which is what placement.py does on any arena failure).
static const unsigned long long kRetiredAddressBudget = 64ull << 40;

int fg_arena_create(int device, uint64_t bytes, uint64_t chunk_bytes, void** arena, uint64_t* chunk_out, uint32_t* chunks_out) {
    if (!arena || bytes == 0) return fail(FG_ERR_BAD_ARG, "fg_arena_create: arena and bytes > 0 required%s");
    if (g_retired_address_bytes.load() > kRetiredAddressBudget)
        return fail(FG_ERR_HIP, "fg_arena_create: the address-space budget of retired reservations (64 TiB) is spent%s");
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    hipError_t err = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (err != hipSuccess || gran == 0) return fail(FG_ERR_HIP, "hipMemGetAllocationGranularity failed: %s", hipGetErrorString(err));
    size_t chunk = chunk_bytes ? (size_t)chunk_bytes : ((size_t)1 << 30);
    chunk = (chunk + gran - 1) / gran * gran;
    Arena* a = new Arena();
    a->dev = device; a->chunk = chunk; a->n = ((size_t)bytes + chunk - 1) / chunk;
    a->handle.resize(a->n); a->live.assign(a->n, 0); a->mapped_in.assign(a->n, -1);
    for (size_t i = 0; i < a->n; ++i) {
        err = hipMemCreate(&a->handle[i], chunk, &prop, 0);
        if (err != hipSuccess) break;
        a->live[i] = 1;
    }
    // Every chunk is mapped once, in index order, and unmapped again (the reservation is retired, see above): a chunk has
    // its physical place from here on, whatever selection it is mapped in later.  (Where that place is, the driver
    // decides: holding memory between two groups of chunks while they are created does NOT put them reliably apart -
    // profiles/r04_place/spread_rule*.txt - which is why selections are timed, not composed by rule.)
    void* va = nullptr;
    size_t mapped = 0;
    if (err == hipSuccess) err = hipMemAddressReserve(&va, a->n * chunk, 0, nullptr, 0);
    if (err == hipSuccess) {
        for (; mapped < a->n; ++mapped) {
            err = hipMemMap((char*)va + mapped * chunk, chunk, 0, a->handle[mapped], 0);
            if (err != hipSuccess) break;
        }
        if (err == hipSuccess) {
            hipMemAccessDesc desc;
            memset(&desc, 0, sizeof(desc));
            desc.location.type = hipMemLocationTypeDevice;
            desc.location.id = device;
            desc.flags = hipMemAccessFlagsProtReadWrite;
            err = hipMemSetAccess(va, a->n * chunk, &desc, 1);
        }
        drain_device(device);
        for (size_t i = 0; i < mapped; ++i) (void)hipMemUnmap((char*)va + i * chunk, chunk);
        g_retired_address_bytes.fetch_add((unsigned long long)(a->n * chunk));
    }
    if (err != hipSuccess) {
        const int rc = fail(FG_ERR_HIP, "fg_arena_create: %s", hipGetErrorString(err));
        (void)hipGetLastError();
        for (size_t i = 0; i < a->n; ++i) if (a->live[i]) (void)hipMemRelease(a->handle[i]);
        delete a;
        return rc;
    }
    *arena = a;
    if (chunk_out) *chunk_out = chunk;
    if (chunks_out) *chunks_out = (uint32_t)a->n;
    return FG_OK;
}

int fg_arena_map(void* arena, const uint32_t* chunk_index, uint32_t count, void** base) {
    Arena* a = (Arena*)arena;
    if (!a || !chunk_index || !base || count == 0) return fail(FG_ERR_BAD_ARG, "fg_arena_map: arena, indices and base required%s");
    std::vector<char> seen(a->n, 0);
    for (uint32_t j = 0; j < count; ++j) {
        const uint32_t c = chunk_index[j];
        if (c >= a->n || !a->live[c]) return fail(FG_ERR_BAD_ARG, "fg_arena_map: chunk index out of range or released%s");
        if (a->mapped_in[c] >= 0 || seen[c]) return fail(FG_ERR_BAD_ARG, "fg_arena_map: a chunk is mapped at one address at a time%s");
        seen[c] = 1;
    }
    if (g_retired_address_bytes.load() > kRetiredAddressBudget)
        return fail(FG_ERR_HIP, "fg_arena_map: the address-space budget of retired reservations (64 TiB) is spent%s");
    void* va = nullptr;
    hipError_t err = hipMemAddressReserve(&va, (size_t)count * a->chunk, 0, nullptr, 0);
    if (err != hipSuccess) return fail(FG_ERR_HIP, "hipMemAddressReserve failed: %s", hipGetErrorString(err));
    uint32_t done = 0;
    for (; done < count; ++done) {
        err = hipMemMap((char*)va + (size_t)done * a->chunk, a->chunk, 0, a->handle[chunk_index[done]], 0);
        if (err != hipSuccess) break;
    }
    if (err == hipSuccess) {
        hipMemAccessDesc desc;
        memset(&desc, 0, sizeof(desc));
        desc.location.type = hipMemLocationTypeDevice;
        desc.location.id = a->dev;
        desc.flags = hipMemAccessFlagsProtReadWrite;
        err = hipMemSetAccess(va, (size_t)count * a->chunk, &desc, 1);
    }
    if (err != hipSuccess) {
        const int rc = fail(FG_ERR_HIP, "fg_arena_map: %s", hipGetErrorString(err));
        (void)hipGetLastError();
        for (uint32_t j = 0; j < done; ++j) (void)hipMemUnmap((char*)va + (size_t)j * a->chunk, a->chunk);
        g_retired_address_bytes.fetch_add((unsigned long long)count * a->chunk);   // held a mapping: retired, not freed
        return rc;
    }
    size_t slot = a->maps.size();
    for (size_t k = 0; k < a->maps.size(); ++k) if (!a->maps[k].base) { slot = k; break; }
    if (slot == a->maps.size()) a->maps.push_back(Arena::Mapping{nullptr, {}});
    a->maps[slot].base = (char*)va;
    a->maps[slot].chunks.assign(chunk_index, chunk_index + count);
    for (uint32_t j = 0; j < count; ++j) a->mapped_in[chunk_index[j]] = (int)slot;
    drain_device(a->dev);                               // the page-table work is complete before the caller launches into it
    *base = va;
    return FG_OK;
}

int fg_arena_unmap(void* arena, void* base) {
    Arena* a = (Arena*)arena;
    if (!a || !base) return fail(FG_ERR_BAD_ARG, "fg_arena_unmap: arena and base required%s");
    for (size_t k = 0; k < a->maps.size(); ++k)
        if (a->maps[k].base == (char*)base) {
            drain_device(a->dev);
            arena_unmap_slot(a, k);
            return FG_OK;
        }
    return fail(FG_ERR_BAD_ARG, "fg_arena_unmap: not a mapping of this arena%s");
}

// Shrinks a mapping to the window [first, first + count) of its chunks: the chunks outside are unmapped (their part of the
// reservation is retired), the window stays where it is.  The probe keeps its winner this way when it lies in the mapping it
// timed last, instead of mapping it once more at a fresh address.
int fg_arena_keep_window(void* arena, void* base, uint32_t first, uint32_t count, void** new_base) {
    Arena* a = (Arena*)arena;
    if (!a || !base || !new_base || count == 0) return fail(FG_ERR_BAD_ARG, "fg_arena_keep_window: arena, base, new_base and count > 0 required%s");
    for (size_t k = 0; k < a->maps.size(); ++k) {
        Arena::Mapping& m = a->maps[k];
        if (m.base != (char*)base) continue;
        if ((size_t)first + count > m.chunks.size()) return fail(FG_ERR_BAD_ARG, "fg_arena_keep_window: window beyond the mapping%s");
        drain_device(a->dev);
        size_t gone = 0;
        for (size_t j = 0; j < m.chunks.size(); ++j) {
            if (j >= first && j < (size_t)first + count) continue;
            (void)hipMemUnmap(m.base + j * a->chunk, a->chunk);
            a->mapped_in[m.chunks[j]] = -1;
            ++gone;
        }
        g_retired_address_bytes.fetch_add((unsigned long long)(gone * a->chunk));
        std::vector<uint32_t> kept(m.chunks.begin() + first, m.chunks.begin() + first + count);
        m.base += (size_t)first * a->chunk;
        m.chunks.swap(kept);
        *new_base = m.base;
        return FG_OK;
    }
    return fail(FG_ERR_BAD_ARG, "fg_arena_keep_window: not a mapping of this arena%s");
}

uint64_t fg_arena_retired_address_bytes(void) { return (uint64_t)g_retired_address_bytes.load(); }

int fg_arena_trim(void* arena) {
    Arena* a = (Arena*)arena;
    if (!a) return fail(FG_ERR_BAD_ARG, "fg_arena_trim: arena is NULL%s");
    for (size_t i = 0; i < a->n; ++i)
        if (a->live[i] && a->mapped_in[i] < 0) { (void)hipMemRelease(a->handle[i]); a->live[i] = 0; }
    return FG_OK;
}

int fg_arena_destroy(void* arena) {
    Arena* a = (Arena*)arena;
    if (!a) return FG_OK;
    drain_device(a->dev);
    for (size_t k = 0; k < a->maps.size(); ++k) arena_unmap_slot(a, k);
    for (size_t i = 0; i < a->n; ++i) if (a->live[i]) (void)hipMemRelease(a->handle[i]);
    delete a;
    return FG_OK;
}

const char* fg_last_error(void) { return g_err; }

int fg_launch_device(void* stream, const void* data) { return resolve_device(stream, data); }

int fg_kernel_config(int N, int* threads, int* envs_per_wg, int* lds_bytes) {
    Geometry g;
    if (!geometry_for(N, &g)) return fail(FG_ERR_UNSUPPORTED_N, "N must be in [2, 1024]%s");
    if (threads) *threads = g.T;
    if (envs_per_wg) *envs_per_wg = g.E;
    if (lds_bytes) *lds_bytes = g.lds;
    return FG_OK;
}

int64_t fg_step_hd_bytes(int N) { return 24LL * N * N + 53LL * N + 16LL; }

// fg_step_hd in two halves: argument checks + the launch description, and the dispatch of a checked description
static int step_hd_describe(const FgParams* params, int B, int N,
                            float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                            const float* act, float* ideal_shape, float* ideal_vel, int32_t* step,
                            float* obs, float* reward, float* indiv_reward, uint8_t* done,
                            int32_t* near_lm, int32_t* near_ag, int32_t* hd_idx, Args* out) {
    int rc = check_params(params);
    if (rc) return rc;
    if (B < 0) return fail(FG_ERR_BAD_ARG, "B must be >= 0%s");
    if (N < 3 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "formation_hd_env needs 3 <= N <= 1024%s");
    if (B > 0 && (!pos_x || !pos_y || !vel_x || !vel_y || !act || !ideal_shape || !ideal_vel || !step || !obs || !reward))
        return fail(FG_ERR_BAD_ARG, "fg_step_hd: a required pointer is NULL%s");
    if (((uintptr_t)obs & 15u) || ((uintptr_t)act & 7u) || ((uintptr_t)ideal_shape & 7u) || ((uintptr_t)ideal_vel & 7u))
        return fail(FG_ERR_ALIGNMENT, "obs must be 16-byte, act/ideal_shape/ideal_vel 8-byte aligned%s");
    Args a; memset(&a, 0, sizeof(a));
    a.p = *params; a.B = B; a.N = N; a.inv_n = 1.0f / (float)N; a.K = 1; a.obs_every = 1; a.do_phys = 1; a.do_post = 1;
    if ((rc = set_obs_pitch(&a)) != FG_OK) return rc;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y; a.act = act;
    a.shape = ideal_shape; a.ivel = ideal_vel; a.step = step;
    a.obs = obs; a.rew = reward; a.indiv = indiv_reward; a.done = done;
    a.near_lm = near_lm; a.near_ag = near_ag; a.hd_idx = hd_idx;
    *out = a;
    return FG_OK;
}

static int step_hd_dispatch(const Args& a, hipStream_t stream) {
    if (a.B == 0) return FG_OK;                       // an empty batch is a no-op (e.g. a rank that owns no envs)
    const bool plain = !world_options_set(a.p) && !a.near_lm && !a.near_ag && !a.hd_idx;
    // 243 agents, >= 4096 envs: pipeline over env batches inside the launch (no index outputs, no World options):
    // 1.85-2.2 ms vs 2.1-2.3 ms at 243 x 8192 (round 1; 1.75 ms on placed buffers).
    if (a.N == 243 && a.B >= FG_WIDE243_MIN_B && plain) return launch_wide<243, 0>(a, stream);
    // 81 agents: the pipelined kernel pays from 16 env batches per workgroup on (81 x 16384: 419 vs 448 us; 81 x 12288 equal,
    // 81 x 2048 62 vs 56: profiles/r03_step/pipelined_single_step_81.txt)
    if (a.N == 81 && a.B >= FG_WIDE81_MIN_B && plain) return launch_wide<81, 0>(a, stream);
    return launch_step(a, stream);
}

int fg_step_hd(const FgParams* params, int B, int N,
               float* pos_x, float* pos_y, float* vel_x, float* vel_y,
               const float* act, float* ideal_shape, float* ideal_vel, int32_t* step,
               float* obs, float* reward, float* indiv_reward, uint8_t* done,
               int32_t* near_lm, int32_t* near_ag, int32_t* hd_idx, void* stream) {
    const DeviceGuard device_guard(stream, pos_x);
    Args a;
    const int rc = step_hd_describe(params, B, N, pos_x, pos_y, vel_x, vel_y, act, ideal_shape, ideal_vel, step,
                                    obs, reward, indiv_reward, done, near_lm, near_ag, hd_idx, &a);
    if (rc) return rc;
    return step_hd_dispatch(a, (hipStream_t)stream);
}

// A step loop that re-uses its buffers launches the same description again and again with only the RNG offset moving: the
// plan keeps the checked description, so the per-step host work is one two-argument call and the kernel launch (a ctypes
// call with 19 arguments cost 1.7 us per env.step on top of the 7.7 us kernel at 9 x 4096: VERDICT r4).
namespace { struct StepPlan { Args a; hipStream_t stream; int device; }; }

int fg_step_hd_plan(const FgParams* params, int B, int N,
                    float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                    const float* act, float* ideal_shape, float* ideal_vel, int32_t* step,
                    float* obs, float* reward, float* indiv_reward, uint8_t* done,
                    int32_t* near_lm, int32_t* near_ag, int32_t* hd_idx, void* stream, void** plan) {
    if (!plan) return fail(FG_ERR_BAD_ARG, "fg_step_hd_plan: plan is NULL%s");
    Args a;
    const int rc = step_hd_describe(params, B, N, pos_x, pos_y, vel_x, vel_y, act, ideal_shape, ideal_vel, step,
                                    obs, reward, indiv_reward, done, near_lm, near_ag, hd_idx, &a);
    if (rc) return rc;
    StepPlan* sp = new StepPlan();
    sp->a = a; sp->stream = (hipStream_t)stream;
    sp->device = visible_devices() < 2 ? -1 : resolve_device(stream, pos_x);
    *plan = sp;
    return FG_OK;
}

int fg_plan_launch(void* plan, uint64_t rng_offset) {
    StepPlan* sp = (StepPlan*)plan;
    if (!sp) return fail(FG_ERR_BAD_ARG, "fg_plan_launch: plan is NULL%s");
    int prev = -1;
    bool switched = false;
    if (sp->device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != sp->device) switched = hipSetDevice(sp->device) == hipSuccess;
    sp->a.p.rng_offset = rng_offset;
    const int rc = step_hd_dispatch(sp->a, sp->stream);
    if (switched) (void)hipSetDevice(prev);
    return rc;
}

int fg_plan_destroy(void* plan) {
    delete (StepPlan*)plan;
    return FG_OK;
}

// the tail of fg_rollout_hd: which kernel runs a checked K-step description (also walked dry by fg_describe_launch)
static int rollout_hd_dispatch(const Args& a, hipStream_t st) {
    int rc = FG_OK;
    // K >= 2 at the specialised agent counts: producer / writer pipelined kernels.  World options (walls, max_speed,
    // accel, u_noise) exist only in step_kernel's OPTS instantiation, whose K-loop runs the rollout then.
    if (a.K == 1 && !world_options_set(a.p)) {                                                       // as fg_step_hd
        if (a.N == 243 && a.B >= FG_WIDE243_MIN_B) return launch_wide<243, 0>(a, st);
        if (a.N == 81 && a.B >= FG_WIDE81_MIN_B) return launch_wide<81, 0>(a, st);
    }
    if (a.K >= 2 && !world_options_set(a.p) && launch_pipelined(a, 0, st, &rc)) return rc;
    return launch_step(a, st);
}

static int launch_policy_state(int B, int N, const FgPolicyLevels& pl, const float* px, const float* py,
                               const float* shape, const float* ivel, float* act, hipStream_t st);

// ... and of fg_rollout_hd_policy (a.pl, a.act_out set)
static int rollout_hd_policy_dispatch(const Args& a, int per_layer, hipStream_t st) {
    int rc = FG_OK;
    // N = per^L up to 243 agents (per 2, 3, 4, 5, 8): the controller runs inside the pipelined kernels, ONE launch
    if (!world_options_set(a.p) && launch_pipelined(a, per_layer, st, &rc)) return rc;
    // everything else: K times (controller launch, single-step launch) chained on the stream - the same device
    // function on the same state, so the results equal the in-kernel loop's and fg_policy_bfs on the written rows
    const size_t bn = (size_t)a.B * a.N;
    for (int k = 0; k < a.K; ++k) {
        float* act_k = a.act_out + (size_t)k * bn * 2;
        rc = launch_policy_state(a.B, a.N, a.pl, a.px, a.py, a.shape, a.ivel, act_k, st);
        if (rc) return rc;
        Args s1 = a;
        s1.K = 1; s1.obs_every = 1; s1.act = act_k; s1.act_out = nullptr;
        s1.p.rng_offset = a.p.rng_offset + (uint64_t)k;
        s1.rew = a.rew + (size_t)k * bn;
        s1.indiv = a.indiv ? a.indiv + (size_t)k * bn : nullptr;
        s1.done = a.done ? a.done + (size_t)k * bn : nullptr;
        s1.obs = (a.obs && (k + 1) % a.obs_every == 0)
                     ? a.obs + (size_t)(k / a.obs_every) * (size_t)a.B * (size_t)a.obs_pitch * 2 : nullptr;
        rc = launch_step(s1, st);
        if (rc) return rc;
        if (g_describe && k == 0 && a.K > 1) { describe("x %d chained; ", a.K); break; }   // dry run: one round says it all
    }
    return FG_OK;
}

int fg_physics_step(const FgParams* params, int B, int N,
                    float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                    const float* act, void* stream) {
    const DeviceGuard device_guard(stream, pos_x);
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
    if ((rc = set_obs_pitch(&a)) != FG_OK) return rc;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y; a.act = act;
    return launch_step(a, (hipStream_t)stream);
}

int fg_observe_hd(const FgParams* params, int B, int N,
                  const float* pos_x, const float* pos_y, const float* vel_x, const float* vel_y,
                  const float* ideal_shape, const float* ideal_vel, const int32_t* step,
                  float* obs, float* reward, float* indiv_reward, uint8_t* done,
                  int32_t* near_lm, int32_t* near_ag, int32_t* hd_idx, void* stream) {
    const DeviceGuard device_guard(stream, pos_x);
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
    if ((rc = set_obs_pitch(&a)) != FG_OK) return rc;
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
    const DeviceGuard device_guard(stream, pos_x);
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
    if ((rc = set_obs_pitch(&a)) != FG_OK) return rc;
    a.do_phys = 1; a.do_post = 1;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y; a.act = act_seq;
    a.shape = ideal_shape; a.ivel = ideal_vel; a.step = step;
    a.obs = obs_seq; a.rew = reward_seq; a.indiv = indiv_seq; a.done = done_seq;
    return rollout_hd_dispatch(a, (hipStream_t)stream);
}

int fg_reset_hd(const FgParams* params, int B, int N, const uint8_t* mask,
                float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                float* ideal_shape, float* ideal_vel, int32_t* step, void* stream) {
    const DeviceGuard device_guard(stream, pos_x);
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
                           const float* act, float* landmarks, float* obst_pos, float* obst_vel,
                           int32_t* step, float* obs, float* reward, float* indiv_reward, uint8_t* done,
                           int32_t* near_ag, void* stream, int K = 1, int obs_every = 1) {
    const DeviceGuard device_guard(stream, pos_x);
    int rc = check_params(params);
    if (rc) return rc;
    if (!sc) return fail(FG_ERR_BAD_ARG, "scenario descriptor is NULL%s");
    if (K == 0) return FG_OK;
    if (K < 0 || obs_every < 1) return fail(FG_ERR_BAD_ARG, "K >= 0 and obs_every >= 1 required%s");
    if (K > 1 && !do_physics) return fail(FG_ERR_BAD_ARG, "a multi-step launch steps the physics%s");
    if (params->comm_state)
        return fail(FG_ERR_BAD_ARG, "comm_state is honoured by the formation_hd_env entry points only%s");
    const int L = sc->num_landmarks, M = sc->num_obstacles;
    if (sc->kind < FG_SCN_BASIC || sc->kind > FG_SCN_OBSTACLE) return fail(FG_ERR_BAD_ARG, "unknown scenario kind%s");
    if (B == 0) return FG_OK;
    if (B < 0 || L <= 0 || M < 0) return fail(FG_ERR_BAD_ARG, "B >= 0, L > 0, M >= 0 required%s");
    if (N < 2 || N + M > FG_MAX_AGENTS || L > 1024)
        return fail(FG_ERR_UNSUPPORTED_N, "scenario kernel needs 2 <= N, N + M <= 1024, L <= 1024%s");
    if (sc->kind == FG_SCN_PARTIAL && (sc->num_obs < 0 || sc->num_obs > 1024)) return fail(FG_ERR_BAD_ARG, "bad num_obs%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !landmarks || !obs || (do_physics && (!act || !reward)) ||
        (M > 0 && (!obst_pos || !obst_vel)))
        return fail(FG_ERR_BAD_ARG, "scenario step: a required pointer is NULL%s");
    if (((uintptr_t)obs & 7u) || ((uintptr_t)landmarks & 7u) || (act && ((uintptr_t)act & 7u)) ||
        ((uintptr_t)obst_pos & 7u) || ((uintptr_t)obst_vel & 7u))
        return fail(FG_ERR_ALIGNMENT, "obs/landmarks/act/obstacle buffers must be 8-byte aligned%s");
    ScnArgs a; memset(&a, 0, sizeof(a));
    a.p = *params; a.sc = *sc; a.B = B; a.N = N; a.do_phys = do_physics ? 1 : 0; a.K = K; a.obs_every = obs_every;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y; a.act = act; a.lm = landmarks;
    a.opos = obst_pos; a.ovel = obst_vel; a.step = step;
    a.obs = obs; a.rew = reward; a.indiv = indiv_reward; a.done = done; a.near_ag = near_ag;
    a.inv_n = (float)(1.0 / (double)N); a.inv_l = (float)(1.0 / (double)L);
    a.coll_scale = (float)((double)params->collide_thresh / (double)params->dist_min);
    hipStream_t st = (hipStream_t)stream;
    if (sc->variant != 1 && !params->agent_props) {        // (per-agent tables: the run-time-count kernel)
        // the reference's own shapes: one env per lane, every count a compile-time constant (fg_scn_lane_kernel.hpp)
        const int nbr = sc->kind == FG_SCN_PARTIAL ? sc->num_obs : N - 1;
        bool launched = false;
        const double obs_bytes = (double)((K > 1 ? K : 1) / obs_every) * B * N * 4.0 * scn_obs_dim(sc->kind, N, L, M, nbr);
#define FG_SCN_LANE(KIND, NN, LL, MM, NBR, WIDE)                                                                        \
        if (!launched && sc->kind == KIND && N == NN && L == LL && M == MM && nbr == NBR) {                                 \
            const int rc = (WIDE && scn_lane_wide<KIND>(B, obs_bytes)) ? launch_scn_lane<KIND, NN, LL, MM, NBR, (WIDE ? 4 : 1)>(a, st)  \
                                                            : launch_scn_lane<KIND, NN, LL, MM, NBR, 1>(a, st);            \
            if (g_describe) return rc;                                                                                      \
            launched = true;                                                                                                \
        }
        // (WIDE: the 256-env form exists for this shape, see scn_lane_wide)
        FG_SCN_LANE(FG_SCN_BASIC, 3, 3, 0, 2, true)            // basic_formation_env.py:7
        FG_SCN_LANE(FG_SCN_PARTIAL, 5, 5, 0, 3, true)          // formation_hd_partial_env.py:15
        FG_SCN_LANE(FG_SCN_RANGE, 4, 4, 0, 3, true)            // formation_hd_partial_range_env.py:15
        FG_SCN_LANE(FG_SCN_OBSTACLE, 4, 4, 3, 3, false)        // formation_hd_obs_env.py:14
        FG_SCN_LANE(FG_SCN_PARTIAL, 3, 5, 0, 3, false)         // make_env(name) passes its own default num_agents = 3 (__init__.py:6-11)
        FG_SCN_LANE(FG_SCN_RANGE, 3, 4, 0, 2, false)
        FG_SCN_LANE(FG_SCN_OBSTACLE, 3, 4, 3, 2, false)
#undef FG_SCN_LANE
        if (launched) {
            const hipError_t err = hipGetLastError();
            if (err != hipSuccess) return fail(FG_ERR_HIP, "scenario launch failed: %s", hipGetErrorString(err));
            return FG_OK;
        }
    }
    constexpr int FG_SCN_T = 64;      // threads per workgroup of the scenario kernel up to 64 entities per env
    const int G = pow2ceil(N + M) < 4 ? 4 : pow2ceil(N + M);
    const int E = G <= 64 ? FG_SCN_T / G : 1;         // more than 64 entities: one env per workgroup of G threads
    const int grid = (B + E - 1) / E;
    int lds = (E * (2 * (N + M) + L) + (G > 64 ? 32 : 0)) * (int)sizeof(float2);
    {   // observation rows composed in LDS and streamed out contiguously when the workgroup's block fits
        const int nbr = (sc->kind == FG_SCN_PARTIAL) ? sc->num_obs : (N - 1);
        const long long D = 2 + (sc->kind == FG_SCN_BASIC ? 2 : 0) + 2LL * L + 2LL * M + 2LL * nbr + 2LL * (N - 1);
        const long long stage = (long long)E * N * D * (long long)sizeof(float);
        a.stage = (lds + stage <= 48 * 1024) ? 1 : 0;
        if (a.stage) lds += (int)stage;
    }
    if (describe("scn_kernel<%d,%d> grid %d lds %d stage %d; ", G, G <= 64 ? FG_SCN_T : G, grid, lds, a.stage)) return FG_OK;
    if (G == 4) hipLaunchKernelGGL((scn_kernel<4, FG_SCN_T>), dim3(grid), dim3(FG_SCN_T), lds, st, a);
    else if (G == 8) hipLaunchKernelGGL((scn_kernel<8, FG_SCN_T>), dim3(grid), dim3(FG_SCN_T), lds, st, a);
    else if (G == 16) hipLaunchKernelGGL((scn_kernel<16, FG_SCN_T>), dim3(grid), dim3(FG_SCN_T), lds, st, a);
    else if (G == 32) hipLaunchKernelGGL((scn_kernel<32, FG_SCN_T>), dim3(grid), dim3(FG_SCN_T), lds, st, a);
    else if (G == 64) hipLaunchKernelGGL((scn_kernel<64, FG_SCN_T>), dim3(grid), dim3(FG_SCN_T), lds, st, a);
    else if (G == 128) hipLaunchKernelGGL((scn_kernel<128, 128>), dim3(grid), dim3(128), lds, st, a);
    else if (G == 256) hipLaunchKernelGGL((scn_kernel<256, 256>), dim3(grid), dim3(256), lds, st, a);
    else if (G == 512) hipLaunchKernelGGL((scn_kernel<512, 512>), dim3(grid), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((scn_kernel<1024, 1024>), dim3(grid), dim3(1024), lds, st, a);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "scenario launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

static int launch_mt_reset(int B, int N, const uint8_t* mask, uint32_t* mt_state,
                           float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                           float* ideal_shape, float* ideal_vel, float* landmark_pos, int32_t* step,
                           int world_length, float* obs, long long obs_env_pitch, void* stream) {
    const DeviceGuard device_guard(stream, pos_x);
    if (B == 0) return FG_OK;                         // an empty batch is a no-op (e.g. a rank that owns no envs)
    if (B < 0) return fail(FG_ERR_BAD_ARG, "B must be >= 0%s");
    if (N < 2 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "N must be in [2, 1024]%s");
    if (!mt_state || !pos_x || !pos_y || !vel_x || !vel_y || !ideal_shape || !ideal_vel)
        return fail(FG_ERR_BAD_ARG, "fg_reset_hd_mt: a required pointer is NULL%s");
    const long long nenv = 6LL * N * N;
    const long long pitch = obs_env_pitch ? obs_env_pitch : nenv;
    if (obs && (pitch < nenv || (pitch & 1))) return fail(FG_ERR_BAD_ARG, "obs_env_pitch must be 0 or an even number of floats >= 6 N^2%s");
    if ((uintptr_t)obs & 7u) return fail(FG_ERR_ALIGNMENT, "obs must be 8-byte aligned%s");
    const int lds = (624 + ((8 * N + 4 + 1) & ~1)) * (int)sizeof(uint32_t) + 2 * (int)sizeof(double) +
                    (2 * N + 1) * (int)sizeof(float2);
    hipLaunchKernelGGL(mt_reset_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, B, N, mask, mt_state,
                       pos_x, pos_y, vel_x, vel_y, ideal_shape, ideal_vel, landmark_pos, step, world_length, obs, pitch / 2);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "mt reset launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

int fg_reset_hd_mt(int B, int N, const uint8_t* mask, uint32_t* mt_state,
                   float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                   float* ideal_shape, float* ideal_vel, float* landmark_pos, int32_t* step, void* stream) {
    return launch_mt_reset(B, N, mask, mt_state, pos_x, pos_y, vel_x, vel_y, ideal_shape, ideal_vel, landmark_pos, step,
                           0, nullptr, 0, stream);
}

int fg_reset_hd_mt_done(int B, int N, int world_length, uint32_t* mt_state,
                        float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                        float* ideal_shape, float* ideal_vel, float* landmark_pos, int32_t* step,
                        float* obs, int64_t obs_env_pitch, void* stream) {
    if (B == 0) return FG_OK;
    if (world_length <= 0 || !step) return fail(FG_ERR_BAD_ARG, "fg_reset_hd_mt_done: world_length > 0 and step required%s");
    return launch_mt_reset(B, N, nullptr, mt_state, pos_x, pos_y, vel_x, vel_y, ideal_shape, ideal_vel, landmark_pos, step,
                           world_length, obs, (long long)obs_env_pitch, stream);
}

int fg_reset_scenario_mt(const FgScenario* scenario, int B, int N, const uint8_t* mask, int world_length, uint32_t* mt_state,
                         float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                         float* landmarks, float* obst_pos, float* obst_vel, int32_t* step, void* stream) {
    const DeviceGuard device_guard(stream, pos_x);
    if (!scenario) return fail(FG_ERR_BAD_ARG, "scenario descriptor is NULL%s");
    const int L = scenario->num_landmarks, M = scenario->num_obstacles;
    if (B == 0) return FG_OK;
    if (B < 0 || L <= 0 || M < 0) return fail(FG_ERR_BAD_ARG, "B >= 0, L > 0, M >= 0 required%s");
    if (N < 2 || N + M > FG_MAX_AGENTS || L > 1024)
        return fail(FG_ERR_UNSUPPORTED_N, "scenario kernel needs 2 <= N, N + M <= 1024, L <= 1024%s");
    if (!mt_state || !pos_x || !pos_y || !vel_x || !vel_y || !landmarks || (M > 0 && (!obst_pos || !obst_vel)))
        return fail(FG_ERR_BAD_ARG, "fg_reset_scenario_mt: a required pointer is NULL%s");
    if (!mask && world_length > 0 && !step) return fail(FG_ERR_BAD_ARG, "fg_reset_scenario_mt: the done rule needs the step counters%s");
    if (((uintptr_t)landmarks & 7u) || ((uintptr_t)obst_pos & 7u) || ((uintptr_t)obst_vel & 7u))
        return fail(FG_ERR_ALIGNMENT, "landmarks / obstacle buffers must be 8-byte aligned%s");
    const int lds = (624 + 4 * (N + L + M)) * (int)sizeof(uint32_t);
    hipLaunchKernelGGL(mt_reset_scn_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, B, N, L, M, mask, mt_state,
                       pos_x, pos_y, vel_x, vel_y, reinterpret_cast<float2*>(landmarks), reinterpret_cast<float2*>(obst_pos),
                       reinterpret_cast<float2*>(obst_vel), step, world_length, scenario->obstacle_vx, scenario->obstacle_vy);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "mt reset launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

int fg_update_comm(const FgParams* params, int B, int N, const float* action_c, float* comm_state, void* stream) {
    const DeviceGuard device_guard(stream, comm_state);
    int rc = check_params(params);
    if (rc) return rc;
    if (B == 0) return FG_OK;
    if (B < 0) return fail(FG_ERR_BAD_ARG, "B must be >= 0%s");
    if (N < 1 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "N must be in [1, 1024]%s");
    if (!action_c || !comm_state) return fail(FG_ERR_BAD_ARG, "fg_update_comm: a required pointer is NULL%s");
    if (((uintptr_t)action_c & 7u) || ((uintptr_t)comm_state & 7u))
        return fail(FG_ERR_ALIGNMENT, "action_c and comm_state must be 8-byte aligned%s");
    const long long count = (long long)B * N;
    hipLaunchKernelGGL(update_comm_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       *params, B, N, reinterpret_cast<const float2*>(action_c), reinterpret_cast<float2*>(comm_state));
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "comm launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

int fg_update_comm_dim(const FgParams* params, int B, int N, int dim_c, const float* action_c, float* comm_state, void* stream) {
    if (dim_c == 2) return fg_update_comm(params, B, N, action_c, comm_state, stream);
    const DeviceGuard device_guard(stream, comm_state);
    int rc = check_params(params);
    if (rc) return rc;
    if (B == 0 || dim_c == 0) return FG_OK;
    if (B < 0 || dim_c < 0 || dim_c > 4096) return fail(FG_ERR_BAD_ARG, "B >= 0 and 0 <= dim_c <= 4096 required%s");
    if (N < 1 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "N must be in [1, 1024]%s");
    if (!action_c || !comm_state) return fail(FG_ERR_BAD_ARG, "fg_update_comm_dim: a required pointer is NULL%s");
    if (((uintptr_t)action_c & 3u) || ((uintptr_t)comm_state & 3u))
        return fail(FG_ERR_ALIGNMENT, "action_c and comm_state must be 4-byte aligned%s");
    const long long count = (long long)B * N * ((dim_c + 1) / 2);
    hipLaunchKernelGGL(update_comm_dim_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       *params, B, N, dim_c, action_c, comm_state);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "comm launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

int fg_step_scenario(const FgParams* params, const FgScenario* scenario, int B, int N, int do_physics,
                     float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                     const float* act, float* landmarks, float* obst_pos, float* obst_vel,
                     int32_t* step, float* obs, float* reward, float* indiv_reward, uint8_t* done,
                     void* stream) {
    return launch_scenario(params, scenario, B, N, do_physics, pos_x, pos_y, vel_x, vel_y, act, landmarks,
                           obst_pos, obst_vel, step, obs, reward, indiv_reward, done, nullptr, stream);
}

int fg_step_basic(const FgParams* params, int B, int N, int L, int do_physics,
                  float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                  const float* act, float* landmarks, int32_t* step,
                  float* obs, float* reward, float* indiv_reward, uint8_t* done,
                  int32_t* near_ag, void* stream) {
    FgScenario sc; memset(&sc, 0, sizeof(sc));
    sc.kind = FG_SCN_BASIC; sc.num_landmarks = L; sc.penalty = 1.0f;
    return launch_scenario(params, &sc, B, N, do_physics, pos_x, pos_y, vel_x, vel_y, act, landmarks,
                           nullptr, nullptr, step, obs, reward, indiv_reward, done, near_ag, stream);
}

int fg_rollout_scenario(const FgParams* params, const FgScenario* scenario, int B, int N, int K,
                        float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                        const float* act_seq, float* landmarks, float* obst_pos, float* obst_vel,
                        int32_t* step, float* obs_seq, float* reward_seq, float* indiv_seq, uint8_t* done_seq,
                        int32_t* near_ag_seq, int obs_every, void* stream) {
    return launch_scenario(params, scenario, B, N, 1, pos_x, pos_y, vel_x, vel_y, act_seq, landmarks,
                           obst_pos, obst_vel, step, obs_seq, reward_seq, indiv_seq, done_seq, near_ag_seq, stream,
                           K, obs_every < 1 ? 1 : obs_every);
}

int fg_reset_scenario(const FgParams* params, const FgScenario* scenario, int B, int N, const uint8_t* mask,
                      float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                      float* landmarks, float* obst_pos, float* obst_vel, int32_t* step, void* stream) {
    const DeviceGuard device_guard(stream, pos_x);
    int rc = check_params(params);
    if (rc) return rc;
    if (!scenario) return fail(FG_ERR_BAD_ARG, "scenario descriptor is NULL%s");
    const int L = scenario->num_landmarks, M = scenario->num_obstacles;
    if (scenario->kind < FG_SCN_BASIC || scenario->kind > FG_SCN_OBSTACLE) return fail(FG_ERR_BAD_ARG, "unknown scenario kind%s");
    if (B == 0) return FG_OK;
    if (B < 0 || L <= 0 || M < 0) return fail(FG_ERR_BAD_ARG, "B >= 0, L > 0, M >= 0 required%s");
    if (N < 2 || N + M > FG_MAX_AGENTS || L > 1024)
        return fail(FG_ERR_UNSUPPORTED_N, "scenario kernel needs 2 <= N, N + M <= 1024, L <= 1024%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !landmarks || (M > 0 && (!obst_pos || !obst_vel)))
        return fail(FG_ERR_BAD_ARG, "fg_reset_scenario: a required pointer is NULL%s");
    if (((uintptr_t)landmarks & 7u) || ((uintptr_t)obst_pos & 7u) || ((uintptr_t)obst_vel & 7u))
        return fail(FG_ERR_ALIGNMENT, "landmarks / obstacle buffers must be 8-byte aligned%s");
    const long long count = (long long)B * (N + L + M);
    hipLaunchKernelGGL(scn_reset_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       *params, *scenario, B, N, mask, pos_x, pos_y, vel_x, vel_y, reinterpret_cast<float2*>(landmarks),
                       reinterpret_cast<float2*>(obst_pos), reinterpret_cast<float2*>(obst_vel), step);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "scenario reset launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

// the controller from the simulator state: one launch, shared by fg_policy_bfs_state and the chained closed loop
static int launch_policy_state(int B, int N, const FgPolicyLevels& pl, const float* px, const float* py,
                               const float* shape, const float* ivel, float* act, hipStream_t st) {
    const int lpe = N <= 16 ? 16 : N <= 32 ? 32 : N <= 64 ? 64 : 256;      // lanes per env
    const int E = 256 / lpe;
    const int grid = (B + E - 1) / E;
    const int lds = E * policy_block_units(N) * (int)sizeof(float2);        // <= 56 KiB
    if (describe("policy_state_kernel<%d> grid %d lds %d; ", pl.per, grid, lds)) return FG_OK;
#define FG_POLICY(PER) case PER: hipLaunchKernelGGL((policy_state_kernel<PER>), dim3(grid), dim3(256), lds, st, B, N, lpe, pl, \
                                                   px, py, shape, ivel, act); break;
    switch (pl.per) { FG_POLICY(2) FG_POLICY(3) FG_POLICY(4) FG_POLICY(5) FG_POLICY(6) FG_POLICY(7) FG_POLICY(8) }
#undef FG_POLICY
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "policy launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

int fg_policy_bfs_state(int B, int N, int per_layer, const float* pos_x, const float* pos_y,
                        const float* ideal_shape, const float* ideal_vel, float* act, void* stream) {
    const DeviceGuard device_guard(stream, pos_x);
    if (B == 0) return FG_OK;
    if (B < 0) return fail(FG_ERR_BAD_ARG, "B must be >= 0%s");
    if (N < 3 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "formation_hd_env needs 3 <= N <= 1024%s");
    FgPolicyLevels pl;
    if (!policy_levels_for(N, per_layer, &pl))
        return fail(FG_ERR_UNSUPPORTED_N, "fg_policy_bfs_state: N must be per_layer^L with 2 <= per_layer <= 8%s");
    if (!pos_x || !pos_y || !ideal_shape || !ideal_vel || !act)
        return fail(FG_ERR_BAD_ARG, "fg_policy_bfs_state: a required pointer is NULL%s");
    if (((uintptr_t)ideal_shape & 7u) || ((uintptr_t)ideal_vel & 7u) || ((uintptr_t)act & 7u))
        return fail(FG_ERR_ALIGNMENT, "ideal_shape, ideal_vel and act must be 8-byte aligned%s");
    return launch_policy_state(B, N, pl, pos_x, pos_y, ideal_shape, ideal_vel, act, (hipStream_t)stream);
}

int fg_rollout_hd_policy(const FgParams* params, int B, int N, int K, int per_layer,
                         float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                         float* act_seq, float* ideal_shape, float* ideal_vel, int32_t* step,
                         float* obs_seq, float* reward_seq, float* indiv_seq, uint8_t* done_seq,
                         int obs_every, void* stream) {
    const DeviceGuard device_guard(stream, pos_x);
    int rc = check_params(params);
    if (rc) return rc;
    if (B == 0 || K == 0) return FG_OK;
    if (B < 0 || K < 0) return fail(FG_ERR_BAD_ARG, "B and K must be >= 0%s");
    if (N < 3 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "formation_hd_env needs 3 <= N <= 1024%s");
    FgPolicyLevels pl;
    if (!policy_levels_for(N, per_layer, &pl))
        return fail(FG_ERR_UNSUPPORTED_N, "fg_rollout_hd_policy: N must be per_layer^L with 2 <= per_layer <= 8%s");
    if (!pos_x || !pos_y || !vel_x || !vel_y || !act_seq || !ideal_shape || !ideal_vel || !step || !reward_seq)
        return fail(FG_ERR_BAD_ARG, "fg_rollout_hd_policy: a required pointer is NULL%s");
    if (((uintptr_t)obs_seq & 15u) || ((uintptr_t)act_seq & 7u) || ((uintptr_t)ideal_shape & 7u) || ((uintptr_t)ideal_vel & 7u))
        return fail(FG_ERR_ALIGNMENT, "obs_seq must be 16-byte, act_seq/ideal_shape/ideal_vel 8-byte aligned%s");
    Args a; memset(&a, 0, sizeof(a));
    a.p = *params; a.B = B; a.N = N; a.inv_n = 1.0f / (float)N; a.K = K; a.obs_every = obs_every < 1 ? 1 : obs_every;
    if ((rc = set_obs_pitch(&a)) != FG_OK) return rc;
    a.do_phys = 1; a.do_post = 1;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y;
    a.shape = ideal_shape; a.ivel = ideal_vel; a.step = step;
    a.obs = obs_seq; a.rew = reward_seq; a.indiv = indiv_seq; a.done = done_seq;
    a.pl = pl; a.act_out = act_seq;
    return rollout_hd_policy_dispatch(a, per_layer, (hipStream_t)stream);
}

int fg_policy_bfs(int B, int N, int per_layer, const float* obs, int64_t obs_env_stride, float* act, void* stream) {
    const DeviceGuard device_guard(stream, act);    // the actions (obs may live in an fg_arena, whose pointers carry no device attribute on every runtime)
    if (B == 0) return FG_OK;
    if (B < 0) return fail(FG_ERR_BAD_ARG, "B must be >= 0%s");
    if (N < 3 || N > FG_MAX_AGENTS) return fail(FG_ERR_UNSUPPORTED_N, "formation_hd_env needs 3 <= N <= 1024%s");
    FgPolicyLevels pl;
    if (!policy_levels_for(N, per_layer, &pl))
        return fail(FG_ERR_UNSUPPORTED_N, "fg_policy_bfs: N must be per_layer^L with 2 <= per_layer <= 8%s");
    if (!obs || !act) return fail(FG_ERR_BAD_ARG, "fg_policy_bfs: a required pointer is NULL%s");
    const int64_t stride = obs_env_stride ? obs_env_stride : 6LL * N * N;
    if (stride < 6LL * N || (stride & 1)) return fail(FG_ERR_BAD_ARG, "fg_policy_bfs: obs_env_stride must be even and >= 6 N%s");
    if (((uintptr_t)obs & 7u) || ((uintptr_t)act & 7u)) return fail(FG_ERR_ALIGNMENT, "obs and act must be 8-byte aligned%s");
    const int lpe = N <= 16 ? 16 : N <= 32 ? 32 : N <= 64 ? 64 : 256;      // lanes per env
    const int E = 256 / lpe;
    const int grid = (B + E - 1) / E;
    const int lds = E * policy_block_units(N) * (int)sizeof(float2);        // <= 56 KiB
    hipStream_t st = (hipStream_t)stream;
#define FG_POLICY(PER) case PER: hipLaunchKernelGGL((policy_bfs_kernel<PER>), dim3(grid), dim3(256), lds, st, B, N, lpe, pl, \
                                                   obs, (long long)stride, act); break;
    switch (per_layer) { FG_POLICY(2) FG_POLICY(3) FG_POLICY(4) FG_POLICY(5) FG_POLICY(6) FG_POLICY(7) FG_POLICY(8) }
#undef FG_POLICY
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(FG_ERR_HIP, "policy launch failed: %s", hipGetErrorString(err));
    return FG_OK;
}

int fg_decode_actions(int mode, int64_t count, void* action, float* u_out, void* stream) {
    const DeviceGuard device_guard(stream, u_out);
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

int fg_describe_launch(const FgParams* params, const FgScenario* scenario, int B, int N, int K, int per_layer, int obs_every,
                       int index_outputs, char* out, int out_len) {
    if (!out || out_len < 2) return fail(FG_ERR_BAD_ARG, "fg_describe_launch: out buffer required%s");
    int rc = check_params(params);
    if (rc) return rc;
    if (B <= 0 || K < 0) return fail(FG_ERR_BAD_ARG, "fg_describe_launch: B > 0 and K >= 0 required%s");
    out[0] = 0;
    // stand-ins for the caller's buffers: only their NULL-ness and alignment steer the dispatch, nothing is dereferenced
    float* const f = reinterpret_cast<float*>((uintptr_t)4096);
    int32_t* const i32 = reinterpret_cast<int32_t*>((uintptr_t)4096);
    uint8_t* const u8 = reinterpret_cast<uint8_t*>((uintptr_t)4096);
    g_describe = out; g_describe_cap = out_len;
    if (scenario) {
        rc = launch_scenario(params, scenario, B, N, 1, f, f, f, f, f, f, scenario->num_obstacles ? f : nullptr,
                             scenario->num_obstacles ? f : nullptr, i32, f, f, f, u8, nullptr, nullptr, K < 1 ? 1 : K,
                             obs_every < 1 ? 1 : obs_every);
    } else if (N < 3 || N > FG_MAX_AGENTS) {
        rc = fail(FG_ERR_UNSUPPORTED_N, "formation_hd_env needs 3 <= N <= 1024%s");
    } else {
        Args a; memset(&a, 0, sizeof(a));
        a.p = *params; a.B = B; a.N = N; a.inv_n = 1.0f / (float)N; a.K = K < 1 ? 1 : K; a.obs_every = obs_every < 1 ? 1 : obs_every;
        a.do_phys = 1; a.do_post = 1;
        a.px = f; a.py = f; a.vx = f; a.vy = f; a.act = f; a.shape = f; a.ivel = f; a.step = i32;
        a.obs = f; a.rew = f; a.indiv = f; a.done = u8;
        if (index_outputs) { a.near_lm = i32; a.near_ag = i32; a.hd_idx = i32; }
        rc = set_obs_pitch(&a);
        if (rc == FG_OK) {
            if (per_layer > 0) {
                if (!policy_levels_for(N, per_layer, &a.pl)) rc = fail(FG_ERR_UNSUPPORTED_N, "N must be per_layer^L with 2 <= per_layer <= 8%s");
                else { a.act_out = f; rc = rollout_hd_policy_dispatch(a, per_layer, nullptr); }
            } else if (K == 0) {
                rc = step_hd_dispatch(a, nullptr);                      // fg_step_hd
            } else {
                rc = rollout_hd_dispatch(a, nullptr);                   // fg_rollout_hd with K steps
            }
        }
    }
    g_describe = nullptr; g_describe_cap = 0;
    return rc;
}

}  // extern "C"
