// formation_hip_f64.hip - "parity mode" build of the fused step kernel: the SAME kernel source as
// libformation_hip (fg_step_kernel.hpp, fg_pair_loops.hpp, the reductions of fg_common.hpp) compiled with
// real = double, run-time agent count, flat observation writer - and of the pipelined rollout kernel
// (fg_rollout_kernels.hpp: rollout_kernel<9 | 27, ...> and rollout_kernel_wide<81 | 243, ...> with the rows writer).  TEST INFRASTRUCTURE ONLY (tests/test_gpu_f64_parity.py):
// it lets the kernel's algorithm free-run against the reference's float64 trajectories
// (/root/reference/formation_gym/core.py:206-225, :289-322; environment.py:113-142) over whole fixtures, which an
// fp32 run cannot do beyond ~10 steps because stiff contacts amplify rounding chaotically.  Not shipped in the
// product path: formation_gym never loads it.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -DFG_F64=1 -fPIC -shared -Iinclude
//         -o lib/libformation_hip_f64.so csrc/formation_hip_f64.hip          (no -fapprox-func: full-precision libm)
#define FG_F64 1
#include "fg_common.hpp"
#include "fg_pair_loops.hpp"
#include "fg_step_kernel.hpp"
#include "fg_rollout_kernels.hpp"

namespace fg {

// The pipelined producer / writer rollout kernel (fg_rollout_kernels.hpp) with real = double: the SAME source as the product's
// rollout_kernel<N, G, TP, TW, E, 0, 0, false> - producer waves, double-buffered LDS tables, the rows writer - so that the K-step
// path has its own whole-fixture free-running check against the reference's fp64 trajectories.
template <int NC, int G, int TP, int TW, int E>
static hipError_t launch_roll64(const Args& a, hipStream_t st) {
    const int grid = (a.B + E - 1) / E;
    constexpr int lds = roll_lds_bytes<NC, TW, E, 0, 0>();
    static_assert(lds <= 64 * 1024, "the parity build stays inside the default LDS limit");
    hipLaunchKernelGGL((rollout_kernel<NC, G, TP, TW, E, 0, 0, false>), dim3(grid), dim3(TP + TW), lds, st, a);
    return hipGetLastError();
}

// ... and the one-env-per-producer-wave form for 81 / 243 agents (rollout_kernel_wide)
template <int NC, int A, int E, int TW>
static hipError_t launch_wide64(Args a, hipStream_t st) {
    a.groups = 1;
    const int grid = (a.B + E - 1) / E;
    constexpr int lds = E * roll_block_floats(NC) * (int)sizeof(real);
    static_assert(lds <= 64 * 1024, "the parity build stays inside the default LDS limit");
    hipLaunchKernelGGL((rollout_kernel_wide<NC, A, E, TW, 0, false>), dim3(grid), dim3(E * 64 + TW), lds, st, a);
    return hipGetLastError();
}

template <int G, int T, int E, bool IDX>
static hipError_t launch64(const Args& a, hipStream_t st) {
    const int grid = (a.B + E - 1) / E;
    const int lds = (E * env_block_floats(a.N) + 72) * (int)sizeof(real);
    hipLaunchKernelGGL((step_kernel<0, G, T, E, IDX, false>), dim3(grid), dim3(T), lds, st,
                       a.B, a.N, (const real*)a.px, (const real*)a.py, (const real*)a.vx, (const real*)a.vy,
                       (const real*)a.shape, (const real*)a.ivel, (const int32_t*)a.step, a);
    return hipGetLastError();
}

}  // namespace fg

extern "C" {

// Constants in double, field for field what FgParams holds in float (include/formation_hip.h)
struct Fg64Params {
    double dt, damping, contact_force, contact_margin, sensitivity, mass, dist_min, collide_thresh;
    int32_t world_length, reserved;
};

// fg_step_hd in fp64: state, actions, formation and every output are double arrays of the fp32 entry point's shapes.
// Returns 0 or a negative code (-1 bad argument, -2 unsupported N, -4 HIP error).
int fg64_step_hd(const Fg64Params* params, int B, int N,
                 double* pos_x, double* pos_y, double* vel_x, double* vel_y,
                 const double* act, double* ideal_shape, double* ideal_vel, int32_t* step,
                 double* obs, double* reward, double* indiv_reward, uint8_t* done,
                 int32_t* near_lm, int32_t* near_ag, int32_t* hd_idx, void* stream) {
    using namespace fg;
    if (!params || B < 0 || !pos_x || !pos_y || !vel_x || !vel_y || !act || !ideal_shape || !ideal_vel || !step || !obs || !reward)
        return -1;
    if (N < 3 || N > 1024) return -2;
    if (B == 0) return 0;
    Args a; memset(&a, 0, sizeof(a));
    a.p.dt = params->dt; a.p.damping = params->damping; a.p.contact_force = params->contact_force;
    a.p.contact_margin = params->contact_margin; a.p.sensitivity = params->sensitivity; a.p.mass = params->mass;
    a.p.dist_min = params->dist_min; a.p.collide_thresh = params->collide_thresh; a.p.world_length = params->world_length;
    a.B = B; a.N = N; a.inv_n = 1.0 / (double)N; a.K = 1; a.obs_every = 1; a.do_phys = 1; a.do_post = 1;
    a.obs_pitch = 3LL * N * N;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y; a.act = act;
    a.shape = ideal_shape; a.ivel = ideal_vel; a.step = step;
    a.obs = obs; a.rew = reward; a.indiv = indiv_reward; a.done = done;
    a.near_lm = near_lm; a.near_ag = near_ag; a.hd_idx = hd_idx;
    hipStream_t st = (hipStream_t)stream;
    const bool idx = near_lm || near_ag || hd_idx;
    int G = 4; while (G < N) G <<= 1;
    if (G > 64 && G < 128) G = 128;
    hipError_t err;
#define FG64(GG, TT, EE) err = idx ? launch64<GG, TT, EE, true>(a, st) : launch64<GG, TT, EE, false>(a, st)
    switch (G) {
        case 4: FG64(4, 64, 16); break;      case 8: FG64(8, 64, 8); break;       case 16: FG64(16, 64, 4); break;
        case 32: FG64(32, 128, 4); break;    case 64: FG64(64, 128, 2); break;    case 128: FG64(128, 128, 1); break;
        case 256: FG64(256, 256, 1); break;  case 512: FG64(512, 512, 1); break;  default: FG64(1024, 1024, 1); break;
    }
#undef FG64
    return err == hipSuccess ? 0 : -4;
}

// fg_rollout_hd in fp64 for 9, 27, 81 and 243 agents (the reference's own hierarchy sizes): K steps in ONE launch of the pipelined kernel; act [K][B][N][2], obs [K][B][N][6N],
// reward / indiv_reward / done [K][B][N].  No auto-reset (the fixtures' episodes do not end inside a launch).
int fg64_rollout_hd(const Fg64Params* params, int B, int N, int K,
                    double* pos_x, double* pos_y, double* vel_x, double* vel_y,
                    const double* act, double* ideal_shape, double* ideal_vel, int32_t* step,
                    double* obs, double* reward, double* indiv_reward, uint8_t* done, void* stream) {
    using namespace fg;
    if (!params || B < 0 || K < 1 || !pos_x || !pos_y || !vel_x || !vel_y || !act || !ideal_shape || !ideal_vel || !step || !obs || !reward)
        return -1;
    if (N != 9 && N != 27 && N != 81 && N != 243) return -2;
    if (B == 0) return 0;
    Args a; memset(&a, 0, sizeof(a));
    a.p.dt = params->dt; a.p.damping = params->damping; a.p.contact_force = params->contact_force;
    a.p.contact_margin = params->contact_margin; a.p.sensitivity = params->sensitivity; a.p.mass = params->mass;
    a.p.dist_min = params->dist_min; a.p.collide_thresh = params->collide_thresh; a.p.world_length = params->world_length;
    a.B = B; a.N = N; a.inv_n = 1.0 / (double)N; a.K = K; a.obs_every = 1; a.do_phys = 1; a.do_post = 1;
    a.obs_pitch = 3LL * N * N;
    a.px = pos_x; a.py = pos_y; a.vx = vel_x; a.vy = vel_y; a.act = act;
    a.shape = ideal_shape; a.ivel = ideal_vel; a.step = step;
    a.obs = obs; a.rew = reward; a.indiv = indiv_reward; a.done = done;
    hipStream_t st = (hipStream_t)stream;
    const hipError_t err = N == 9 ? launch_roll64<9, 16, 64, 128, 4>(a, st) : N == 27 ? launch_roll64<27, 32, 128, 256, 4>(a, st)
                         : N == 81 ? launch_wide64<81, 2, 2, 128>(a, st) : launch_wide64<243, 4, 1, 256>(a, st);
    return err == hipSuccess ? 0 : -4;
}

}  // extern "C"
