/*
 * formation_hip.h - C ABI of libformation_hip.so, the MI355X (gfx950) native
 * implementation of the formation_gym hot path.
 *
 * The reference (jc-bao/gym-formation) is pure Python and has no FFI; its
 * boundary for this path is the plugin API
 *     formation_gym.make_env(...)                 formation_gym/__init__.py:6-17
 *     MultiAgentEnv.reset() / .step(action_n)     formation_gym/environment.py:113-156
 *     Scenario.{reset_world,observation,reward}   formation_gym/envs/formation_hd_env.py:38-95
 *     World.step()                                formation_gym/core.py:206-225
 * Each entry point below names the reference interface it replaces.  The host
 * side that binds them through ctypes is gym-formation_amd/formation_gym/_native.py;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller
 *     (e.g. torch.Tensor.data_ptr()); the library allocates nothing persistent
 *     and keeps no pointer after a call returns.
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as
 *     void*, NULL = the default stream); no device synchronisation inside, so
 *     every entry point may be captured into a hipGraph.
 *   - return value: FG_OK (0) or a negative FgStatus; nothing throws or exits.
 *     An empty batch (B = 0), zero steps (K = 0) or count = 0 is a successful no-op:
 *     no launch, buffers may be NULL.
 *     fg_last_error() returns a thread-local description of the last failure.
 *   - layouts are env-major and contiguous.  B = number of independent
 *     environments (the data-parallel unit), N = agents per environment.
 *       pos_x,pos_y,vel_x,vel_y  float [B][N]     structure-of-arrays agent state
 *       act                      float [B][N][2]  raw actions (scaled by `sensitivity` inside)
 *       ideal_shape              float [B][N][2]  Scenario.ideal_shape (already centred)
 *       ideal_vel                float [B][2]     Scenario.ideal_vel
 *       step                     int32 [B]        MultiAgentEnv.current_step per env
 *       obs                      float [B][N][6N] (16-byte aligned base)
 *       reward                   float [B][N]     shared reward, broadcast to every agent
 *       indiv_reward             float [B][N]     info_n[i]['individual_reward']   (may be NULL)
 *       done                     uint8 [B][N]     current_step >= world_length
 *       near_lm, near_ag         int32 [B][N]     landmark-index assignments        (may be NULL)
 *       hd_idx                   int32 [B][4]     Hausdorff witness indices         (may be NULL)
 */
#ifndef FORMATION_HIP_H_
#define FORMATION_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FG_ABI_VERSION 8
#define FG_MAX_AGENTS 1024
#define FG_MAX_WALLS 4

typedef enum FgStatus {
    FG_OK = 0,
    FG_ERR_BAD_ARG = -1,        /* NULL pointer, B < 0, K < 0, bad params              */
    FG_ERR_UNSUPPORTED_N = -2,  /* N < 2 or N > FG_MAX_AGENTS                          */
    FG_ERR_ALIGNMENT = -3,      /* obs base not 16-byte aligned                        */
    FG_ERR_HIP = -4             /* a HIP runtime call failed (see fg_last_error)       */
} FgStatus;

/* Physics / scenario constants.  Field -> reference source:
 *   dt              core.py:125            damping         core.py:127
 *   contact_force   core.py:129            contact_margin  core.py:130
 *   sensitivity     environment.py:218-221 mass            core.py:69,73-75
 *   dist_min        core.py:307 (size_a + size_b, force contact distance)
 *   collide_thresh  formation_hd_env.py:121 ((size_a+size_b)/2) or
 *                   basic_formation_env.py:91 (size_a+size_b)
 *   world_length    formation_hd_env.py:13,16 / core.py:113
 *   auto_reset      0: like the reference env (caller resets);
 *                   1: vec-env worker semantics, train/maddpg-v2/utils/env_wrappers.py:14-18
 *                      (when done: re-initialise the env on device and return the RESET
 *                      observation together with the pre-reset reward/done)
 *   seed, rng_offset  counter-RNG key / per-call offset for the device-side reset
 *                   (distributional parity with formation_hd_env.py:77-95 only)
 * World options that no reference scenario switches on (0 = None, the reference default):
 *   accel           core.py:236 + environment.py:219-220: force = mass*accel*(accel*action)
 *   max_speed       core.py:271-276 speed clamp after the velocity update
 *   u_noise         core.py:232-233 Gaussian motor noise (device counter RNG, distributional parity)
 *   walls           core.py:27-41,255-261,325-362 get_wall_collision_force; hard or soft (FgWall.soft)
 * Layout option of the observation output (fg_step_hd, fg_observe_hd, fg_rollout_hd, fg_rollout_hd_policy):
 *   obs_env_pitch   the [N][6N] block of env b starts obs_env_pitch floats after env b-1's (even, >= 6 N^2);
 *                   every agent count of the reference is odd, so contiguous env blocks are only 8-byte aligned -
 *                   a pitch rounded up to 32 floats puts every env on its own 128-byte lines (a strided
 *                   [B][N][6N] view on the caller's side).  A rollout's step slots are B * pitch apart.  The pad
 *                   floats between env blocks belong to the library: they may be overwritten with zeros.      */
typedef struct FgWall {
    int32_t vertical;    /* orient: 0 = 'H' (lies on y = axis_pos), 1 = 'V' */
    float axis_pos;
    float end0, end1;    /* endpoints along the wall */
    float width;
    int32_t soft;        /* core.py:36-37 Wall.hard: 0 = hard (every entity feels it), 1 = soft (ghost entities pass through, :326-327) */
} FgWall;

typedef struct FgParams {
    float dt;
    float damping;
    float contact_force;
    float contact_margin;
    float sensitivity;
    float mass;
    float dist_min;
    float collide_thresh;
    int32_t world_length;
    int32_t auto_reset;
    uint64_t seed;
    uint64_t rng_offset;
    float accel;
    float max_speed;
    float u_noise;
    int32_t num_walls;
    FgWall walls[FG_MAX_WALLS];
    int32_t obs_env_pitch;   /* floats between the observation blocks of consecutive envs; 0 = 6 N^2 (contiguous) */
    int32_t env_index_base;  /* GLOBAL index of this batch's env 0 (a rank's slice of a sharded batch): the device
                                counter RNG (auto-reset, fg_reset_hd, motor noise) is keyed by seed and GLOBAL env index,
                                so its draws do not depend on how the batch is cut over GPUs */
    const uint64_t* rng_offset_dev;  /* optional DEVICE counter added to rng_offset when the launch runs (NULL = none): a
                                caller that replays a captured hipGraph keeps its per-step offset here and advances it
                                with a device-side add between launches, since by-value arguments are frozen in a graph.
                                The library only reads it. */
    const float* agent_props;  /* optional DEVICE table float [N][FG_AGENT_PROPS], one row per agent (NULL = every agent
                                takes the scalars above - all the reference's scenarios):
                                  [0] mass       Entity.initial_mass, core.py:68-75: contact forces are scaled by
                                                 force_ratio = m_b / m_a (core.py:314-317), velocity gains F / m (:270)
                                  [1] size       Entity.size: contact distance of a pair = size_a + size_b (core.py:307),
                                                 collision penalty distance = collide_thresh / dist_min * (size_a + size_b)
                                                 (formation_hd_env.py:119-121), wall contact (core.py:340-344)
                                  [2] accel      Entity.accel (0 = None): core.py:236, environment.py:219-220
                                  [3] max_speed  Entity.max_speed (0 = None): core.py:271-276
                                  [4] u_noise    Agent.u_noise (0 = None): core.py:232-233
                                  [5] c_noise    Agent.c_noise for fg_update_comm (0 = None; < 0 = a silent agent)
                                  [6] flags      an integer stored as a float, 0 for an ordinary agent:
                                                 1 = not movable (core.py:231, 266-267: no action force, state never
                                                     integrated; a partner's contact force is then not scaled by the mass
                                                     ratio, :319-321),
                                                 2 = does not collide (:292-293: no contact force with anybody; the reward's
                                                     collision penalties of such an agent are not counted,
                                                     formation_hd_env.py:71),
                                                 4 = ghost (passes through soft walls, :326-327)
                                  [7] reserved (0)
                                With a table, `sensitivity` is the value for agents whose accel is None (5.0) and
                                mass / dist_min / accel / max_speed / u_noise above are not read.  Honoured by
                                fg_step_hd, fg_physics_step, fg_observe_hd, fg_rollout_hd, fg_rollout_hd_policy, and - columns
                                [0] ... [4] and [6], the agents only: obstacles keep FgScenario.obstacle_size, unit mass and
                                are ordinary movable colliders - by fg_step_scenario, fg_step_basic, fg_rollout_scenario
                                (penalty distance of a pair = collide_thresh / dist_min * (size_a + size_b)). */
    const float* comm_state;   /* optional DEVICE float [B][N][2] = AgentState.c of every agent (World.dim_c = 2): copied
                                into the communication block of the observation, row i = c_j for j != i in index order
                                (formation_hd_env.py:48-51,59); NULL = zeros, the silent agents of every reference
                                scenario (core.py:281-282).  Honoured by fg_step_hd, fg_observe_hd, fg_rollout_hd. */
    int32_t obs_placed;        /* hint: 1 = the observation buffer of this launch was composed with fg_arena_* of chunks spread
                                over the device's memory.  Such a buffer takes the store stream of MORE writer
                                waves: the 27-agent rollout then runs 8 paced writer waves per workgroup (11.7 us/step at
                                27 x 4096) where 4 are best on an ordinary allocation (12.9 there, 13.2-14.4 with 8).  0 = unknown */
    int32_t reserved0;
} FgParams;
#define FG_AGENT_PROPS 8
#define FG_AGENT_IMMOVABLE 1
#define FG_AGENT_NO_COLLIDE 2
#define FG_AGENT_GHOST 4
#define FG_AGENT_SCRIPTED 8   /* the action of this agent is a scripted agent's `action.u` (Agent.action_callback, core.py:210-211):
                                 taken as it is - environment.py:216-221's sensitivity scaling applies to policy agents only */

/* Landmark scenarios with few agents (fg_step_scenario).  Field -> reference source:
 *   kind            which Scenario file under formation_gym/envs/
 *   num_landmarks   L: make_world default of that file (make_env passes only num_agents)
 *   num_obstacles   M: formation_hd_obs_env.py:14 (movable colliding landmarks)
 *   num_obs         formation_hd_partial_env.py:15,44 ring neighbours observed
 *   obs_range       formation_hd_partial_range_env.py:15,46 clip of relative positions
 *   obstacle_size   formation_hd_obs_env.py:39;  obstacle_v{x,y}, obstacle_floor  :84-89
 *   penalty         reward per collision: 1 (basic/partial/range), 2 (obstacle, :92-98)
 * At the reference's own shapes - basic (N, L) = (3, 3), partial (5, 5, num_obs 3), range (4, 4), obstacle (4, 4, M 3), and
 * what make_env's default num_agents = 3 makes of the last three - a launch runs the one-env-per-lane kernel
 * (csrc/fg_scn_lane_kernel.hpp: every count a compile-time constant), else the run-time-count kernel; same results.   */
typedef enum FgScenarioKind {
    FG_SCN_BASIC = 1, FG_SCN_PARTIAL = 2, FG_SCN_RANGE = 3, FG_SCN_OBSTACLE = 4
} FgScenarioKind;

typedef struct FgScenario {
    int32_t kind;
    int32_t num_landmarks;
    int32_t num_obstacles;
    int32_t num_obs;
    float obs_range;
    float obstacle_size;
    float obstacle_vx;
    float obstacle_vy;
    float obstacle_floor;
    float penalty;
    int32_t variant;         /* 0 = the library's choice of kernel; 1 = the run-time-count kernel even where a
                                one-env-per-lane instantiation exists (tests / A-B runs: the two agree bit for bit) */
    int32_t reserved;
} FgScenario;

/* Placed device memory ------------------------------------------------------
 * The rate at which the rollout kernels stream observations depends on WHICH physical memory the buffer is composed of: a
 * multi-GB buffer on physically neighbouring memory runs the same launch at 5.0-5.4 TB/s, a well composed one at 6.2-6.8
 * TB/s (a physically contiguous allocation: 2-2.6 TB/s; profiles/r03_place/).  Which composition is fast cannot be told in
 * advance (the driver decides where a chunk lies; compositions made by rule do not hold up: profiles/r04_place/), so a caller
 * TIMES its own launch on a few compositions and keeps the best - from an arena that need not be larger than 1.5-6 x the
 * buffer (profiles/r04_place/arena_size.txt).  An arena is a set of separately created physical chunks (HIP virtual memory
 * management) from which a caller composes buffers:
 *   fg_arena_create   creates ceil(bytes / chunk) chunks of device memory on `device` (chunk_bytes is rounded up to the
 *                     allocation granularity; 0 = 1 GiB) and gives them their place in memory in INDEX order (a chunk is
 *                     placed when it is first mapped, so every chunk is mapped once and unmapped again: chunks far apart
 *                     in index lie far apart in memory); on return nothing is mapped.  *chunk_out = chunk size,
 *                     *chunks_out = count
 *   fg_arena_map      maps `count` chunks (indices, any order) at fresh contiguous addresses *base, read-write for the
 *                     device: a buffer made of exactly those chunks.  A chunk is mapped at ONE address at a time
 *   fg_arena_unmap    removes one such mapping (its chunks become available again; their contents stay)
 *   fg_arena_trim     hands every chunk that is not mapped right now back to the driver
 *   fg_arena_destroy  unmaps and releases everything
 * unmap / destroy drain the device first (no launch may still be using the addresses); map returns when the mapping is
 * usable.  An address range that has held a mapping is never used again in this process: hipMemUnmap leaves the GPU's
 * translations of it behind on this stack (profiles/r03_place/va_reuse_check.txt), so the library retires the reservation
 * instead of freeing it - that costs address space only (fg_arena_retired_address_bytes: the running total; a placement of
 * a 1.4 GB buffer retires ~250 GB of the 128 TiB; once 64 TiB are retired fg_arena_create / _map refuse with FG_ERR_HIP and the
 * caller uses ordinary allocations), the physical memory goes back at trim / destroy.
 * These are the only entry points that allocate; they enqueue nothing and take no stream.  Calls on ONE arena must not
 * run concurrently (the arena is the caller's object; different arenas are independent). */
int fg_arena_create(int device, uint64_t bytes, uint64_t chunk_bytes, void** arena, uint64_t* chunk_out, uint32_t* chunks_out);
int fg_arena_map(void* arena, const uint32_t* chunk_index, uint32_t count, void** base);
int fg_arena_unmap(void* arena, void* base);
/* shrinks the mapping at `base` to chunks [first, first + count) of it (the others are unmapped, their addresses retired); the
 * window keeps its address, returned in *new_base = base + first * chunk */
int fg_arena_keep_window(void* arena, void* base, uint32_t first, uint32_t count, void** new_base);
int fg_arena_trim(void* arena);
uint64_t fg_arena_retired_address_bytes(void);
int fg_arena_destroy(void* arena);

/* library / diagnostics --------------------------------------------------- */
int fg_abi_version(void);
const char* fg_last_error(void);
/* The device an entry point would launch on for this stream and this first state pointer: the stream's device when the
 * stream is not NULL, else the device the pointer's memory lives on (hipPointerGetAttributes), else -1 = "cannot tell: the
 * calling thread's current device".  Every entry point applies this rule (and switches device for the duration of the call
 * when the process sees more than one GPU), so that an env living on cuda:1 can be driven from a thread whose current device
 * is cuda:0; this call only reports it.  Memory composed with fg_arena_map counts as the device's like any other allocation
 * (tests/test_gpu_multidevice.py::test_launch_device_rule_on_one_gpu). */
int fg_launch_device(void* stream, const void* data);
/* Launch geometry the library will use for N agents: threads per workgroup,
 * environments per workgroup, dynamic LDS bytes.  Returns FgStatus. */
int fg_kernel_config(int N, int* threads, int* envs_per_wg, int* lds_bytes);
/* Dry run of the dispatch: which kernel instantiation(s) and launch geometry fg_step_hd (K = 0), fg_rollout_hd (K >= 1,
 * per_layer = 0), fg_rollout_hd_policy (per_layer > 0) or - with a scenario descriptor - fg_rollout_scenario would use for this
 * shape and these params (obs_env_pitch, obs_placed, World options and all), written to `out` as text.  Touches no device:
 * callable without a GPU.  tests/test_dispatch_snapshot.py holds the committed choices over a grid of shapes. */
int fg_describe_launch(const FgParams* params, const FgScenario* scenario, int B, int N, int K, int per_layer, int obs_every,
                       int index_outputs, char* out, int out_len);
/* Algorithmic bytes per env-step of fg_step_hd (24 N^2 + 53 N + 16). */
int64_t fg_step_hd_bytes(int N);

/* MultiAgentEnv.step for formation_hd_env, all B envs, ONE fused launch (more than 64 agents in at most 128 envs: two
 * launches - everything but the observation, then the observation streamed by several workgroups per env; same results):
 * _set_action (environment.py:187-236) -> World.step (core.py:206-225:
 * apply_action_force :228-237, apply_environment_force :240-262 with
 * get_entity_collision_force :289-322, integrate_state :264-277,
 * update_agent_state :279-286) -> Scenario.observation / reward / done for every
 * agent (formation_hd_env.py:38-75, environment.py:126-138,172-178).
 * State (pos/vel/step, and ideal_shape/ideal_vel when auto_reset) is updated in place. */
int fg_step_hd(const FgParams* params, int B, int N,
               float* pos_x, float* pos_y, float* vel_x, float* vel_y,
               const float* act, float* ideal_shape, float* ideal_vel, int32_t* step,
               float* obs, float* reward, float* indiv_reward, uint8_t* done,
               int32_t* near_lm, int32_t* near_ag, int32_t* hd_idx, void* stream);

/* The same step as a PLAN: fg_step_hd_plan checks the arguments once and keeps the launch description (the pointers are the
 * caller's and must stay valid while the plan lives - like fg_arena_*, a declared exception to "keeps no pointer");
 * fg_plan_launch(plan, rng_offset) enqueues one step on the plan's stream with FgParams.rng_offset = rng_offset and nothing
 * else changed; fg_plan_destroy frees the description.  For step loops that re-use their buffers (MultiAgentEnv.step called
 * per step, environment.py:113-142, test.py:17-28): the per-step host work is a two-argument call plus the kernel launch.
 * Results equal fg_step_hd's bit for bit (the same dispatch). */
int fg_step_hd_plan(const FgParams* params, int B, int N,
                    float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                    const float* act, float* ideal_shape, float* ideal_vel, int32_t* step,
                    float* obs, float* reward, float* indiv_reward, uint8_t* done,
                    int32_t* near_lm, int32_t* near_ag, int32_t* hd_idx, void* stream, void** plan);
int fg_plan_launch(void* plan, uint64_t rng_offset);
int fg_plan_destroy(void* plan);

/* World.step only (core.py:206-225) incl. the action scaling of
 * environment.py:216-221; for per-stage parity tests.  Updates pos/vel in place. */
int fg_physics_step(const FgParams* params, int B, int N,
                    float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                    const float* act, void* stream);

/* Scenario.observation + Scenario.reward + _get_done on the CURRENT state, no
 * physics and no step increment (formation_hd_env.py:38-75; what env.reset()
 * returns, environment.py:154-155).  reward/indiv_reward/done may be NULL. */
int fg_observe_hd(const FgParams* params, int B, int N,
                  const float* pos_x, const float* pos_y, const float* vel_x, const float* vel_y,
                  const float* ideal_shape, const float* ideal_vel, const int32_t* step,
                  float* obs, float* reward, float* indiv_reward, uint8_t* done,
                  int32_t* near_lm, int32_t* near_ag, int32_t* hd_idx, void* stream);

/* K consecutive env.step calls in ONE launch (the caller's rollout loop,
 * test.py:17-28 / train/maddpg-v2/main.py:77-91) with pre-staged actions.
 *   act_seq [K][B][N][2]; obs_seq [K][B][N][6N]; reward_seq, indiv_seq [K][B][N];
 *   done_seq [K][B][N].  If obs_every > 1 only steps k with (k+1) % obs_every == 0
 *   write an observation (into slot k / obs_every); rewards/dones are always written. */
int fg_rollout_hd(const FgParams* params, int B, int N, int K,
                  float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                  const float* act_seq, float* ideal_shape, float* ideal_vel, int32_t* step,
                  float* obs_seq, float* reward_seq, float* indiv_seq, uint8_t* done_seq,
                  int obs_every, void* stream);

/* Scenario.reset_world on device for the envs whose mask byte is non-zero
 * (mask NULL = all), counter-based RNG (formation_hd_env.py:77-95 draw
 * distribution: pos ~ U(-1,1)^2, vel = 0, ideal_shape = centred U(-1,1)^2,
 * ideal_vel ~ U(-1,1)^2, step = 0). */
int fg_reset_hd(const FgParams* params, int B, int N, const uint8_t* mask,
                float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                float* ideal_shape, float* ideal_vel, int32_t* step, void* stream);

/* The same reset drawn from each env's own legacy NumPy MT19937 stream (environment.py:106-110
 * np.random.seed; formation_hd_env.py:77-95 draw order), bit-exact with the reference's host RNG:
 *   mt_state uint32 [B][626] = RandomState.get_state() key[624], pos, pad; advanced in place.
 *   landmark_pos float [B][N][2] (may be NULL) receives the un-centred landmark positions. */
int fg_reset_hd_mt(int B, int N, const uint8_t* mask, uint32_t* mt_state,
                   float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                   float* ideal_shape, float* ideal_vel, float* landmark_pos, int32_t* step, void* stream);

/* The vec-env worker's reset (train/maddpg-v2/utils/env_wrappers.py:14-18: `if all(done): ob = env.reset()`) decided ON
 * THE DEVICE: every env whose episode is over (step[b] >= world_length) is re-initialised from its own MT19937 stream as
 * fg_reset_hd_mt does, and - if obs is not NULL - its block of the observation tensor [B][N][6N] (env blocks
 * obs_env_pitch floats apart, 0 = contiguous) is overwritten with the RESET observation (formation_hd_env.py:52-59 on the
 * fresh state, the bits fg_observe_hd gives).  No mask upload, no host read-back, one launch. */
int fg_reset_hd_mt_done(int B, int N, int world_length, uint32_t* mt_state,
                        float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                        float* ideal_shape, float* ideal_vel, float* landmark_pos, int32_t* step,
                        float* obs, int64_t obs_env_pitch, void* stream);

/* Scenario.reset_world of the landmark scenarios (basic_formation_env.py:54-65, formation_hd_partial_env.py:88-99,
 * formation_hd_partial_range_env.py:76-87, formation_hd_obs_env.py:101-120) continued on the device from each env's own legacy
 * NumPy MT19937 stream (mt_state uint32 [B][626] as for fg_reset_hd_mt): N agent positions, num_landmarks landmark positions,
 * num_obstacles obstacles from uniform([s_k, 2.0], [s_k+1, 2.5]) with the scenario's obstacle velocity - bit-exact with the
 * reference's draws.  Which envs: mask bytes (mask != NULL), every env (mask NULL, world_length <= 0), or the vec-env
 * worker's rule step[b] >= world_length (mask NULL, world_length > 0; env_wrappers.py:14-18). */
int fg_reset_scenario_mt(const FgScenario* scenario, int B, int N, const uint8_t* mask, int world_length, uint32_t* mt_state,
                         float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                         float* landmarks, float* obst_pos, float* obst_vel, int32_t* step, void* stream);

/* World.update_agent_state (core.py:279-286) for all B x N agents: state.c = action.c + c_noise * N(0,1) for a
 * non-silent agent, zeros for a silent one (agent_props[i][5] < 0; without a table every agent is non-silent and
 * noise-free).  dim_c = 2.  action_c, comm_state float [B][N][2]; the noise comes from the device counter RNG
 * (seed, env_index_base + b, agent, rng_offset: distributional parity, the reference draws np.random.randn). */
int fg_update_comm(const FgParams* params, int B, int N, const float* action_c, float* comm_state, void* stream);
/* The same for any World.dim_c (core.py:279-286 takes whatever the World says): action_c, comm_state float [B][N][dim_c].
 * dim_c = 2 is fg_update_comm itself; the fused formation_hd_env kernels read a communication block of dim_c = 2 only
 * (FgParams.comm_state), other widths serve the World API (World.step / update_agent_state). */
int fg_update_comm_dim(const FgParams* params, int B, int N, int dim_c, const float* action_c, float* comm_state, void* stream);

/* MultiAgentEnv.step for basic_formation_env (BASELINE config 1):
 * same physics; observation basic_formation_env.py:29-41, reward :43-52.
 *   landmarks float [B][L][2]; obs float [B][N][4 + 2L + 4(N-1)].
 * params->auto_reset: as for fg_step_scenario below. */
int fg_step_basic(const FgParams* params, int B, int N, int L, int do_physics,
                  float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                  const float* act, float* landmarks, int32_t* step,
                  float* obs, float* reward, float* indiv_reward, uint8_t* done,
                  int32_t* near_ag, void* stream);

/* MultiAgentEnv.step for formation_hd_partial_env / formation_hd_partial_range_env /
 * formation_hd_obs_env (and basic_formation_env), N + M <= 1024 (one env per lane group of a wave up to 64 entities,
 * per workgroup beyond):
 *   landmarks float [B][L][2]; obst_pos, obst_vel float [B][M][2] (updated in place, NULL if M = 0);
 *   obs float [B][N][D], D = 2 (+2 basic) + 2L + 2M + 2*nbr + 2(N-1), nbr = num_obs (partial) or N-1.
 * do_physics = 0 evaluates observation/reward/done on the current state (env.reset()).
 * params->auto_reset (with do_physics): the vec-env worker's rule inside the launch (env_wrappers.py:14-18) - an env whose
 * step counter reaches world_length restarts at once: agents, landmarks and obstacles are re-drawn exactly as
 * fg_reset_scenario draws them (same counter RNG: seed, env_index_base + b, rng_offset), step = 0, and the observation
 * written is the RESET observation while reward / indiv_reward / done keep the finished step's values.  `landmarks` is
 * written only then. */
int fg_step_scenario(const FgParams* params, const FgScenario* scenario, int B, int N, int do_physics,
                     float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                     const float* act, float* landmarks, float* obst_pos, float* obst_vel,
                     int32_t* step, float* obs, float* reward, float* indiv_reward, uint8_t* done,
                     void* stream);

/* K consecutive fg_step_scenario calls in ONE launch (the landmark scenarios' counterpart of fg_rollout_hd; basic_formation_env
 * with scenario->kind = FG_SCN_BASIC): the state stays on chip, results are bit-identical to K single-step launches with
 * rng_offset, rng_offset + 1, ... (params->auto_reset included).  act_seq float [K][B][N][2]; reward_seq, indiv_seq float
 * [K][B][N]; done_seq uint8 [K][B][N]; near_ag_seq int32 [K][B][L] (basic only, may be NULL); obs_seq float
 * [K / obs_every][B][N][D] (steps k with (k+1) % obs_every == 0).  K = 0 is a no-op. */
int fg_rollout_scenario(const FgParams* params, const FgScenario* scenario, int B, int N, int K,
                        float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                        const float* act_seq, float* landmarks, float* obst_pos, float* obst_vel,
                        int32_t* step, float* obs_seq, float* reward_seq, float* indiv_seq, uint8_t* done_seq,
                        int32_t* near_ag_seq, int obs_every, void* stream);

/* Scenario.reset_world of the landmark scenarios on device for the envs whose mask byte is non-zero (mask NULL = all),
 * counter-based RNG (basic_formation_env.py:54-65, formation_hd_partial_env.py:88-99, formation_hd_partial_range_env.py:76-87,
 * formation_hd_obs_env.py:101-114 draw distribution): agent and landmark positions ~ U(-1,1)^2, velocities 0, obstacle k
 * ~ U([s_k, 2.0], [s_k+1, 2.5]) with s = linspace(-1.8, 1.8, M + 1) and velocity (obstacle_vx, obstacle_vy), step = 0.
 * basic_formation_env: scenario->kind = FG_SCN_BASIC, num_landmarks = L, num_obstacles = 0. */
int fg_reset_scenario(const FgParams* params, const FgScenario* scenario, int B, int N, const uint8_t* mask,
                      float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                      float* landmarks, float* obst_pos, float* obst_vel, int32_t* step, void* stream);

/* Action decoding of MultiAgentEnv._set_action (environment.py:187-215) for the non-default action
 * modes, before the sensitivity scaling (which the step kernels apply).  `count` = B*N agents.
 *   FG_ACT_ONEHOT5 (discrete_action_space, :207-210): action float [count][5] -> u = (a1 - a2, a3 - a4)
 *   FG_ACT_INDEX   (discrete_action_input, :194-205): action int32 [count], 1:-x 2:+x 3:-y 4:+y, else 0
 *   FG_ACT_ARGMAX  (force_discrete_action, :212-216): action float [count][2] -> one-hot of the arg-max
 *                   (first maximum, as np.argmax); the one-hot is also written back to `action`,
 *                   as the reference overwrites the caller's array
 * u_out float [count][2] is what fg_step_hd & co. take as `act`. */
#define FG_ACT_ONEHOT5 1
#define FG_ACT_INDEX 2
#define FG_ACT_ARGMAX 3
int fg_decode_actions(int mode, int64_t count, void* action, float* u_out, void* stream);

/* The reference's built-in demo controller for formation_hd_env, all B envs in one launch:
 * formation_gym.get_action_BFS(formation_gym.ezpolicy, obs_n, per_layer) (formation_gym/__init__.py:49-99
 * driving :19-47; caller test.py:23).  N must be per_layer^L with 2 <= per_layer <= 8.
 *   obs  float [B][N][6N] as fg_step_hd / fg_observe_hd write it (only row 0 of every env is read: relative
 *        positions, ideal shape, ideal velocity); obs_env_stride = floats between consecutive envs, 0 = 6 N^2
 *   act  float [B][N][2] raw actions, what fg_step_hd takes as `act`. */
int fg_policy_bfs(int B, int N, int per_layer, const float* obs, int64_t obs_env_stride, float* act, void* stream);

/* The same controller evaluated straight from the simulator state (no observation buffer needed); bit-identical
 * to fg_policy_bfs on the observation fg_observe_hd / fg_step_hd writes for that state. */
int fg_policy_bfs_state(int B, int N, int per_layer, const float* pos_x, const float* pos_y,
                        const float* ideal_shape, const float* ideal_vel, float* act, void* stream);

/* Closed-loop rollout with the built-in controller: the loop of test.py:17-27
 *     act_n = get_action_BFS(ezpolicy, obs_n, per_layer); obs_n, ... = env.step(act_n)
 * for K steps and all B envs.  Like fg_rollout_hd, except that act_seq [K][B][N][2] is an OUTPUT (the actions
 * taken, required).  For N in {3, 9, 27, 81, 243} with per_layer = 3 the controller runs inside the pipelined
 * rollout kernels (ONE launch); otherwise K x (controller launch + step launch) are chained on `stream`.
 * Results equal K x (fg_policy_bfs on the last observation, fg_step_hd) bit for bit. */
int fg_rollout_hd_policy(const FgParams* params, int B, int N, int K, int per_layer,
                         float* pos_x, float* pos_y, float* vel_x, float* vel_y,
                         float* act_seq, float* ideal_shape, float* ideal_vel, int32_t* step,
                         float* obs_seq, float* reward_seq, float* indiv_seq, uint8_t* done_seq,
                         int obs_every, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FORMATION_HIP_H_ */
