#!/usr/bin/env python3
"""Benchmark of the formation_gym hot path on MI355X.

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (MultiAgentEnv.step: _set_action ->
World.step -> observation/reward/done for every agent) over one batch of B
environments per GPU.  Two launch modes, both doing and writing exactly the
same per-step work (every observation of every step lands in HBM):
  rollout (default)  `fg_rollout_hd`: --chunk consecutive steps per launch with
                     pre-staged actions, outputs to [K,B,N,...] rollout buffers
                     (SURVEY.md 7.4 K4 / 8(d) "pre-staged [K,B,N,2]");
  step               `fg_step_hd`: one launch per env.step (what a policy in
                     the loop uses); reported beside it as `other_mode`.
Workload (BASELINE.json configs[2], the shape the north-star target is quoted
on): formation_hd_env, 27 agents x 4096 envs per GPU, fp32, synthetic
random-policy rollout: env b starts from np.random.RandomState(1 + 1000 b) in
the reference's reset draw order, actions iid U(-1,1) fp32 (seed 0, pre-staged
in HBM), episode length 100 with device-side auto-reset.  Environments are
independent, so N GPUs run N disjoint slices with no collective (weak scaling:
B per GPU fixed); the only communication is the timing barrier.

Timing.  A timed BLOCK is exactly --steps steps.  After the warm-up a series of
R consecutive blocks is timed (the rollout simply continues; R sized so that at
least --min-timed-ms are timed: a short `--steps 20` command is ONE 0.3 ms
launch per block and would otherwise be a single noisy sample), the series
bracketed by a barrier + torch.cuda.synchronize() on both sides, every block
delimited by HIP events on the launch stream, MAX over ranks per block.  ONE
clock feeds `value`, `ms_per_step` and `roofline.achieved`: the MEDIAN block on
the GPU's own timeline (gaps between launches included); min / max / count and
the host wall clock of the same series are reported beside it (`timing`).

`roofline.traffic` (HBM bytes per launch from the PMC counters) is measured in the same run on the same box at N = 1:
before this process touches the GPU, two short child runs of the workload under `rocprofv3 --pmc WRITE_SIZE` and
`--pmc FETCH_SIZE` (`live_traffic`; ~40 s; `--no-live-traffic` or any failure falls back to the figure of the committed
profile, `profiles/*_<N>x<B>*.json`, which is reported beside it either way).

Prints ONE JSON line (rank 0).  `value` = env-steps/s over all GPUs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "gym-formation_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0          # MI355X spec peak (MI355X_MICROARCH.md); 6290 measured copy
INFINITY_CACHE_BYTES = 256 << 20


def hbm_bytes_per_launch(n, envs, steps_per_launch):
    """Bytes ONE launch of `steps_per_launch` steps has to move through HBM (what the PMC counters should see): per step
    the observation 24 N^2, shared reward 4 N, individual reward 4 N, done N (written) and the actions 8 N (read); the
    state ONCE per launch (read pos, vel, ideal shape 8 N each, ideal_vel 8, step 4; written pos, vel 8 N each, step 4).
    SURVEY 8(d)'s formula charges the state round trip (32 N + ..) to EVERY step and leaves the individual reward out: it
    equals this at one step per launch up to those 4 N, and overstates a K-step launch, which keeps the state on chip."""
    per_step = 24 * n * n + 4 * n + 4 * n + n + 8 * n
    return envs * (steps_per_launch * per_step + 40 * n + 16)


def measured_traffic(n_agents, envs, mode, steps_per_launch):
    """HBM bytes per launch from the newest committed rocprofv3 PMC summary of this workload AND
    launch mode (profiles/*_<N>x<B>*.json, made by profiles/run_profile.sh + summarize.py)."""
    import glob
    best = (None, None)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%dx%d*.json" % (n_agents, envs)))):
        try:
            d = json.load(open(f))
            cfg = d["bench"]["config"]
        except Exception:
            continue
        if "hbm_traffic_bytes_per_launch" in d and cfg.get("mode") == mode and \
                cfg.get("steps_per_launch", 1) == steps_per_launch:
            best = (d["hbm_traffic_bytes_per_launch"], os.path.basename(f))
    return best


def live_traffic(a):
    """HBM bytes per launch of this run's kernel measured NOW, on this box: a short bench of the same workload is run as a
    CHILD process under `rocprofv3 --pmc`, once per counter (WRITE_SIZE, FETCH_SIZE: separate passes, KiB units, read side
    doubled on gfx950 - MI355X_MICROARCH.md, HBM section; the same arithmetic as profiles/summarize.py), before this process
    touches the GPU.  Returns (bytes or None, description).  Any failure (no rocprofv3, a pass that times out, an empty
    counter file) leaves the committed profile's figure in place - the reason is reported in `traffic_live_error`."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = next((p for p in ("/opt/rocm/bin/rocprofv3", shutil.which("rocprofv3")) if p and os.path.exists(p)), None)
    if exe is None:
        return None, "rocprofv3 not found"
    want = "rollout_kernel" if a.mode == "rollout" else "step_kernel"
    tmp = tempfile.mkdtemp(prefix="fg_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--agents", str(a.agents), "--envs", str(a.envs), "--mode", a.mode,
             "--chunk", str(a.chunk), "--steps", str(2 * a.chunk), "--warmup", str(a.chunk), "--min-timed-ms", "1",
             "--prewarm-ms", "0", "--no-cpu-baseline", "--no-extra", "--no-live-traffic",
             "--placement-candidates", str(a.placement_candidates)]
    kib, launches = {}, 0
    try:
        for counter in ("WRITE_SIZE", "FETCH_SIZE"):
            out = os.path.join(tmp, counter)
            proc = subprocess.Popen([exe, "--pmc", counter, "--output-format", "csv", "-d", out, "--"] + child, cwd="/tmp",
                                    env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                    start_new_session=True)
            try:
                rc = proc.wait(timeout=150)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)                # the session this call started, nothing else
                proc.wait()
                return None, "the %s pass did not finish in 150 s" % counter
            if rc != 0:
                return None, "the %s pass exited with %d" % (counter, rc)
            per_kernel = {}
            for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == counter and "fg::" in r.get("Kernel_Name", "") and want in r["Kernel_Name"]:
                        per_kernel.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            if not per_kernel:
                return None, "no %s rows for an fg:: %s" % (counter, want)
            vals = sorted(max(per_kernel.values(), key=len))       # the kernel with the most launches: the workload's own
            kib[counter] = vals[len(vals) // 2]
            launches = len(vals)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = int(1024 * (kib["WRITE_SIZE"] + 2.0 * kib["FETCH_SIZE"]))
    return total, ("live: rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE (separate passes, read side x 2) around a short run "
                   "of this workload on this box, median of %d launches" % launches)


def n1_reference(workload_key):
    """env-steps/s of a GLOBAL-batch config measured on ONE GPU of ANOTHER box in round 2
    (profiles/r02_n1_global_configs.json): reported for orientation only - the denominator of
    `scaling_efficiency_vs_n1` is measured in the same run (see global_configs below)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r02_n1_global_configs.json")))[workload_key]
    except Exception:
        return None


def _cpu_port_worker(args):
    """One env of the faithful per-env port, `steps` steps; returns elapsed seconds."""
    n_agents, steps, seed = args
    import numpy as np
    from oracle.formation_oracle import PortEnv
    env = PortEnv(n_agents)
    env.seed(seed)
    env.reset()
    acts = np.random.RandomState(seed).uniform(-1, 1, (steps, n_agents, 2))
    t0 = time.perf_counter()
    for t in range(steps):
        _, _, done, _ = env.step(list(acts[t]))
        if all(done):
            env.reset()
    return time.perf_counter() - t0


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            return next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    except OSError:
        return "unknown"


def usable_cores():
    """CPU cores this process can actually keep busy: the affinity mask, cut down to the cgroup's CPU quota
    (a GPU box hands a 16-core share of a 256-thread host to each lease: cpu.max = '1600000 100000')."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()])):
        try:
            quota, period = parse(open(path).read())
            if quota != "max" and int(quota) > 0:
                n = min(n, max(1, int(int(quota) / int(period))))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def cpu_baseline(n_agents, budget_s=12.0):
    """The oracle's per-env port (same loop structure as the reference) on the host
    cores: one env per process, bounded sample.  Reported baseline only."""
    import multiprocessing as mp
    host_cores = os.cpu_count() or 1
    cores = usable_cores()                                    # affinity mask and cgroup CPU quota of this process
    probe = _cpu_port_worker((n_agents, 2, 12345))          # seconds for 2 steps, 1 core
    per_step = max(probe / 2, 1e-4)
    steps = int(max(3, min(400, budget_s / per_step)))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        elapsed = pool.map(_cpu_port_worker, [(n_agents, steps, 1 + 1000 * r) for r in range(cores)])
    wall = max(elapsed)
    return {
        "value": round(cores * steps / wall, 2), "unit": "env-steps/s", "cores": cores, "host_cores": host_cores, "kind": "port",
        "sample": "%d envs x %d steps of formation_hd_env N=%d, one env per process "
                  "(oracle.PortEnv, numpy/scipy, fp64), wall = slowest worker %.1fs; "
                  "pool start-up excluded (%.1fs total)" % (cores, steps, n_agents, wall, time.perf_counter() - t0),
        "agent_steps_per_s": round(cores * steps * n_agents / wall, 1),
        "cpu_model": _cpu_model(),
        # the reference's files never travel to the GPU box; this port was timed against the REAL
        # reference in the build container (8 vCPU Xeon 2.1 GHz, one env per process):
        "calibration": "port / reference env-steps/s on the same 8 cores: 325 / 321 at N=27, 1130 / 1298 at N=9",
    }


def config1(dev):
    """BASELINE.json configs[0] / SURVEY 8(d) C1: basic_formation_env, 3 agents, 1 env, U(-1,1) actions, 1 000 steps
    with a reset every 50 - the plumbing case.  (a) the oracle's restatement on ONE host core, (b) the same loop
    through this build's reference-style list API (`make_env` / `env.step(list of arrays)`, one launch and one
    pinned copy each way per step)."""
    import numpy as np
    from oracle import formation_oracle as O
    steps = 1000
    acts = np.random.RandomState(1).uniform(-1, 1, (steps, 1, 3, 2))
    st = O.reset_basic(1, 3)
    t0 = time.perf_counter()
    ep = 0
    for t in range(steps):
        st, out = O.step_basic(st, acts[t])
        if out["done"].all():
            ep += 1
            st = O.reset_basic(1 + ep, 3)
    cpu_s = time.perf_counter() - t0
    import formation_gym
    env = formation_gym.make_env("basic_formation_env", False, 3, device=dev)
    env.seed(1)
    env.reset()
    a32 = acts.astype(np.float32)
    for t in range(20):                                                  # bindings, pinned buffers, clocks
        env.step([a32[t, 0, i].copy() for i in range(3)])
    env.reset()
    t0 = time.perf_counter()
    for t in range(steps):
        _, _, done_n, _ = env.step([a32[t, 0, i].copy() for i in range(3)])
        if all(done_n):
            env.reset()
    gpu_s = time.perf_counter() - t0
    return {"workload": "basic_formation_env, 3 agents x 1 env, 1000 steps, reset every 50 (BASELINE configs[0])",
            "cpu_port": {"env_steps_per_s": round(steps / cpu_s, 1), "cores": 1, "kind": "port",
                         "what": "oracle.step_basic (vectorised NumPy restatement, fp64) at B = 1"},
            "gpu_list_api": {"env_steps_per_s": round(steps / gpu_s, 1), "us_per_call": round(gpu_s / steps * 1e6, 1),
                             "what": "env.step(list of 3 arrays) -> lists: one fg_step_basic launch + one pinned copy "
                                     "each way per call, host resets (MT19937) every 50 steps; latency-bound by design"}}


def other_scenarios(dev):
    """The landmark scenarios at the reference's own shapes, 65536 envs: K-step rollout launches
    (`fg_rollout_scenario`, the one-env-per-lane kernels of csrc/fg_scn_lane_kernel.hpp) with device auto-reset into a PLACED
    buffer beyond the Infinity Cache (K chosen so that the observations of one launch exceed 1.1 GB): us per env step, the
    observation bytes written per second and ALL bytes the launch has to move (observations + rewards + done flags written,
    actions read: with 70-130 floats of observation per env these are 15-24 % on top).  Beside the headline, never `value`."""
    import torch
    import formation_gym
    lines = []
    for scenario, n, b in (("basic_formation_env", 3, 65536), ("formation_hd_partial_env", 5, 65536),
                           ("formation_hd_partial_range_env", 4, 65536), ("formation_hd_obs_env", 4, 65536)):
        env = formation_gym.make_env(scenario, False, n, num_envs=b, device=dev)
        env.seed(1)
        env.scenario.reset_device(env.world, rng_offset=999)
        env.auto_reset = True
        d = env._out["obs"].shape[-1]
        K = 20
        while K * b * n * d * 4 < 1.1e9:
            K += 20
        gen = torch.Generator(device=dev); gen.manual_seed(0)
        acts = (torch.rand((K, b, n, 2), generator=gen, device=dev) * 2 - 1).contiguous()
        out = env.alloc_rollout_buffers(K)
        for _ in range(3):
            env.rollout(acts, out=out)
        torch.cuda.synchronize()
        blocks = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                env.rollout(acts, out=out)
            e1.record()
            torch.cuda.synchronize()
            blocks.append(e0.elapsed_time(e1) / 3 / K)
        blocks.sort()
        ms = blocks[len(blocks) // 2]
        obs_bytes = b * n * d * 4
        all_bytes = obs_bytes + b * n * (8 + 4 + 4 + 1)
        lines.append({"workload": "%s, %d agents x %d envs, %d steps per launch, device auto-reset" % (scenario, n, b, K),
                      "observation_buffer_MB": round(K * obs_bytes / 1e6, 1),
                      "ms_per_step": round(ms, 6), "env_steps_per_s": round(b / (ms * 1e-3), 1),
                      "observation_GBps": round(obs_bytes / (ms * 1e-3) / 1e9, 1),
                      "observation_frac_of_hbm_peak": round(obs_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                      "all_bytes_GBps": round(all_bytes / (ms * 1e-3) / 1e9, 1),
                      "frac_hbm": round(all_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                      "placement": {k: (env.placement or {}).get(k) for k in ("arena_GB", "kept", "probe_seconds")},
                      "state_finite": bool(torch.isfinite(env.world.pos_x).all())})
        del env, out, acts
        torch.cuda.empty_cache()
    return lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--agents", type=int, default=27)
    ap.add_argument("--envs", type=int, default=4096, help="environments PER GPU")
    ap.add_argument("--mode", choices=["step", "rollout"], default="rollout",
                    help="step: one fg_step_hd launch per step; rollout: fg_rollout_hd, --chunk steps per launch")
    ap.add_argument("--chunk", type=int, default=20)
    ap.add_argument("--min-timed-ms", type=float, default=50.0,
                    help="repeat the timed block of --steps steps until this much GPU time has been timed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-auto-reset", action="store_true", help="tuning aid: episodes never reset")
    ap.add_argument("--obs-every", type=int, default=1,
                    help="tuning aid (rollout mode): write an observation only every n-th step; the JSON line is "
                         "then NOT a valid benchmark result")
    ap.add_argument("--no-extra", action="store_true", help="headline workload only (no other mode / shapes / config 1)")
    ap.add_argument("--no-small-buffer", action="store_true",
                    help="skip the extra 4-steps-per-launch (Infinity-Cache-sized buffer) measurement")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed stepping before the W warm-up steps, so that short runs are not measured during the clock ramp")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of the other BASELINE shapes")
    ap.add_argument("--obs-pitch", type=int, default=0,
                    help="rollout mode: floats between consecutive envs' observation blocks in the rollout buffer "
                         "(0 = contiguous 6 N^2; -1 = 6 N^2 rounded up to 32 floats = whole 128-byte lines per env)")
    ap.add_argument("--global-div", type=int, default=1,
                    help="test aid: run the GLOBAL-batch configs (BASELINE configs[3], [4]) with their batch sizes divided "
                         "by this, whatever the headline shape is (the lines are marked)")
    ap.add_argument("--placement-candidates", type=int, default=8,
                    help="observation buffers beyond the Infinity Cache: allocate up to this many candidates, time the launch "
                         "on each, keep the fastest (formation_gym/placement.py); 1 = no probe")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed profile instead of two rocprofv3 --pmc child runs on this box "
                         "(implied by --no-extra, by more than one rank, and when this process runs under a profiler)")
    ap.add_argument("--quick", action="store_true",
                    help="perf gate: the headline workload only (no other modes / shapes / scenarios, no CPU baseline, no counter "
                         "passes), ~15 s; with --max-ms-per-step the exit code says whether the rate held")
    ap.add_argument("--max-ms-per-step", type=float, default=0.0,
                    help="exit with status 3 (after printing the JSON line) when ms_per_step exceeds this")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the timing barrier (gloo: ranks may share a GPU, test only)")
    a = ap.parse_args()
    if a.quick:
        a.no_extra = a.no_cpu_baseline = a.no_live_traffic = True

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: start one rank per GPU as child processes (this process has not
        # touched the GPU and only relays the exit code)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode)

    # roofline.traffic measured on THIS box (N = 1 only), before this process initialises the GPU
    live = (None, None)
    profiled = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.gpus == 1 and not (a.no_live_traffic or a.no_extra or profiled) \
            and a.obs_every == 1:
        try:
            live = live_traffic(a)
        except Exception as exc:                                   # never let the side measurement take the bench down
            live = (None, "%s: %s" % (type(exc).__name__, exc))

    import numpy as np
    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = max(1, torch.cuda.device_count())
    gpu_share = max(1, -(-world_size // ndev))                   # ranks per GPU (1 on a real multi-GPU node)
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    # The only communication of this benchmark is the timing barrier and a MAX (no data-path
    # collective: environments are independent).  The default group is gloo over 127.0.0.1, always
    # available; with --backend nccl an RCCL group is created on top and used for the barrier if every
    # rank can bring it up (ranks sharing one GPU cannot: then all ranks agree to stay on gloo).
    sync_group, red_dev, sync_backend = None, None, "none"
    if world_size > 1:
        import datetime
        # generous: rank 0 times the GLOBAL batches alone while the others wait in a broadcast on this group
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=3600))
        sync_backend = "gloo"
        if a.backend == "nccl" and ndev < world_size:
            # ranks share a GPU (fewer devices than ranks): RCCL cannot form a communicator over duplicate devices and
            # its bring-up may hang rather than fail - decided from the device count alone, identically on every rank
            if rank == 0:
                print("bench: %d ranks on %d GPU(s): timing barrier stays on gloo" % (world_size, ndev), file=sys.stderr, flush=True)
        elif a.backend == "nccl":
            ok, why = 1, ""
            try:
                g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe, group=g)
                torch.cuda.synchronize()
                ok = int(float(probe[0]) == world_size)
            except Exception as exc:              # noqa: BLE001 - any RCCL bring-up failure means "use gloo"
                ok, why = 0, "%s: %s" % (type(exc).__name__, str(exc).splitlines()[0][:200] if str(exc) else "")
            agreed = torch.tensor([ok])
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            if int(agreed[0]) == 1:
                sync_group, red_dev, sync_backend = g, dev, "rccl"
            elif rank == 0:
                print("bench: RCCL group not available on every rank (%s); timing barrier stays on gloo" % why,
                      file=sys.stderr, flush=True)

    import formation_gym
    from formation_gym import _native, sharding

    def barrier():
        torch.cuda.synchronize()
        if world_size > 1 and not solo[0]:
            dist.barrier(group=sync_group)
        torch.cuda.synchronize()

    def per_rank(value):
        """The value of every rank, in rank order (host-side gather over the gloo group)."""
        if world_size == 1:
            return [float(value)]
        bucket = [None] * world_size
        dist.all_gather_object(bucket, float(value))
        return bucket

    def max_vec(xs):
        """Element-wise MAX over ranks of a list of floats."""
        if world_size == 1 or solo[0]:
            return list(xs)
        t = torch.tensor(xs, dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=sync_group)
        return t.tolist()

    def median(xs):
        s = sorted(xs)
        return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])

    gate_failed = [False]
    solo = [False]            # a measurement rank 0 takes ALONE (the others wait at a barrier): no collective inside

    def measure(N, B, mode, steps, warmup, chunk_req, other_steps=0, global_envs=None, policy=False):
        """Times blocks of `steps` env steps of N agents x B envs on this GPU in `mode` (and, if other_steps > 0,
        the other launch mode beside it).  global_envs: this rank owns its slice of a GLOBAL batch of that many envs
        (default: B per GPU, weak scaling).  policy: closed loop with the built-in controller (rollout_policy)."""
        # initial states: this rank owns the contiguous slice [lo, hi) of the global env range;
        # global env g is seeded 1 + 1000 g, so results do not depend on the GPU count
        ws = 1 if solo[0] else world_size
        env, lo, hi = sharding.make_env_shard("formation_hd_env", N, global_envs or B * ws, seed=1, rank=0 if solo[0] else rank,
                                              world_size=ws, local_rank=local_rank)
        B = hi - lo
        if B * N > 4000000:
            env.scenario.reset_device(env.world, rng_offset=999)   # host MT19937 streams: ~50 us per env and agent pair
        else:
            env.reset()
        env.auto_reset = not a.no_auto_reset                     # vec-env semantics: episodes restart on device
        env.world.step_count.zero_()

        # steps per rollout launch, bounded so that the [K,B,N,6N] rollout buffer stays under 48 GB (fewer steps per launch
        # cost more than a better placement gains: 243 x 8192 at 2 steps per launch 2.01 ms/step, at 4 1.83)
        # (ranks that SHARE a GPU - a rehearsal of N ranks on one device - share its memory too)
        chunk = max(1, min(chunk_req, int(48e9 / gpu_share // max(1, B * N * 6 * N * 4)) or 1))
        P = 3 * chunk if chunk >= 8 else (64 if B * N <= 4096 * 81 else 8)   # pre-staged action pool, cycled
        gen = torch.Generator(device=dev); gen.manual_seed(0 + rank)
        placed = {}
        # the solo leg of rank 0 (the N = 1 denominator of the scaling efficiency) places its buffer exactly as the sharded legs
        # do - same candidates, same memory fraction: an un-placed denominator against placed shards inflated the efficiency by
        # the placement gain, 5 % at 81 x 16384 and 10-20 % at 2048 envs per GPU; the probe takes 0.2-1 s
        candidates = a.placement_candidates
        if candidates > 1 and (mode == "step" or other_steps > 0) and not policy:
            placed["step"] = env.place_step_buffers(candidates=candidates, mem_fraction=0.5 / gpu_share)
        out = env._out
        act_pool, launchers = None, []
        if not policy:
            act_pool = (torch.rand((P, B, N, 2), generator=gen, device=dev) * 2 - 1).contiguous()
            launchers = [env.scenario.bind_step(env.world, act_pool[i], out, auto_reset=not a.no_auto_reset) for i in range(P)]

        def run_steps(n, start):
            for t in range(start, start + n):
                launchers[t % P](t)

        def run_rollout(n, start):
            t = start
            while t < start + n:
                k = min(chunk, start + n - t)
                lo_ = t % P
                if lo_ + k > P:
                    k = P - lo_
                env.rollout(act_pool[lo_:lo_ + k], out={k2: v[:k // (a.obs_every if k2 == "obs" else 1)] for k2, v in seq.items()
                                                        if torch.is_tensor(v)},
                            obs_every=a.obs_every)
                t += k

        def run_policy_rollout(n, start):
            t = start
            while t < start + n:
                k = min(chunk, start + n - t)
                env.rollout_policy(k, 3, out={k2: v[:k] for k2, v in seq.items() if torch.is_tensor(v)})
                t += k

        def run_policy_steps(n, start):
            obs = out["obs"]
            for t in range(start, start + n):
                obs = env.step(formation_gym.get_action_BFS(formation_gym.ezpolicy, obs, 3))[0]

        last_group = [1]                                       # blocks per event pair of the latest series

        def timed_once(fn, n, w, cursor, prewarm):
            # untimed: bring the GPU to its running clocks first (a short --steps/--warmup pair would otherwise
            # be measured during the DVFS ramp), then the W warm-up steps the caller asked for
            t_end = time.perf_counter() + (a.prewarm_ms * 1e-3 if prewarm else 0.0)
            while time.perf_counter() < t_end:
                fn(chunk, cursor)
                cursor += chunk
                if cursor % (8 * chunk) == 0:
                    torch.cuda.synchronize()
            fn(w, cursor)
            cursor += w
            # the cursor only selects the slice of the pre-staged action pool: continue on a launch boundary, so that a
            # block of `chunk` steps is ONE launch (a --warmup that is no multiple of it left every third block of the
            # driver's 20-step blocks split into a 15- and a 5-step launch)
            cursor = -(-cursor // chunk) * chunk
            # calibration (not reported): 4 consecutive blocks, the first one (which starts on an idle GPU) left out
            cal = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            barrier()
            cal[0].record()
            for c_ in range(4):
                fn(n, cursor)
                cursor += n
                cal[c_ + 1].record()
            torch.cuda.synchronize()
            t1 = cal[1].elapsed_time(cal[4]) / 3.0
            if not solo[0]:
                t1 = sharding.max_over_ranks(t1, red_dev, sync_group)   # same value on every rank
            R = int(min(4000, max(1, -(-1.25 * a.min_timed_ms // max(t1, 1e-3)))))   # 25 % margin: the calibration blocks run cold-ish
            # the timed series: R consecutive blocks of exactly n steps delimited by HIP events on the launch stream, the
            # series bracketed by barrier + device synchronize.  The stream never idles between blocks (the host queues
            # ahead), which is how a rollout loop runs; a block that started on an idle GPU would add the launch latency
            # of its first kernel to every sample.  An event between two launches costs the stream ~5 us (the barrier
            # packet and its signal: 27 x 4096 x 20 measured 2.4 % slower with an event after every 0.24 ms launch than with
            # one per 50 launches, host wall clock over the series included) - a cost of the measurement, not of the loop -
            # so blocks shorter than 5 ms share an event pair: G consecutive blocks per pair, each sample = pair / G.
            G = int(max(1, min(64, 5.0 // max(t1, 1e-3))))
            last_group[0] = G
            Rg = -(-R // G)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(Rg + 1)]
            barrier()
            t0 = time.perf_counter()
            evs[0].record()
            for r_ in range(Rg):
                for _ in range(G):
                    fn(n, cursor)
                    cursor += n
                evs[r_ + 1].record()
            while not evs[Rg].query():                         # completion seen by polling the closing event
                pass
            wall = time.perf_counter() - t0
            torch.cuda.synchronize()
            barrier()
            local_blocks = [evs[r_].elapsed_time(evs[r_ + 1]) / G for r_ in range(Rg)]
            dev_blocks = max_vec(local_blocks)                 # per block: the slowest rank
            wall_blocks = [(wall if solo[0] else sharding.max_over_ranks(wall, red_dev, sync_group)) * 1e3 / (Rg * G)] * Rg
            return dev_blocks, wall_blocks, local_blocks, cursor

        def timed(fn, n, w):
            """`timed_once`, repeated while the series came out shorter than --min-timed-ms (a calibration taken while the
            clocks were still ramping oversizes the blocks' duration and undersizes R); the decision is taken on the
            MAX-over-ranks figures, so every rank takes it alike."""
            start = 0
            for attempt in range(3):
                dev_blocks, wall_blocks, local_blocks, start = timed_once(fn, n, w if attempt == 0 else 0, start, attempt == 0)
                if sum(dev_blocks) * last_group[0] >= 0.8 * a.min_timed_ms or len(dev_blocks) * last_group[0] >= 4000:
                    break
            return dev_blocks, wall_blocks, local_blocks

        seq = None
        if mode == "rollout" or other_steps > 0:
            f = dict(dtype=torch.float32, device=dev)
            pitch = a.obs_pitch if a.obs_pitch > 0 else (-(-6 * N * N // 32) * 32 if a.obs_pitch < 0 else 6 * N * N)
            if a.obs_every == 1:
                # the observation buffer is PLACED: candidates timed with this env's own launch, the fastest kept
                seq = env.alloc_rollout_buffers(chunk, obs_env_pitch=0 if pitch == 6 * N * N else pitch, policy=policy,
                                                candidates=candidates, mem_fraction=0.5 / gpu_share)
                placed["rollout"] = env.placement
            else:
                obs_buf = torch.empty((chunk, B, pitch), **f)[:, :, :6 * N * N].view(chunk, B, N, 6 * N)
                seq = dict(obs=obs_buf, reward=torch.empty((chunk, B, N), **f),
                           indiv=torch.empty((chunk, B, N), **f),
                           done=torch.zeros((chunk, B, N), dtype=torch.uint8, device=dev))
                if policy:
                    seq["act"] = torch.empty((chunk, B, N, 2), **f)
        if policy:
            env.scenario.observe_batch(env.world, {"obs": out["obs"], "reward": out["reward"]})
            fns = {"step": run_policy_steps, "rollout": run_policy_rollout}
        else:
            fns = {"step": run_steps, "rollout": run_rollout}
        dev_blocks, wall_blocks, local_blocks = timed(fns[mode], steps, warmup)
        blocks_per_event = last_group[0]
        side = {}
        if mode == "rollout" and not policy and a.obs_every == 1 and placed.get("rollout", {}).get("probed") and chunk <= P:
            # beside the timed series (never `value`): the same launch (a) into an ORDINARY allocation - what the round-3
            # default API gave - and (b) through the documented default API, env.rollout(action_seq) with no buffers passed,
            # which places its own buffer on first use
            def quick(fn, reps=5, per=8):                             # `per` launches per event pair, as the timed series
                for _ in range(5):
                    fn()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
                ev[0].record()
                for r_ in range(reps):
                    for _ in range(per):
                        fn()
                    ev[r_ + 1].record()
                torch.cuda.synchronize()
                ts = sorted(ev[r_].elapsed_time(ev[r_ + 1]) / per for r_ in range(reps))
                return ts[len(ts) // 2] / chunk
            f = dict(dtype=torch.float32, device=dev)
            plain = dict(obs=torch.empty((chunk, B, N, 6 * N), **f), reward=torch.empty((chunk, B, N), **f),
                         indiv=torch.empty((chunk, B, N), **f), done=torch.zeros((chunk, B, N), dtype=torch.uint8, device=dev))
            side["unplaced_ms_per_step"] = quick(lambda: env.rollout(act_pool[:chunk], out=plain))
            del plain
            env._roll_launchers.clear()
            torch.cuda.empty_cache()
            env.rollout(act_pool[:chunk])                            # first use: the env places its own buffer
            side["default_api_placement"] = {k: env.placement.get(k) for k in ("arena_GB", "kept", "probe_seconds", "kept_ms", "as_created_ms")}
            side["default_api_ms_per_step"] = quick(lambda: env.rollout(act_pool[:chunk]))
            env.close()
        bytes_per_env_step = _native.step_hd_bytes(N)
        med = median(dev_blocks)                                     # every block is already the MAX over ranks
        r = {"ms": med, "blocks": dev_blocks, "blocks_per_event": blocks_per_event, "wall_blocks": wall_blocks, "local_ms": median(local_blocks),
             "chunk": chunk, "B": B, "placement": placed, "side": side,
             "bytes_per_env_step": bytes_per_env_step, "extra": None,
             "GBps": bytes_per_env_step * B * steps / (med * 1e-3) / 1e9}
        if other_steps > 0:
            other = "rollout" if mode == "step" else "step"
            d2, w2, _ = timed(fns[other], other_steps, min(warmup, 40))
            m2 = median(d2)
            r["extra"] = {"mode": other, "steps": other_steps, "blocks": len(d2),
                          "env_steps_per_s": round(world_size * B * other_steps / (m2 * 1e-3), 1),
                          "ms_per_step": round(m2 / other_steps, 5),
                          "achieved_GBps": round(bytes_per_env_step * B * other_steps / (m2 * 1e-3) / 1e9, 1)}
            if other == "rollout":
                r["extra"]["chunk"] = chunk
            if policy:
                # the same launch-by-launch loop captured once in a hipGraph and replayed (FormationVecEnv.capture): what a
                # trainer-shaped loop (policy forward -> env.step, train/maddpg-v2/main.py:77-91) gets without writing
                # capture code; 20 steps per replay, device auto-reset inside the graph
                from formation_gym.vec_env import FormationVecEnv
                env.auto_reset = False                         # FormationVecEnv switches it on again ('device' mode)
                venv = FormationVecEnv(env, reset_mode="device")
                env.scenario.observe_batch(env.world, {"obs": out["obs"], "reward": out["reward"]})
                loop = venv.capture(lambda o, out=None: formation_gym.get_action_BFS(formation_gym.ezpolicy, o, 3, out=out), 20)
                dg, _, _ = timed(lambda n_, s_: [loop.replay() for _ in range(max(1, n_ // 20))], 20 * max(1, other_steps // 20),
                                 20)
                mg = median(dg)
                gsteps = 20 * max(1, other_steps // 20)
                r["extra"]["graph_replay"] = {
                    "what": "FormationVecEnv.capture(get_action_BFS(ezpolicy), 20): the launch-by-launch loop (fg_policy_bfs + "
                            "fg_step_hd per step) as one replayed hipGraph of 20 steps",
                    "ms_per_step": round(mg / gsteps, 5),
                    "env_steps_per_s": round(world_size * B * gsteps / (mg * 1e-3), 1)}
                del loop, venv
                # the demo loop one step at a time in ONE launch per step: fg_rollout_hd_policy with K = 1
                one = {k2: v[:1] for k2, v in seq.items() if torch.is_tensor(v)}
                d3, _, _ = timed(lambda n_, s_: [env.rollout_policy(1, 3, out=one) for _ in range(n_)], other_steps, min(warmup, 40))
                m3 = median(d3)
                r["extra"]["fused_single_step_launch"] = {
                    "what": "env.rollout_policy(1): controller + step in one launch per step",
                    "ms_per_step": round(m3 / other_steps, 5),
                    "env_steps_per_s": round(world_size * B * other_steps / (m3 * 1e-3), 1)}
        pos, _ = env.world.get_state()
        r["finite"] = bool(torch.isfinite(pos).all())
        del env, act_pool, launchers, seq, out
        torch.cuda.empty_cache()
        return r

    def shape_line(n2, b2, st2, m2, mode, global_envs=None):
        g = m2["GBps"]
        total_envs = global_envs if global_envs else m2["B"] * world_size
        line = {"workload": "formation_hd_env, %d agents x %d envs per GPU" % (n2, m2["B"]), "mode": mode,
                "steps": st2, "blocks": len(m2["blocks"]), "steps_per_launch": 1 if mode == "step" else m2["chunk"],
                "env_steps_per_s": round(total_envs * st2 / (m2["ms"] * 1e-3), 1),
                "agent_steps_per_s": round(total_envs * n2 * st2 / (m2["ms"] * 1e-3), 1),
                "ms_per_step": round(m2["ms"] / st2, 5), "achieved_GBps": round(g, 1),
                "frac_of_hbm_peak": round(g / HBM_PEAK_GBPS, 4), "other_mode": m2["extra"],
                "state_finite": m2["finite"]}
        spl2 = line["steps_per_launch"]
        hbm = hbm_bytes_per_launch(n2, m2["B"], spl2)
        line["hbm_bytes_per_launch"] = hbm
        line["frac_hbm"] = round(hbm / (m2["ms"] / st2 * spl2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        obs_buffer = (spl2 if mode == "rollout" else 1) * m2["B"] * 24 * n2 * n2
        line["observation_buffer_MB"] = round(obs_buffer / 1e6, 1)
        if obs_buffer <= 1.5 * INFINITY_CACHE_BYTES:
            line["cache_resident"] = ("the observation buffer (%.0f MB) is about the size of the 256 MiB Infinity Cache and is overwritten launch "
                                      "after launch: cache-absorbed stores, NOT an HBM-streaming figure" % (obs_buffer / 1e6))
        if any(v and v.get("probed") for v in m2["placement"].values()):
            line["placement"] = m2["placement"]
        tr, src = measured_traffic(n2, m2["B"], mode, line["steps_per_launch"])
        if tr:
            alg = m2["bytes_per_env_step"] * m2["B"] * line["steps_per_launch"]
            line["counter_to_algorithmic_bytes"] = round(tr / alg, 4)
            line["traffic_source"] = src
        return line

    N, B = a.agents, a.envs
    headline = (N, B) == (27, 4096)
    m = measure(N, B, a.mode, a.steps, a.warmup, a.chunk, 0 if a.no_extra else min(a.steps, 400))
    chunk, extra, finite = m["chunk"], m["extra"], m["finite"]
    if profiled and a.placement_candidates > 1 and not a.no_extra:
        # under rocprofv3 the memory an arena hands back never returns to the driver (profiles/r03_place/
        # arena_memory_check_rocprofv3.txt): only the headline buffers are placed in a profiled process
        a.placement_candidates = 1
        if rank == 0:
            print("bench: running under a profiler - the extra legs use ordinary allocations", file=sys.stderr)
    bytes_per_env_step = m["bytes_per_env_step"]
    ms_block = m["ms"]                                          # median block, HIP events, MAX over ranks

    # The same workload with a rollout buffer that (nearly) fits the 256 MiB Infinity Cache: 4 steps per launch, the
    # buffer overwritten launch after launch.  The store stream is then absorbed by the cache instead of streaming to
    # HBM, so this is NOT an HBM-roofline number; reported beside the headline (never as `value`) because a consumer
    # that reads the observations right after each launch sees this rate.
    small = None
    if world_size == 1 and not a.no_extra and not a.no_small_buffer and a.mode == "rollout" and a.chunk > 4 \
            and a.obs_every == 1 and headline:
        s4 = min(a.steps, 400)
        m4 = measure(N, B, "rollout", s4, min(a.warmup, 40), 4)
        small = {"steps_per_launch": m4["chunk"], "buffer_MB": round(m4["chunk"] * B * N * 6 * N * 4 / 1e6, 1),
                 "ms_per_step": round(m4["ms"] / s4, 5),
                 "env_steps_per_s": round(B * s4 / (m4["ms"] * 1e-3), 1), "algorithmic_GBps": round(m4["GBps"], 1),
                 "note": "observation buffer resident in the Infinity Cache and overwritten every launch: cache-absorbed "
                         "stores, not an HBM-streaming figure"}

    others, global_cfgs, closed_loop, c1, scn_lines = [], [], None, None, None
    if not a.no_extra and not a.no_other_configs and (headline or a.global_div > 1):
        if world_size == 1 and headline:
            # the other BASELINE.json per-GPU shapes, short runs in the same process (reported beside the headline
            # workload, never as `value`)
            # 9 x 4096 moves 10 MB per step: its rollout buffer reaches HBM size (1.02 GB) at 128 steps per launch; the
            # 20-steps-per-launch run (160 MB, Infinity-Cache resident) is reported under `cache_resident_20_steps`
            for n2, b2, st2, ch2 in ((9, 4096, 512, 128 if a.mode == "rollout" else a.chunk), (81, 2048, 200, a.chunk),
                                     (243, 8192, 24, a.chunk)):
                m2 = measure(n2, b2, a.mode, st2, max(4, st2 // 10), ch2, min(st2, 400))
                others.append(shape_line(n2, b2, st2, m2, a.mode))
                if n2 == 9 and a.mode == "rollout":
                    m3 = measure(n2, b2, a.mode, 400, 40, a.chunk, 0)
                    l3 = shape_line(n2, b2, 400, m3, a.mode)
                    others[-1]["cache_resident_20_steps"] = {k: l3[k] for k in (
                        "steps_per_launch", "ms_per_step", "env_steps_per_s", "achieved_GBps", "observation_buffer_MB", "cache_resident")}
            # closed loop with the built-in controller (get_action_BFS + ezpolicy, reference test.py:23) in the loop
            mp_ = measure(N, B, a.mode, min(a.steps, 200), 20, a.chunk, min(a.steps, 200), policy=True)
            closed_loop = shape_line(N, B, min(a.steps, 200), mp_, a.mode)
            closed_loop["workload"] += ", actions from the built-in BFS controller"
            closed_loop["note"] = ("rollout: fg_rollout_hd_policy, the controller runs inside the rollout kernel; step: one "
                                   "fg_policy_bfs + one fg_step_hd launch per step")
        # BASELINE.json configs[3] and configs[4]: GLOBAL batches of 81 x 16384 and 243 x 65536 envs cut into
        # world_size contiguous slices (strong scaling: the global batch is fixed, each rank owns 1/world_size)
        for n2, g2, st2 in ((81, 16384 // a.global_div, 40), (243, 65536 // a.global_div, 8)):
            per_gpu = g2 // world_size
            if per_gpu * n2 * 6 * n2 * 4 > 150e9 / gpu_share:     # the observation buffer of ONE step must fit beside the rest
                global_cfgs.append({"workload": "formation_hd_env, %d agents x %d envs GLOBAL" % (n2, g2),
                                    "skipped": "a slice of %d envs per GPU needs %.0f GB of observations per step"
                                               % (per_gpu, per_gpu * n2 * 6 * n2 * 4 / 1e9)})
                continue
            m2 = measure(n2, per_gpu, a.mode, st2, max(2, st2 // 10), a.chunk, 0, global_envs=g2)
            line = shape_line(n2, per_gpu, st2, m2, a.mode, global_envs=g2)
            line["workload"] = "formation_hd_env, %d agents x %d envs GLOBAL over %d GPU(s) (BASELINE configs[%d])" % (
                n2, g2, world_size, 3 if n2 == 81 else 4)
            line["scaling"] = "strong"
            if a.global_div > 1:
                line["test_scale_div"] = a.global_div
            line["envs_per_gpu"] = m2["B"]
            ranks_ms = per_rank(m2["local_ms"])                   # each rank's own median block
            line["per_rank_ms_per_step"] = [round(x / st2, 5) for x in ranks_ms]
            line["slowest_rank"] = int(max(range(world_size), key=lambda r_: ranks_ms[r_]))
            # denominator of the scaling efficiency: the SAME global batch on ONE GPU, measured in THIS run on THIS
            # node - at world_size 1 that is the line itself (efficiency 1.0 by construction); at world_size > 1 rank 0
            # times it alone while the other ranks wait at the barrier (both global batches fit one 288 GB GPU)
            if world_size == 1:
                ref = line["env_steps_per_s"]
            else:
                ref_t = torch.zeros(1, dtype=torch.float64)
                if rank == 0:
                    solo[0] = True
                    try:
                        m1 = measure(n2, g2, a.mode, st2, max(2, st2 // 10), a.chunk, 0, global_envs=g2)
                        ref_t[0] = g2 * st2 / (m1["ms"] * 1e-3)
                        n1_placement = (m1.get("placement") or {}).get("rollout" if a.mode == "rollout" else "step") or {}
                        line["n1_placement"] = {k: n1_placement.get(k) for k in ("probed", "kept", "arena_GB", "kept_ms", "as_created_ms")}
                        if n1_placement.get("probed") and n1_placement.get("kept_ms") and n1_placement.get("as_created_ms"):
                            # what the same N = 1 run would have given on an ordinary allocation (the probe's own two timings)
                            line["n1_unplaced_over_placed"] = round(n1_placement["as_created_ms"] / n1_placement["kept_ms"], 4)
                    except Exception as exc:          # noqa: BLE001 - e.g. out of memory on a shared GPU: no denominator
                        print("bench: N = 1 run of %d x %d failed: %r" % (n2, g2, exc), file=sys.stderr, flush=True)
                    finally:
                        solo[0] = False
                dist.broadcast(ref_t, src=0)           # gloo group: also the barrier the other ranks wait at
                ref = float(ref_t[0]) or None
            line["n1_env_steps_per_s"] = round(ref, 1) if ref else None
            line["n1_same_run"] = True
            line["scaling_efficiency_vs_n1"] = round(line["env_steps_per_s"] / (world_size * ref), 4) if ref else None
            line["scaling_denominator"] = "the same global batch on ONE GPU in this run, its buffer placed like the shards' (round 4: un-placed)"
            if ref and line.get("n1_unplaced_over_placed"):
                line["scaling_efficiency_vs_n1_unplaced"] = round(line["scaling_efficiency_vs_n1"] * line["n1_unplaced_over_placed"], 4)
            line["n1_other_box_r02"] = n1_reference("%dx%d" % (n2, g2))
            global_cfgs.append(line)
        if world_size == 1 and headline:
            c1 = config1(dev)
            scn_lines = other_scenarios(dev)

    if rank == 0:
        blocks, walls = m["blocks"], m["wall_blocks"]
        value = world_size * B * a.steps / (ms_block * 1e-3)
        achieved = m["GBps"]                                      # GB/s per GPU, same clock as `value`
        cfg = _native.kernel_config(N)
        spl = 1 if a.mode == "step" else chunk
        alg_launch = bytes_per_env_step * B * spl
        traffic, traffic_src = measured_traffic(N, B, a.mode, spl)
        committed = (traffic, traffic_src)
        if live[0]:
            traffic, traffic_src = live
        res = {
            "metric": "env-steps/sec", "value": round(value, 1), "unit": "env-steps/s",
            "agent_steps_per_s": round(value * N, 1),
            "n_gpus": world_size, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_block / a.steps, 6), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "formation_hd_env, %d agents x %d envs per GPU, random policy, "
                                   "episode 100 with device auto-reset" % (N, B),
                       "baseline_config": "BASELINE.json configs[2] (27 agents x 4096 envs, the shape the north-star target "
                                          "is quoted on); configs[1] and the per-GPU shapes of configs[3], configs[4] are "
                                          "under other_configs at N = 1, their GLOBAL batches under global_configs"
                                          if headline else "custom shape",
                       "agents": N, "envs_per_gpu": B, "global_envs": B * world_size, "mode": a.mode,
                       "parallelism": "env-batch sharded over %d GPU(s), no collective" % world_size,
                       "timing_barrier": sync_backend, "ranks_per_gpu": gpu_share,
                       "steps_per_launch": spl,
                       "kernel": ("fg::step_kernel<%d> T=%d E=%d" % (N, cfg["threads"], cfg["envs_per_wg"]))
                       if a.mode == "step" else "fg::rollout_kernel<%d> (producer/writer pipelined)" % N},
            "timing": {"clock": "HIP events on the launch stream delimiting consecutive blocks of --steps steps (the series "
                                "bracketed by barrier + device synchronize, MAX over ranks per block); blocks shorter than "
                                "5 ms share an event pair (blocks_per_event consecutive blocks, sample = pair / that: an event "
                                "after every launch costs the stream ~5 us); value, ms_per_step and roofline.achieved all "
                                "come from the MEDIAN sample",
                       "blocks": len(blocks) * m.get("blocks_per_event", 1), "blocks_per_event": m.get("blocks_per_event", 1),
                       "timed_ms_total": round(sum(blocks) * m.get("blocks_per_event", 1), 3),
                       "block_ms_median": round(ms_block, 5), "block_ms_min": round(min(blocks), 5),
                       "block_ms_max": round(max(blocks), 5),
                       "wall_block_ms_median": round(median(walls), 5),
                       "wall_env_steps_per_s_median": round(world_size * B * a.steps / (median(walls) * 1e-3), 1),
                       "note": "wall = host clock over the whole series / blocks (first launch latency and the completion "
                                "poll included once)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_launch,
                         # what this launch really has to move through HBM (the state once, not per step): the honest figure
                         "hbm_bytes_per_launch": hbm_bytes_per_launch(N, B, spl),
                         "frac_hbm": round(hbm_bytes_per_launch(N, B, spl) / (ms_block / a.steps * spl * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "avg_launch_us": round(ms_block * 1e3 / a.steps * spl, 3),
                         "frac_of_measured_copy_peak_6290": round(achieved / 6290.0, 4)},
            "state_finite": finite,
        }
        if any(v and v.get("probed") for v in m["placement"].values()):
            res["placement"] = m["placement"]
        if m["side"]:
            sd = m["side"]
            hb = hbm_bytes_per_launch(N, B, spl)
            res["roofline"]["frac_unplaced"] = round(bytes_per_env_step * B / (sd["unplaced_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            res["roofline"]["frac_hbm_unplaced"] = round(hb / spl / (sd["unplaced_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            res["default_api"] = {
                "what": "env.rollout(action_seq) with no buffers passed: the env places its own observation buffer on first use "
                        "(formation_gym/placement.py, small arena) and re-uses it",
                "ms_per_step": round(sd["default_api_ms_per_step"], 6),
                "vs_headline": round(sd["default_api_ms_per_step"] / (ms_block / a.steps), 4),
                "placement": sd["default_api_placement"],
                "ordinary_allocation_ms_per_step": round(sd["unplaced_ms_per_step"], 6)}
        if live[0]:
            res["roofline"]["traffic_committed_profile"] = {"bytes": committed[0], "source": committed[1]}
        elif live[1]:
            res["roofline"]["traffic_live_error"] = live[1]
        if traffic:
            # the SURVEY 8(d) formula counts the pos/vel round trip of every step; a K-step launch keeps the state in
            # registers, so the PMC counters see fewer bytes (committed profile: ratio below); both are reported
            ratio = traffic / alg_launch
            res["roofline"]["counter_to_algorithmic_bytes"] = round(ratio, 4)
            res["roofline"]["achieved_counter_bytes"] = round(achieved * ratio, 1)
            res["roofline"]["frac_counter_bytes"] = round(achieved * ratio / HBM_PEAK_GBPS, 4)
        if a.obs_every != 1:
            res["INVALID"] = "observations written only every %d-th step (tuning run)" % a.obs_every
        if extra:
            res["other_mode"] = extra
        if small:
            res["rollout_cache_resident_buffer"] = small
        if others:
            res["other_configs"] = others
        if closed_loop:
            res["closed_loop_policy"] = closed_loop
        if global_cfgs:
            res["global_configs"] = global_cfgs
        if c1:
            res["config1"] = c1
        if scn_lines:
            res["other_scenarios"] = scn_lines
        if world_size == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(N)
            if others:
                # the per-env port beside the other agent counts too (short samples; N = 243 costs ~1 s per
                # env-step and core, so it gets its minimum of 3 steps per process)
                res["cpu_baseline_other_shapes"] = []
                for n2 in (9, 81, 243):
                    cb = cpu_baseline(n2, budget_s=4.0)
                    res["cpu_baseline_other_shapes"].append({k: cb[k] for k in ("value", "unit", "cores", "host_cores", "kind", "sample", "agent_steps_per_s")}
                                                            | {"agents": n2})
        print(json.dumps(res), flush=True)
        if a.max_ms_per_step > 0 and res["ms_per_step"] > a.max_ms_per_step:
            print("bench: ms_per_step %.6f exceeds --max-ms-per-step %.6f" % (res["ms_per_step"], a.max_ms_per_step), file=sys.stderr)
            gate_failed[0] = True
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    if gate_failed[0]:
        sys.exit(3)


if __name__ == "__main__":
    main()
