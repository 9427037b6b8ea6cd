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


def cpu_baseline(n_agents, budget_s=12.0):
    """The oracle's per-env port (same loop structure as the reference) on the host
    cores: one env per process, bounded sample.  Reported baseline only."""
    import multiprocessing as mp
    cores = min(os.cpu_count() or 1, 32)
    probe = _cpu_port_worker((n_agents, 2, 12345))          # seconds for 2 steps, 1 core
    per_step = max(probe / 2, 1e-4)
    steps = int(max(3, min(400, budget_s / per_step)))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        elapsed = pool.map(_cpu_port_worker, [(n_agents, steps, 1 + 1000 * r) for r in range(cores)])
    wall = max(elapsed)
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    except OSError:
        pass
    return {
        "value": round(cores * steps / wall, 2), "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": "%d envs x %d steps of formation_hd_env N=%d, one env per process "
                  "(oracle.PortEnv, numpy/scipy, fp64), wall = slowest worker %.1fs; "
                  "pool start-up excluded (%.1fs total)" % (cores, steps, n_agents, wall, time.perf_counter() - t0),
        "agent_steps_per_s": round(cores * steps * n_agents / wall, 1),
        "cpu_model": model,
        # the reference's files never travel to the GPU box; this port was timed against the REAL
        # reference in the build container (8 vCPU Xeon 2.1 GHz, one env per process):
        "calibration": "port / reference env-steps/s on the same 8 cores: 325 / 321 at N=27, 1130 / 1298 at N=9",
    }


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-auto-reset", action="store_true", help="tuning aid: episodes never reset")
    ap.add_argument("--obs-every", type=int, default=1,
                    help="tuning aid (rollout mode): write an observation only every n-th step; the JSON line is "
                         "then NOT a valid benchmark result")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary rollout-mode measurement")
    ap.add_argument("--no-small-buffer", action="store_true",
                    help="skip the extra 4-steps-per-launch (Infinity-Cache-sized buffer) measurement")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed stepping before the W warm-up steps, so that short runs are not measured during the clock ramp")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of the other BASELINE per-GPU shapes (9 x 4096, 81 x 2048, 243 x 8192)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the timing barrier (gloo: ranks may share a GPU, test only)")
    a = ap.parse_args()

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
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    # The only communication of this benchmark is the timing barrier and a MAX (no data-path
    # collective: environments are independent).  The default group is gloo over 127.0.0.1, always
    # available; with --backend nccl an RCCL group is created on top and used for the barrier if every
    # rank can bring it up (ranks sharing one GPU cannot: then all ranks agree to stay on gloo).
    sync_group, red_dev, sync_backend = None, None, "none"
    if world_size > 1:
        import datetime
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
        sync_backend = "gloo"
        if a.backend == "nccl":
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
        if world_size > 1:
            dist.barrier(group=sync_group)
        torch.cuda.synchronize()

    def measure(N, B, mode, steps, warmup, chunk_req, other_steps):
        """Times `steps` env steps of N agents x B envs per GPU in `mode` (and, if other_steps > 0, the
        other launch mode beside it).  Returns wall seconds and HIP-event milliseconds, MAX over ranks."""
        # initial states: this rank owns the contiguous slice [lo, hi) of the global env range;
        # global env g is seeded 1 + 1000 g, so results do not depend on the GPU count
        env, lo, hi = sharding.make_env_shard("formation_hd_env", N, B * world_size, seed=1, rank=rank,
                                              world_size=world_size, local_rank=local_rank)
        env.reset()
        env.scenario._seed = 1 + rank                            # device auto-reset streams differ per rank
        env.auto_reset = not a.no_auto_reset                     # vec-env semantics: episodes restart on device
        env.world.step_count.zero_()

        # steps per rollout launch, bounded so that the [K,B,N,6N] rollout buffer stays under 48 GB
        chunk = max(1, min(chunk_req, int(48e9 // (B * N * 6 * N * 4)) or 1))
        P = 3 * chunk if chunk >= 8 else (64 if B * N <= 4096 * 81 else 8)   # pre-staged action pool, cycled
        gen = torch.Generator(device=dev); gen.manual_seed(0 + rank)
        act_pool = (torch.rand((P, B, N, 2), generator=gen, device=dev) * 2 - 1).contiguous()
        out = env._out
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
                env.rollout(act_pool[lo_:lo_ + k], out={k2: v[:k // (a.obs_every if k2 == "obs" else 1)] for k2, v in seq.items()},
                            obs_every=a.obs_every)
                t += k

        def timed(fn, n, w):
            # untimed: bring the GPU to its running clocks first (a short --steps/--warmup pair would otherwise
            # be measured during the DVFS ramp), then the W warm-up steps the caller asked for
            t_end = time.perf_counter() + a.prewarm_ms * 1e-3
            pre = 0
            while time.perf_counter() < t_end:
                fn(chunk, pre)
                pre += chunk
                if pre % (8 * chunk) == 0:
                    torch.cuda.synchronize()
            fn(w, pre)
            w += pre
            barrier()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record()
            fn(n, w)
            ev1.record()
            while not ev1.query():                             # completion seen by polling the closing event: the clock
                pass                                           # is read when the K steps are done, not when a blocked
            wall = time.perf_counter() - t0                    # host thread has been woken up; start barrier -> local completion
            torch.cuda.synchronize()
            barrier()
            dev_ms = ev0.elapsed_time(ev1)
            wall = sharding.max_over_ranks(wall, red_dev, sync_group)          # slowest rank
            dev_ms = sharding.max_over_ranks(dev_ms, red_dev, sync_group)
            return wall, dev_ms

        seq = None
        if mode == "rollout" or other_steps > 0:
            f = dict(dtype=torch.float32, device=dev)
            seq = dict(obs=torch.empty((chunk, B, N, 6 * N), **f), reward=torch.empty((chunk, B, N), **f),
                       indiv=torch.empty((chunk, B, N), **f),
                       done=torch.zeros((chunk, B, N), dtype=torch.uint8, device=dev))
        fns = {"step": run_steps, "rollout": run_rollout}
        wall, dev_ms = timed(fns[mode], steps, warmup)
        bytes_per_env_step = _native.step_hd_bytes(N)
        r = {"wall": wall, "dev_ms": dev_ms, "chunk": chunk, "bytes_per_env_step": bytes_per_env_step, "extra": None}
        if other_steps > 0:
            other = "rollout" if mode == "step" else "step"
            w2, d2 = timed(fns[other], other_steps, min(warmup, 40))
            r["extra"] = {"mode": other, "steps": other_steps,
                          "env_steps_per_s": round(world_size * B * other_steps / w2, 1),
                          "ms_per_step": round(w2 * 1e3 / other_steps, 5),
                          "achieved_GBps": round(bytes_per_env_step * B * other_steps / (d2 * 1e-3) / 1e9, 1)}
            if other == "rollout":
                r["extra"]["chunk"] = chunk
        pos, _ = env.world.get_state()
        r["finite"] = bool(torch.isfinite(pos).all())
        del env, act_pool, launchers, seq, out
        torch.cuda.empty_cache()
        return r

    N, B = a.agents, a.envs
    m = measure(N, B, a.mode, a.steps, a.warmup, a.chunk, 0 if a.no_extra else min(a.steps, 400))
    wall, dev_ms, chunk, extra, finite = m["wall"], m["dev_ms"], m["chunk"], m["extra"], m["finite"]
    bytes_per_env_step = m["bytes_per_env_step"]

    # The same workload with a rollout buffer that (nearly) fits the 256 MiB Infinity Cache: 4 steps per launch, the
    # buffer overwritten launch after launch.  The store stream is then absorbed by the cache instead of streaming to
    # HBM, so this is NOT an HBM-roofline number; reported beside the headline (never as `value`) because a consumer
    # that reads the observations right after each launch sees this rate.
    small = None
    if world_size == 1 and not a.no_extra and not a.no_small_buffer and a.mode == "rollout" and a.chunk > 4 \
            and a.obs_every == 1 and (N, B) == (27, 4096):
        m4 = measure(N, B, "rollout", min(a.steps, 400), min(a.warmup, 40), 4, 0)
        g4 = m4["bytes_per_env_step"] * B * min(a.steps, 400) / (m4["dev_ms"] * 1e-3) / 1e9
        small = {"steps_per_launch": m4["chunk"], "buffer_MB": round(m4["chunk"] * B * N * 6 * N * 4 / 1e6, 1),
                 "ms_per_step": round(m4["wall"] * 1e3 / min(a.steps, 400), 5),
                 "env_steps_per_s": round(B * min(a.steps, 400) / m4["wall"], 1), "algorithmic_GBps": round(g4, 1),
                 "note": "observation buffer resident in the Infinity Cache and overwritten every launch: cache-absorbed "
                         "stores, not an HBM-streaming figure"}

    # the other BASELINE.json per-GPU shapes, short runs in the same process (N = 1 only; reported beside the
    # headline workload, never as `value`)
    others = []
    if world_size == 1 and not a.no_extra and not a.no_other_configs and (N, B) == (27, 4096):
        for n2, b2, st2 in ((9, 4096, 400), (81, 2048, 200), (243, 8192, 24)):
            m2 = measure(n2, b2, a.mode, st2, max(4, st2 // 10), a.chunk, st2)
            g = m2["bytes_per_env_step"] * b2 * st2 / (m2["dev_ms"] * 1e-3) / 1e9
            others.append({"workload": "formation_hd_env, %d agents x %d envs per GPU" % (n2, b2), "mode": a.mode,
                           "steps": st2, "steps_per_launch": 1 if a.mode == "step" else m2["chunk"],
                           "env_steps_per_s": round(b2 * st2 / m2["wall"], 1),
                           "agent_steps_per_s": round(b2 * n2 * st2 / m2["wall"], 1),
                           "ms_per_step": round(m2["wall"] * 1e3 / st2, 5), "achieved_GBps": round(g, 1),
                           "frac_of_hbm_peak": round(g / HBM_PEAK_GBPS, 4), "other_mode": m2["extra"],
                           "state_finite": m2["finite"]})

    if rank == 0:
        launches = a.steps if a.mode == "step" else None
        achieved = bytes_per_env_step * B * a.steps / (dev_ms * 1e-3) / 1e9      # GB/s per GPU
        cfg = _native.kernel_config(N)
        res = {
            "metric": "env-steps/sec", "value": round(world_size * B * a.steps / wall, 1), "unit": "env-steps/s",
            "agent_steps_per_s": round(world_size * B * N * a.steps / wall, 1),
            "n_gpus": world_size, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(wall * 1e3 / a.steps, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "formation_hd_env, %d agents x %d envs per GPU, random policy, "
                                   "episode 100 with device auto-reset" % (N, B),
                       "baseline_config": "BASELINE.json configs[2] (27 agents x 4096 envs, the shape the north-star target "
                                          "is quoted on); configs[1] and the per-GPU shapes of configs[3], configs[4] are "
                                          "under other_configs at N = 1" if (N, B) == (27, 4096) else "custom shape",
                       "agents": N, "envs_per_gpu": B, "global_envs": B * world_size, "mode": a.mode,
                       "parallelism": "env-batch sharded over %d GPU(s), no collective" % world_size,
                       "timing_barrier": sync_backend,
                       "steps_per_launch": 1 if a.mode == "step" else chunk,
                       "kernel": ("fg::step_kernel<%d> T=%d E=%d" % (N, cfg["threads"], cfg["envs_per_wg"]))
                       if a.mode == "step" else "fg::rollout_kernel<%d> (producer/writer pipelined)" % N},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": measured_traffic(N, B, a.mode, 1 if a.mode == "step" else chunk)[0],
                         "traffic_source": measured_traffic(N, B, a.mode, 1 if a.mode == "step" else chunk)[1],
                         "algorithmic_bytes_per_launch": bytes_per_env_step * B * (1 if a.mode == "step" else chunk),
                         "avg_launch_us": round(dev_ms * 1e3 / a.steps * (1 if a.mode == "step" else chunk), 3),
                         "frac_of_measured_copy_peak_6290": round(achieved / 6290.0, 4)},
            "state_finite": finite,
        }
        if a.obs_every != 1:
            res["INVALID"] = "observations written only every %d-th step (tuning run)" % a.obs_every
        if extra:
            res["other_mode"] = extra
        if small:
            res["rollout_cache_resident_buffer"] = small
        if others:
            res["other_configs"] = others
        if world_size == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(N)
        print(json.dumps(res), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
