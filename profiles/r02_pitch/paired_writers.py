import ctypes, os, sys, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "gym-formation_amd")]
import formation_gym
from formation_gym import _native
N, B, K = 27, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 20
libs = {}
for name in ("old", "base"):
    lib = ctypes.CDLL(os.path.join(os.getcwd(), "build", "lib_%s.so" % name))
    lib.fg_rollout_hd.restype = ctypes.c_int
    lib.fg_rollout_hd.argtypes = _native.SIGNATURES["fg_rollout_hd"][1]
    libs[name] = lib
env = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
env.scenario.reset_device(env.world, rng_offset=1)
w, sc = env.world, env.scenario
acts = torch.rand((K, B, N, 2), device="cuda") * 2 - 1
f = dict(dtype=torch.float32, device="cuda")
rew, ind = torch.empty((K, B, N), **f), torch.empty((K, B, N), **f)
done = torch.zeros((K, B, N), dtype=torch.uint8, device="cuda")
p = sc.params(w, True, 0)
def launch(lib, obs, off):
    p.rng_offset = off
    rc = lib.fg_rollout_hd(p, B, N, K, w.pos_x.data_ptr(), w.pos_y.data_ptr(), w.vel_x.data_ptr(), w.vel_y.data_ptr(),
                           acts.data_ptr(), sc.ideal_shape.data_ptr(), sc.ideal_vel.data_ptr(), w.step_count.data_ptr(),
                           obs.data_ptr(), rew.data_ptr(), ind.data_ptr(), done.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
def timed(lib, obs, reps=16):
    for i in range(3): launch(lib, obs, i)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    e[0].record()
    for r in range(reps):
        launch(lib, obs, 10 + r); e[r + 1].record()
    torch.cuda.synchronize()
    ms = sorted(e[r].elapsed_time(e[r + 1]) for r in range(reps))
    return ms[len(ms) // 2] * 1e3 / K
keep = []
print("27 x %d x %d-step launches; the same observation buffer timed with the plain (old) and the paced (base) writer" % (B, K))
for i in range(10):
    obs = torch.empty((K, B, N, 6 * N), **f)
    keep.append(obs)
    a = timed(libs["old"], obs); b = timed(libs["base"], obs); a2 = timed(libs["old"], obs); b2 = timed(libs["base"], obs)
    print("allocation %d at %#x   plain %.2f / %.2f   paced %.2f / %.2f us/step" % (i, obs.data_ptr(), a, a2, b, b2))

# sustained: 6 seconds each, rate per second
import time
obs = keep[-1]
for name in ("base", "old", "base"):
    lib = libs[name]
    torch.cuda.synchronize()
    rates = []
    off = 100
    for sec in range(6):
        t0 = time.perf_counter(); n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.perf_counter() - t0 < 1.0:
            for _ in range(50):
                launch(lib, obs, off); off += 1
            n += 50
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        rates.append(e0.elapsed_time(e1) * 1e3 / (n * K))
    print("sustained %s: us/step per second of running: %s" % ("paced" if name == "base" else "plain", " ".join("%.2f" % r for r in rates)))
