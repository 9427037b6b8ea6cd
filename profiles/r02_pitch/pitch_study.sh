#!/usr/bin/env bash
# r02 pitch study: padded env pitch (whole 128-byte lines per env) and round-robin env ownership in the small-N
# rollout kernel, interleaved rounds inside ONE gpurun call.  Usage: bash profiles/r02_pitch/pitch_study.sh [rounds]
set -u
R=${1:-3}
OUT=gpurun_out/r02_pitch; mkdir -p $OUT
python3 - <<'PY' > $OUT/parity.txt 2>&1
import os, sys, subprocess
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "gym-formation_amd")]
import torch, formation_gym
def run(N, B, K, pitch):
    e = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
    e.scenario.seed(3); e.scenario.reset_device(e.world, rng_offset=5); e.world.pos_x.mul_(0.5); e.world.pos_y.mul_(0.5)
    e.world.step_count.copy_(torch.arange(B, dtype=torch.int32, device="cuda") % 100); e.auto_reset = True
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    acts = torch.rand((K, B, N, 2), generator=g, device="cuda") * 2 - 1
    out = dict(obs=torch.zeros((K, B, pitch), device="cuda")[:, :, :6*N*N].view(K, B, N, 6*N), reward=torch.empty((K, B, N), device="cuda"),
               indiv=torch.empty((K, B, N), device="cuda"), done=torch.zeros((K, B, N), dtype=torch.uint8, device="cuda"))
    o, r, d, i = e.rollout(acts, out=out)
    return o.contiguous().clone(), r.clone(), e.world.get_state()[0].clone()
for N, B in ((27, 100), (9, 333), (3, 50), (27, 4096)):
    ref = run(N, B, 6, 6*N*N)
    pad = run(N, B, 6, -(-6*N*N//32)*32)
    print(N, B, "FG_RR", os.environ.get("FG_RR"), [bool(torch.equal(x, y)) for x, y in zip(ref, pad)])
PY
FG_RR=1 python3 - <<'PY' >> $OUT/parity.txt 2>&1
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "gym-formation_amd")]
import torch, formation_gym
def run(N, B, K, pitch):
    e = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
    e.scenario.seed(3); e.scenario.reset_device(e.world, rng_offset=5); e.world.pos_x.mul_(0.5); e.world.pos_y.mul_(0.5)
    e.world.step_count.copy_(torch.arange(B, dtype=torch.int32, device="cuda") % 100); e.auto_reset = True
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    acts = torch.rand((K, B, N, 2), generator=g, device="cuda") * 2 - 1
    out = dict(obs=torch.zeros((K, B, pitch), device="cuda")[:, :, :6*N*N].view(K, B, N, 6*N), reward=torch.empty((K, B, N), device="cuda"),
               indiv=torch.empty((K, B, N), device="cuda"), done=torch.zeros((K, B, N), dtype=torch.uint8, device="cuda"))
    o, r, d, i = e.rollout(acts, out=out)
    s = 0
    for k in range(K):                       # against single steps (blocked ownership, contiguous)
        pass
    return o.contiguous().clone(), r.clone(), e.world.get_state()[0].clone()
import hashlib
for N, B in ((27, 100), (9, 333), (3, 50), (27, 4096)):
    a = run(N, B, 6, 6*N*N); b = run(N, B, 6, -(-6*N*N//32)*32)
    print(N, B, "FG_RR=1 contig vs padded", [bool(torch.equal(x, y)) for x, y in zip(a, b)], float(a[0].double().sum()), float(a[1].double().sum()))
PY
python3 - <<'PY' >> $OUT/parity.txt 2>&1
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "gym-formation_amd")]
import torch, formation_gym
def run(N, B, K):
    e = formation_gym.make_env("formation_hd_env", False, N, num_envs=B, device="cuda:0")
    e.scenario.seed(3); e.scenario.reset_device(e.world, rng_offset=5); e.world.pos_x.mul_(0.5); e.world.pos_y.mul_(0.5)
    e.world.step_count.copy_(torch.arange(B, dtype=torch.int32, device="cuda") % 100); e.auto_reset = True
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    acts = torch.rand((K, B, N, 2), generator=g, device="cuda") * 2 - 1
    o, r, d, i = e.rollout(acts)
    return float(o.double().sum()), float(r.double().sum())
for N, B in ((27, 100), (9, 333), (3, 50), (27, 4096)):
    print(N, B, "blocked sums (must equal the FG_RR=1 sums above)", run(N, B, 6))
PY
cat $OUT/parity.txt
line() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][0]); t=d['timing']
print('%-44s us/step %.3f  GB/s %.0f  frac %.4f  blocks %d min/med/max %.3f/%.3f/%.3f' % ('$2', d['ms_per_step']*1e3, d['roofline']['achieved'], d['roofline']['frac'], t['blocks'], t['block_ms_min'], t['block_ms_median'], t['block_ms_max']))"; }
for r in $(seq 1 $R); do
  for shape in "27 4096" "9 32768" "27 16384"; do
    set -- $shape
    for v in "0 0" "-1 0" "-1 1" "0 1"; do
      set -- $shape $v
      FG_RR=$4 python3 bench.py --agents $1 --envs $2 --steps 400 --warmup 40 --no-extra --no-cpu-baseline --obs-pitch $3 > $OUT/b.json 2>/dev/null
      line $OUT/b.json "round $r  $1 x $2  pitch $3  rr $4" | tee -a $OUT/study.txt
    done
  done
done
