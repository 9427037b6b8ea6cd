# single-step launches at 243 x 8192: pipelined (writer threads, target workgroups) vs plain step_kernel
for cfg in "256 256" "256 512" "256 128" "512 256" "128 256"; do set -- $cfg
FG_TW=$1 FG_STEPWG=$2 python bench.py --mode step --agents 243 --envs 8192 --steps 60 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('243x8192 step tw=$1 wg=$2: %.1f us %.0f GB/s' % (d['ms_per_step']*1e3, d['roofline']['achieved']))"
done
FG_NOPIPE=1 python bench.py --mode step --agents 243 --envs 8192 --steps 60 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('243x8192 step nopipe: %.1f us %.0f GB/s' % (d['ms_per_step']*1e3, d['roofline']['achieved']))"
for tw in 128 256 512; do
FG_TW=$tw python bench.py --agents 243 --envs 8192 --steps 60 --warmup 8 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('243x8192 rollout tw=$tw: %.1f us/step %.0f GB/s' % (d['ms_per_step']*1e3, d['roofline']['achieved']))"
FG_TW=$tw python bench.py --agents 81 --envs 2048 --steps 400 --warmup 40 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('81x2048 rollout tw=$tw: %.1f us/step %.0f GB/s' % (d['ms_per_step']*1e3, d['roofline']['achieved']))"
done
