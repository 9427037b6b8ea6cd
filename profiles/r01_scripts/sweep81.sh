for tw in 128 256 512; do for wg in 128 256 512; do
FG_TW=$tw FG_STEPWG=$wg python bench.py --mode step --agents 81 --envs 2048 --steps 600 --warmup 60 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('81x2048 step tw=$tw wg=$wg: %.1f us %.0f GB/s' % (d['ms_per_step']*1e3, d['roofline']['achieved']))"
done; done
FG_NOPIPE=1 python bench.py --mode step --agents 81 --envs 2048 --steps 600 --warmup 60 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('81x2048 step nopipe: %.1f us %.0f GB/s' % (d['ms_per_step']*1e3, d['roofline']['achieved']))"
