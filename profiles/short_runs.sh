for pw in 0 150 0 150; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --prewarm-ms $pw 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('prewarm $pw | steps 20: %.3f us/step %.0f GB/s' % (d['ms_per_step'] * 1e3, d['roofline']['achieved']))"
done
for pw in 0 150; do
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --prewarm-ms $pw 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('prewarm $pw | steps 100: %.3f us/step %.0f GB/s' % (d['ms_per_step'] * 1e3, d['roofline']['achieved']))"
done
