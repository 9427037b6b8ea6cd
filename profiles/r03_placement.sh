#!/usr/bin/env bash
# Round 3: does the placement probe (formation_gym/placement.py) remove the allocation lottery?  Fresh PROCESSES of the
# same bench command, with the probe (default) and without (--placement-candidates 1), interleaved so
# that both arms see the same box state.  Usage on the GPU box: bash profiles/r03_placement.sh [runs]
RUNS="${1:-10}"
R=$PWD
OUT=$R/gpurun_out/r03_placement
rm -rf "$OUT"; mkdir -p "$OUT"
for cfg in "243 8192 4 24" "81 2048 20 200" "27 4096 20 1000"; do
  set -- $cfg
  if [ -n "${ONLY:-}" ] && [ "$1" != "$ONLY" ]; then continue; fi     # ONLY=27: one shape
  for i in $(seq 1 $RUNS); do
    for arm in 8 1; do
      timeout -k 10 200 python3 bench.py --agents $1 --envs $2 --chunk $3 --steps $4 --warmup $(($4 / 5)) --no-cpu-baseline --no-extra \
        --placement-candidates $arm > "$OUT/n$1_c${arm}_$i.json" 2> "$OUT/n$1_c${arm}_$i.err" || { echo "run failed: $cfg arm $arm"; tail -5 "$OUT/n$1_c${arm}_$i.err"; exit 1; }
    done
    echo "N=$1 round $i done"
  done
done
python3 - "$OUT" <<'PY' | tee "$R/gpurun_out/r03_placement.md"
import glob, json, os, sys
out = sys.argv[1]
print("# Placement probe: fresh bench.py processes on one box, with the probe (>= 8 spread selections) and without (--placement-candidates 1)\n")
print("| shape | arm | runs | TB/s min | median | max | spread (max/min - 1) | probe: worst/kept per run |")
print("|---|---|---|---|---|---|---|---|")
for n, shape in ((243, "243 x 8192, 4 steps/launch"), (81, "81 x 2048, 20 steps/launch"), (27, "27 x 4096, 20 steps/launch")):
    for arm in (8, 1):
        vals, ratios = [], []
        for f in sorted(glob.glob(os.path.join(out, "n%d_c%d_*.json" % (n, arm)))):
            lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
            if not lines:
                continue
            d = json.loads(lines[-1])
            vals.append(d["roofline"]["achieved"] / 1e3)
            p = d.get("placement", {}).get("rollout")
            if p and p.get("probed"):
                ratios.append("%.3f" % p["worst_over_kept"])
        if vals:
            s = sorted(vals)
            print("| %s | %s | %d | %.2f | %.2f | %.2f | %.1f %% | %s |" % (
                shape, "probe" if arm == 8 else "no probe", len(s), s[0], s[len(s) // 2], s[-1], (s[-1] / s[0] - 1) * 100, " ".join(ratios)))
PY
