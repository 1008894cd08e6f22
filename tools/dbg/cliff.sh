# Development: the exact mode's step with stations that cannot hold pilot lock (bench.py --unlocked-frac), one line per case
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],4), d.get("speculation",{}).get("pll"), {a: round(b,3) for a,b in (d.get("kernels_ms_per_step") or d["roofline"]["kernels_ms_per_step"]).items()})'
for a in "--exact" "--exact --unlocked-frac 0.01 --steps 20" "--exact --unlocked-frac 0.1 --steps 20" "--exact --unlocked-frac 1.0 --steps 20" "--exact --unlocked-frac 0.1 --unlocked-kind zero --steps 20" "--exact --unlocked-frac 0.1 --unlocked-kind detuned --steps 20"; do
  python bench.py $a --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "$P" "$a"
done
