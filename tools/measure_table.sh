#!/bin/bash
# Measurement table: one line per configuration, on one box.  Output: gpurun_out/table/.
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/table; mkdir -p $O
run() { python bench.py $2 --no-cpu-baseline --no-other-mode --no-configs --no-host-fed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'cfg': sys.argv[1], 'value': round(d['value']), 'ms': round(d['ms_per_step'],4), 'kernels': {k: round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()}, 'spec': d.get('speculation', {}).get('pll', {}).get('samples_per_span')}))" "$1" | tee -a $O/table.jsonl; }
rm -f $O/table.jsonl
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
run "fast 4096" ""
run "exact 4096" "--exact"
run "fast u8" "--u8"
run "exact u8" "--exact --u8"
for p in 0.01 0.1 1.0; do run "fast unlocked $p" "--unlocked-frac $p"; done
for p in 0.01 0.1 1.0; do run "exact unlocked $p" "--exact --unlocked-frac $p --steps 20"; done
run "exact unlocked 0.1 zero" "--exact --unlocked-frac 0.1 --unlocked-kind zero --steps 20"
run "exact unlocked 0.1 detuned" "--exact --unlocked-frac 0.1 --unlocked-kind detuned --steps 20"
run "exact 8192 unlocked 0.01" "--exact --channels 8192 --unlocked-frac 0.01 --steps 20"
run "exact 8192 unlocked 0.1" "--exact --channels 8192 --unlocked-frac 0.1 --steps 20"
run "exact 8192 unlocked 1.0" "--exact --channels 8192 --unlocked-frac 1.0 --steps 20"
run "exact 16384 unlocked 0.01" "--exact --channels 16384 --unlocked-frac 0.01 --steps 20"
run "exact 1024 unlocked 0.01" "--exact --channels 1024 --unlocked-frac 0.01 --steps 20"
run "exact 1024" "--exact --channels 1024"
run "fast unlocked 0.1 zero" "--unlocked-frac 0.1 --unlocked-kind zero"
run "fast unlocked 0.25 detuned" "--unlocked-frac 0.25 --unlocked-kind detuned"
run "fast unlocked 0.25 mix" "--unlocked-frac 0.25 --unlocked-kind mix"
run "fast deemph 50" "--deemphasis 50"
run "fast deemph 75" "--deemphasis 75"
run "fast deemph 150 (serial stage)" "--deemphasis 150"
run "exact deemph 50" "--exact --deemphasis 50"
run "fast 1024" "--channels 1024"
run "fast 2048" "--channels 2048"
run "fast 8192" "--channels 8192"
run "fast 16384" "--channels 16384 --steps 40"
run "exact 8192" "--exact --channels 8192"
run "exact 16384" "--exact --channels 16384 --steps 40"
run "fast 1.024M" "--fs 1024000"
run "fast 1.024M u8" "--fs 1024000 --u8"
run "exact 1.024M" "--exact --fs 1024000"
run "exact 1.024M u8" "--exact --fs 1024000 --u8"
run "fast 2.048M 2048ch" "--fs 2048000 --channels 2048"
run "fast no-pipeline" "--no-pipeline"
run "exact no-pipeline" "--exact --no-pipeline"
python bench.py --wideband 2>/dev/null | tail -1 | cut -c1-400
