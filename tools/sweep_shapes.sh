#!/bin/bash
# other workloads through the same bench.py (one box, one sitting): BASELINE configs 2 / 4, other per-GPU batches, eager launches, fp32 storage
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2> /tmp/err.txt | python -c "import sys,json; L=sys.stdin.read().splitlines(); J=[l for l in L if l.startswith(chr(123))]; d=json.loads(J[0]) if J else None; print('$*', (round(d['value'],1), 'patches/s', round(d['ms_per_step'],2), 'ms', d['config']['launch'], 'step_frac', d['roofline']['step_frac'] and round(d['roofline']['step_frac'],3)) if d else 'FAILED', flush=True)"; grep -q "Memory access fault" /tmp/err.txt && echo FAULT; tail -1 /tmp/err.txt | grep -i "error" ; }
run
run --size 96
run --classes 3
run --batch 1
run --batch 4
run --no-graph
run --dtype f32 --no-families
run --rehearse-comm
