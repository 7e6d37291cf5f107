#!/bin/bash
# prints ms_per_step of a bench run with the given environment assignments: tools/ms.sh LTU_X=1 LTU_Y=2
env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print('%s  %.3f ms' % (' '.join(sys.argv[1:]) or 'default', json.loads(sys.stdin.read())['ms_per_step']))" "$@"
