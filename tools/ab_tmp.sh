cd tools
for rep in 1 2; do
for v in "LTU_TAIL_BWD_WPS=3" "LTU_TAIL_BWD_WPS=4 LTU_TAIL_BWD_PREU=0" "LTU_TAIL_BWD_WPS=4 LTU_TAIL_BWD_PREU=1"; do echo "$v"; env $v python bench_tail.py 2>/dev/null | grep "d=128"; done
done
cd ..
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>/dev/null | python -c "import sys,json; L=sys.stdin.read().splitlines(); d=json.loads([l for l in L if l.startswith(chr(123))][0]); print('$*', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2; do run LTU_TAIL_BWD_WPS=3; run LTU_TAIL_BWD_WPS=4 LTU_TAIL_BWD_PREU=0; done
