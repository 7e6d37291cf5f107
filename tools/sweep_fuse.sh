run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-families 2>/dev/null | python -c "import sys,json; L=sys.stdin.read().splitlines(); d=json.loads([l for l in L if l.startswith(chr(123))][0]); print('$*', round(d['ms_per_step'], 3), flush=True)"; }
for rep in 1 2; do
run LTU_NO_FUSE_QKV=1
run LTU_FUSE_QKV_MAX_TOKENS=1024
run LTU_FUSE_QKV_MAX_TOKENS=8640
run LTU_FUSE_QKV_MAX_TOKENS=21504
run LTU_FUSE_QKV_MAX_TOKENS=1000000
run LTU_FUSE_QKV_MIN_TOKENS=100000
run LTU_FUSE_QKV_MIN_TOKENS=8000 LTU_FUSE_QKV_MAX_TOKENS=9000
done
