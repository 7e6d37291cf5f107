"""Static check of the compiled kernels for serialised memory round trips: for every kernel of the given gfx950 .s files, the
number of global loads, stores, and of `s_waitcnt vmcnt(0)` that follow exactly one load (one load in flight: hipcc sinks loads
next to their first use unless the order is pinned - see tlayer.hip tl_issue / tl_run).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -c lintransunet_amd/csrc/X.hip -save-temps -o /tmp/X.o; python tools/asm_waits.py *.s"""
import re, sys

for f in sys.argv[1:]:
    t = open(f).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', t, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(';')]
        is_load = lambda l: l.startswith(('global_load', 'buffer_load', 'flat_load')) and ' lds' not in l
        nload = sum(1 for l in lines if is_load(l))
        nstore = sum(1 for l in lines if l.startswith(('global_store', 'buffer_store', 'flat_store')))
        ser = since = 0
        after_store = 0        # loads issued after a store in the same straight-line region and waited with vmcnt(0)
        stores_since = 0
        for l in lines:
            if is_load(l):
                since += 1
            if l.startswith(('global_store', 'buffer_store')):
                stores_since += 1
            mm = re.search(r'vmcnt\((\d+)\)', l)
            if mm:
                if int(mm.group(1)) == 0 and since == 1:
                    ser += 1
                    if stores_since:
                        after_store += 1
                since = 0
                stores_since = 0
        if nload >= 3 and ser >= 2:
            print(f'{f.split("/")[-1].split("-")[0]:10s} {name[:64]:64s} loads {nload:3d} stores {nstore:3d} one-load waits {ser:3d} (behind a store: {after_store})')
