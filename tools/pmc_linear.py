"""Runs exactly the forward transformer-projection launches of one bench step (128^3, 2 patches, bf16) so that
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE can be collected for them (one counter per pass).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python tools/pmc_linear.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python tools/pmc_linear.py
    python tools/pmc_linear.py --parse gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r01_pmc_linear.json
"""
import csv, glob, json, os, sys

STACKS = [(2 * 57408, 128), (2 * 10752, 256), (2 * 4320, 256), (2 * 512, 256)]      # (tokens, d) of the four transformers
LAYERS = 8


def launches():
    out = []
    for M, d in STACKS:
        for _ in range(LAYERS):
            out += [(M, d, 3 * d, 3), (M, d, d, 1), (M, d, 2 * d, -1), (M, 2 * d, d, 1)]      # nw = -1: FFN front half (GELU epilogue)
    return out


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lintransunet_amd import ops
    torch.manual_seed(0)
    cache = {}
    with torch.no_grad():
        for M, K, N, nw in launches():
            key = (M, K, N, nw)
            n_w = abs(nw)
            if key not in cache:
                x = torch.randn(M, K, device='cuda').bfloat16()
                ws = [torch.randn(N // n_w, K, device='cuda') * 0.05 for _ in range(n_w)]
                bs = [torch.zeros(N // n_w, device='cuda') for _ in range(n_w)]
                prep = ops.LinPrep([w.bfloat16() for w in ws], None)
                cache[key] = (x, ws, bs, prep)
            x, ws, bs, prep = cache[key]
            if nw < 0:
                ops.linear_gelu(x, ws[0], bs[0], 0.3, 7, prep=prep)
            else:
                ops.linear(x, ws, bs, prep=prep)
    torch.cuda.synchronize()


def parse(fetch_dir, write_dir):
    def total(d, name):
        tot, n = 0.0, 0
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] == name and ('linear_ring_bf16' in r['Kernel_Name'] or 'igemm_nt_bf16' in r['Kernel_Name']):
                    tot += float(r['Counter_Value'])
                    n += 1
        return tot, n
    fetch, nf = total(fetch_dir, 'FETCH_SIZE')
    write, nw = total(write_dir, 'WRITE_SIZE')
    L = launches()
    algo = sum((M * K + (2 if nw < 0 else 1) * M * N + N * K) * 2 for M, K, N, nw in L)      # the FFN front half writes u and h
    # MI355X_MICROARCH.md (HBM): counters are in KiB; FETCH_SIZE reports 1/2 of a wide coalesced read on gfx950
    hbm = (2.0 * fetch + write) * 1024.0
    print(json.dumps({'kernel': 'linear_ring_bf16_kernel (forward transformer projections of one bench step)',
                      'launches': len(L), 'dispatches_seen': [nf, nw], 'FETCH_SIZE_KiB': fetch, 'WRITE_SIZE_KiB': write,
                      'correction': 'hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950 FETCH_SIZE counts half of 16 B/lane reads)',
                      'hbm_bytes_total': hbm, 'hbm_bytes_per_launch': hbm / len(L),
                      'algorithmic_bytes_total': algo, 'algorithmic_bytes_per_launch': algo / len(L),
                      'traffic_over_algorithmic': hbm / algo}, indent=1))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--parse':
        parse(sys.argv[2], sys.argv[3])
    else:
        run()
