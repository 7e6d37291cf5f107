"""Per-kernel-family hardware counters of the bench step (128^3, 2 patches, bf16), collected with rocprofv3 --pmc.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_a -- python3 tools/pmc_step.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_b -- python3 tools/pmc_step.py
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_c -- python3 tools/pmc_step.py
    python3 tools/pmc_step.py --parse /tmp/pmc_a /tmp/pmc_b /tmp/pmc_c > profiles/r02_pmc_step.json

The run itself is two eager training steps after two warm-up steps (counters are per dispatch; eager launches are what rocprofv3 can
attribute).  Corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KiB, FETCH_SIZE reports half of a wide
coalesced read on gfx950 (x2); GRBM_GUI_ACTIVE is summed over the 8 XCDs (/8 = cycles of the dispatch)."""
import csv, glob, json, os, re, sys

STEPS = 2


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lintransunet_amd.model import get_model_dict
    from lintransunet_amd import train, data
    dev = torch.device('cuda:0')
    torch.manual_seed(1234)
    model = get_model_dict('MaskTransUnet')([16, 32, 64, 128, 256], [100, 65, 40, 25, 10], [False, True, True, True, True], 1, 2,
                                            dropout=0.3, act_dtype=torch.bfloat16).to(dev).train()
    reducer = train.GradReducer(model, unused=train.UNUSED_PARAMETERS)
    weights = train.get_dynamic_weight(1)[0]
    x, lab = data.synthetic_patches(2, (128, 128, 128), 100, dev)
    for i in range(2 + STEPS):
        if i == 2:
            torch.cuda.synchronize()
            print('PMC_MARK measured steps begin', flush=True)
        reducer.zero_grad()
        train.train_step(model, x, lab, weights, reducer=reducer)
    torch.cuda.synchronize()


FAMILIES = [
    ('projection weight gradients (grouped ring kernel + fold)', r'wgrad_group_ring|wgroup_fold'),
    ('transformer layer chain kernels (forward + backward)', r'tail_fwd|tail_bwd'),
    ('projection forward / data gradient (weight-stationary ring)', r'linear_ring'),
    ('linear attention core', r'linattn_'),
    ('3x3x3 conv forward / data gradient (LDS halo)', r'conv3_halo_bf16|conv3_halo_ws'),
    ('3x3x3 conv weight gradient (LDS halo) + fold', r'conv3_wgrad_halo|wgrad_reduce_kernel'),
    ('class convolutions (sub-pixel un-embedding, strided dgrad)', r'conv_class_ring|upconv_wgrad|upconv_ring|sdgrad_ring'),
    ('implicit-GEMM convs (strided forward, gather weight gradient)', r'igemm_nt|wgrad_tn'),
    ('InstanceNorm', r'instnorm_'),
    ('attention gates (1x1x1 convs, gate kernels)', r'pw_small|gate_'),
]


def parse(dir_fetch, dir_write, dir_sq):
    def collect(d):
        rows = []
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            rows += list(csv.DictReader(open(f)))
        return rows
    out = {'steps_measured': STEPS, 'note': 'all dispatches of the run (2 warm-up + 2 measured steps) / 4 = per step',
           'corrections': 'HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; dispatch cycles = GRBM_GUI_ACTIVE / 8; '
                          'MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (dispatch cycles * 1024 SIMDs)', 'families': {}}
    fetch, write, sq = collect(dir_fetch), collect(dir_write), collect(dir_sq)
    nsteps = 2 + STEPS
    for name, pat in FAMILIES:
        rx = re.compile(pat)
        fam = {}
        f = sum(float(r['Counter_Value']) for r in fetch if r['Counter_Name'] == 'FETCH_SIZE' and rx.search(r['Kernel_Name']))
        w = sum(float(r['Counter_Value']) for r in write if r['Counter_Name'] == 'WRITE_SIZE' and rx.search(r['Kernel_Name']))
        n = sum(1 for r in fetch if r['Counter_Name'] == 'FETCH_SIZE' and rx.search(r['Kernel_Name']))
        fam['dispatches_per_step'] = n / nsteps
        fam['hbm_read_GB_per_step'] = 2.0 * f * 1024 / nsteps / 1e9
        fam['hbm_write_GB_per_step'] = w * 1024 / nsteps / 1e9
        c = {k: 0.0 for k in ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES')}
        for r in sq:
            if r['Counter_Name'] in c and rx.search(r['Kernel_Name']):
                c[r['Counter_Name']] += float(r['Counter_Value'])
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0
        fam['dispatch_cycles_per_step'] = cyc / nsteps
        fam['mfma_busy_cycles_per_step'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / nsteps
        fam['mfma_utilisation'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0) if cyc > 0 else None
        out['families'][name] = fam
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--parse':
        parse(*sys.argv[2:5])
    else:
        run()
