"""Graph-replayed micro-benchmark of the data-side kernels at the reference's patch size (2 x 512 x 512 x 32 f32)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lintransunet_amd import data as P
from bench_nt import timed

n, shape = 2, (512, 512, 32)
img = torch.randn(n, 1, *shape, device='cuda')
mb = img.numel() * 4 / 1e6
mats = np.stack([P.rotate_matrix((0.2, -0.3, 0.25), shape)] * n)
m_dev = torch.as_tensor(mats.reshape(n, 12)).cuda()
z_dev = torch.tensor([[int(s * 0.8) for s in shape], [int(s * 1.25) for s in shape]], dtype=torch.int32).cuda()
g_dev = torch.tensor([0.7, 3.0]).cuda()
ws = torch.empty(2 * n, dtype=torch.int32, device='cuda')
out = torch.empty_like(img)
from lintransunet_amd import _lib
from lintransunet_amd.ops import _p, _s
H, W, D = shape
for name, f, traffic in (
        ('rotate (trilinear pull, border)', lambda: _lib.call('ltu_affine_sample', _p(img), _p(out), _p(m_dev), n, H, W, D, _s()), 2 * mb),
        ('zoom 0.8 / 1.25 (+ pad / crop)', lambda: _lib.call('ltu_zoom_sample', _p(img), _p(out), _p(z_dev), n, H, W, D, _s()), 2 * mb),
        ('contrast (min/max + pow)', lambda: _lib.call('ltu_adjust_contrast', _p(img), _p(out), _p(g_dev), _p(ws), n, H * W * D, _s()), 3 * mb)):
    t = timed(f)
    print(f'{name:34s}: {t:7.1f} us  ({traffic / t:.2f} TB/s algorithmic)', flush=True)
