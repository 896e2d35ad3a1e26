"""Debug aid: per-sample comparison of the device and the oracle on the small C5 render."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pytracer_amd as pa
from pytracer_amd import _abi as A, host, workloads
from pytracer_amd._lib import Context
import oracle_lib as O
xs, ys = int(sys.argv[1]), int(sys.argv[2])
tex = int(sys.argv[3])
flat = workloads.c5_head_measured(os.path.join(ROOT, 'tests', 'golden', 'head_mesh_scene.npz'))
ctx = Context(0); ctx.upload(flat)
cam, film = host.head_mesh_camera_film(48, 48)
smp = pa.sampler_desc(0, 48, 0, 48, xs, ys, True, 3)
tile = pa.tile_desc(8, 40, 8, 40, 0, xs * ys)
integ = pa.integrator_desc(A.PT_INTEG_PATH, 5, False)
n = 32 * 32 * xs * ys
acc = torch.zeros((film.y_cnt, film.x_cnt, 4), dtype=torch.float64, device='cuda')
L = torch.zeros((n, 3), dtype=torch.float64, device='cuda')
ctx.render(cam, film, smp, integ, tile, acc, sample_L=L)
torch.cuda.synchronize()
oacc, oL = O.OracleScene(flat).render(cam, film, smp, integ, tile)
L = L.cpu().numpy()
err = np.abs(L - oL).max(1)
rel = err / np.maximum(np.abs(oL).max(1), 1e-300)
order = np.argsort(-rel)[:12]
print('samples', n, 'n with rel > 1e-9:', int((rel > 1e-9).sum()), 'max abs', err.max(), 'max L', oL.max())
for i in order:
    print(i, L[i], oL[i], rel[i])
