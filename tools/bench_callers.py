"""Throughput of the hot path's callers (BodyConverter.convert, fit_with_known_shape / pose) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyConverter, BodyFitter, BodyModel

dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=('smpl',))
m = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev)
f = BodyFitter(m)
B = 4096
rs = np.random.RandomState(0)
t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
pose, betas, trans = t(rs.randn(B, 72) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3))
fw = m(pose, betas, trans)
conv = BodyConverter(m, m)


def timeit(name, fn, steps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'{name}: {dt*1e3:.2f} ms per {B}, {B/dt:,.0f} /s', flush=True)


timeit('BodyModel.forward', lambda: m(pose, betas, trans))
# cross-topology conversion on the synthetic transfer files (SMPL <-> the SMPL-X-shaped model)
os.environ['DATA_ROOT'] = synth.write_transfer_files('/tmp/smplfit_bench_data')
rootx = synth.ensure_model_root(kinds=('smplx',))
mx = BodyModel('smplx', 'neutral', model_root=f'{rootx}/smplx', num_betas=10, device=dev)
cx = BodyConverter(m, mx)
vin = m(pose, betas, trans)['vertices']
timeit('convert_vertices SMPL->SMPL-X (smplfit_transfer_f32)', lambda: cx.convert_vertices(vin))
timeit('BodyConverter(SMPL->SMPL-X).convert (num_iter=1)', lambda: cx.convert(pose, betas, trans, num_iter=1))
timeit('BodyConverter(SMPL->SMPL-X).convert (num_iter=3)', lambda: cx.convert(pose, betas, trans, num_iter=3))
del cx, mx, vin
torch.cuda.empty_cache()
timeit('BodyConverter.convert (num_iter=1)', lambda: conv.convert(pose, betas, trans, num_iter=1))
timeit('BodyConverter.convert (num_iter=3)', lambda: conv.convert(pose, betas, trans, num_iter=3))
timeit('fit joints omitted (num_iter=3)', lambda: f.fit(fw['vertices'], None, num_iter=3))
timeit('fit weighted (num_iter=3)', lambda: f.fit(fw['vertices'], fw['joints'], vertex_weights=torch.ones(B, 6890, device=dev), joint_weights=torch.ones(B, 24, device=dev), num_iter=3))
timeit('fit_with_known_shape (num_iter=3)', lambda: f.fit_with_known_shape(betas, fw['vertices'], fw['joints'], num_iter=3))
timeit('fit_with_known_pose', lambda: f.fit_with_known_pose(pose, fw['vertices'], fw['joints']))

# the differentiable fit (inputs that require gradients: the PyTorch restatement pt/_autograd.py, NOT the HIP path —
# recorded so that nobody mistakes one for the other); forward + backward of a scalar loss, batch 256
import warnings

warnings.simplefilter('ignore', RuntimeWarning)
Bg = 256
tvg = fw['vertices'][:Bg].clone().requires_grad_(True)
tjg = fw['joints'][:Bg].clone()


def grad_step():
    r = f.fit(tvg, tjg, num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs', 'shape_betas', 'trans'])
    (r['shape_betas'].square().sum() + r['trans'].sum()).backward()
    tvg.grad = None


for _ in range(2):
    grad_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    grad_step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f'fit with gradients (PyTorch restatement, forward + backward, num_iter=3): {dt*1e3:.1f} ms per {Bg}, {Bg/dt:,.0f} /s', flush=True)
