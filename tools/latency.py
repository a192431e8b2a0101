"""Latency of one fit call at small batch sizes, and throughput of the non-default configurations at 4096
(enable_kid, vertex weights, joints omitted): ms per call, fits/s."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel

dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=('smpl',))
model = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev)
J = model.num_joints
rs = np.random.RandomState(1)


def data(B):
    pose = torch.from_numpy((rs.randn(B, 3 * J) * 0.1).astype(np.float32)).to(dev)
    betas = torch.from_numpy((rs.randn(B, 10) * 0.5).astype(np.float32)).to(dev)
    trans = torch.from_numpy(rs.randn(B, 3).astype(np.float32)).to(dev)
    fw = model(pose, betas, trans)
    return fw['vertices'].contiguous(), fw['joints'].contiguous()


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


out = {}
fitter = BodyFitter(model)
keys = ['pose_rotvecs', 'shape_betas', 'trans']
for B in (1, 8, 64, 256, 1024):
    tv, tj = data(B)
    dt = timeit(lambda: fitter.fit(tv, tj, num_iter=3, beta_regularizer=1.0, requested_keys=keys), 100)
    # synchronous latency: the caller waits for the result of every call
    def sync_call():
        fitter.fit(tv, tj, num_iter=3, beta_regularizer=1.0, requested_keys=keys)
        torch.cuda.synchronize()
    dts = timeit(sync_call, 50)
    out[f'B{B}'] = dict(ms_per_call_pipelined=round(dt * 1e3, 4), ms_per_call_sync=round(dts * 1e3, 4), fits_per_s=round(B / dt))
B = 4096
tv, tj = data(B)
vw = torch.rand(B, model.num_vertices, device=dev) + 0.5
jw = torch.rand(B, J, device=dev) + 0.5
cases = dict(
    default=(fitter, dict(target_joints=tj)),
    kid=(BodyFitter(model, enable_kid=True), dict(target_joints=tj)),
    kid_reg0=(BodyFitter(model, enable_kid=True), dict(target_joints=tj, beta_regularizer=0.0)),
    weights=(fitter, dict(target_joints=tj, vertex_weights=vw, joint_weights=jw)),
    no_joints=(fitter, dict(target_joints=None)),
    share_beta=(fitter, dict(target_joints=tj, share_beta=True)),
    scale_target=(fitter, dict(target_joints=tj, scale_target=True)),
)
for name, (f, kw) in cases.items():
    kw = dict(dict(num_iter=3, beta_regularizer=1.0, requested_keys=keys), **kw)
    tjx = kw.pop('target_joints')
    try:
        dt = timeit(lambda: f.fit(tv, tjx, **kw), 10)
        out[f'cfg_{name}'] = dict(ms_per_call=round(dt * 1e3, 3), fits_per_s=round(B / dt))
    except Exception as e:  # noqa: BLE001
        out[f'cfg_{name}'] = dict(error=str(e)[:200])
print(json.dumps(out, indent=1))
