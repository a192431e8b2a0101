"""The calls of tools/asan_device.sh: every kernel family once, at sizes that exercise partial tiles (B not a multiple
of 64 / 128), two chunks, the kid unknown, weights, joints omitted, the fused conversion and the transfer kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import _lib, synth
from smplfitter_amd.pt import BodyConverter, BodyFitter, BodyModel

dev = torch.device('cuda:0')
print('library:', _lib.load().smplfit_version().decode(), flush=True)
root = synth.ensure_model_root(kinds=('smpl', 'smplx'))
os.environ['DATA_ROOT'] = synth.write_transfer_files('/tmp/smplfit_asan_data')
rs = np.random.RandomState(0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
models = {}
for kind, B in (('smpl', 1100), ('smplx', 200)):
    t0 = time.time()
    m = models[kind] = BodyModel(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10, device=dev)
    J = m.num_joints
    pose, betas, trans = t(rs.randn(B, 3 * J) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3))
    fw = m(pose, betas, trans)
    f, fk = BodyFitter(m), BodyFitter(m, enable_kid=True)
    r = f.fit(fw['vertices'], fw['joints'], num_iter=2, beta_regularizer=1.0)                    # batch-major path, 2 chunks
    r = f.fit(fw['vertices'], None, num_iter=2, beta_regularizer=1.0)                            # joints omitted
    r = fk.fit(fw['vertices'], fw['joints'], num_iter=2, beta_regularizer=1.0)                   # kid unknown
    n = min(B, 130)
    vw, jw = torch.rand(n, m.num_vertices, device=dev) + 0.5, torch.rand(n, J, device=dev) + 0.5
    r = f.fit(fw['vertices'][:n], fw['joints'][:n], vertex_weights=vw, joint_weights=jw, num_iter=2)   # weighted (wave path)
    r = f.fit(fw['vertices'][:n], fw['joints'][:n], num_iter=2, initial_pose_rotvecs=pose[:n], initial_shape_betas=betas[:n])
    r = f.fit_with_known_shape(betas[:n], fw['vertices'][:n], fw['joints'][:n], num_iter=2)
    r = f.fit_with_known_pose(pose[:n], fw['vertices'][:n], fw['joints'][:n])
    r = f.fit(fw['vertices'][:n], fw['joints'][:n], num_iter=2, share_beta=True)
    r = f.fit(fw['vertices'][:n], fw['joints'][:n], num_iter=2, scale_target=True)
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in r.values())
    print(kind, 'fits ok', round(time.time() - t0, 1), 's', flush=True)
for a, b, B in (('smpl', 'smplx', 200), ('smplx', 'smpl', 130), ('smpl', 'smpl', 70)):
    mi, mo = models[a], models[b]
    conv = BodyConverter(mi, mo)
    pose, betas, trans = t(rs.randn(B, 3 * mi.num_joints) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3))
    r = conv.convert(pose, betas, trans, num_iter=2)
    v = conv.convert_vertices(mi(pose, betas, trans)['vertices'])
    torch.cuda.synchronize()
    assert torch.isfinite(r['pose_rotvecs']).all() and torch.isfinite(v).all()
    print('convert', a, '->', b, 'fused' if conv._plan(dev) is not None else 'unfused', 'ok', flush=True)
print('ASAN_DEVICE_RUN_DONE', flush=True)
