"""Rate of the default fit on the models of the GENERAL path (smplfit_info.vertex_path == 2): 32 betas, twelve skinning
weights per vertex (B = 4096) and every column of a 300-column file (num_betas=None, B = 256); forward rate beside it."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel

dev = torch.device('cuda:0')
out = {}
CASES = (('smpl_b32', 32, 4096), ('smpl_w12', 10, 4096), ('smpl_b300', None, 256))
if len(sys.argv) > 1:  # python tools/bench_general.py smpl_b300 [batch]
    CASES = tuple((k, nb, int(sys.argv[2]) if len(sys.argv) > 2 else B) for k, nb, B in CASES if k == sys.argv[1])
for kind, nb, B in CASES:
    root = synth.ensure_model_root(kinds=(kind,))
    model = BodyModel('smpl', 'neutral', model_root=f'{root}/{kind}', num_betas=nb, device=dev)
    fitter = BodyFitter(model)
    J, S = model.num_joints, model.num_betas
    rs = np.random.RandomState(42)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
    pose, betas, trans = t(rs.randn(B, 3 * J) * 0.1), t(rs.randn(B, S) * (0.5 if S <= 32 else 0.15)), t(rs.randn(B, 3))
    fw = model(pose, betas, trans)
    tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
    kw = dict(num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs', 'shape_betas', 'trans'])
    for _ in range(2):
        fitter.fit(tv, tj, **kw)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        fitter.fit(tv, tj, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(n):
        model(pose, betas, trans)
    torch.cuda.synchronize()
    dtf = time.perf_counter() - t0
    info = model._native(dev).info
    out[kind] = dict(batch=B, num_betas=S, skin_width=int(info.skin_width), vertex_path=int(info.vertex_path),
                     fits_per_s=round(B * n / dt), ms_per_fit_call=round(dt / n * 1e3, 3), forward_per_s=round(B * n / dtf))
    del model, fitter, fw, tv, tj
    torch.cuda.empty_cache()
print(json.dumps(out))
