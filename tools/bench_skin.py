"""Rate of the default fit at B = 4096 on the variants of the synthetic models: four weights per vertex, six (eight pairs:
pieces of up to eight joints since round 5), random joint sets (a piece per vertex), 16 betas."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel

dev = torch.device('cuda:0')
B = 4096
out = {}
for kind, nb in (('smpl', 10), ('smpl_w6', 10), ('smpl_rnd', 10), ('smpl_b16', 16), ('smplx', 10), ('smplx_w6', 10)):
    root = synth.ensure_model_root(kinds=(kind,))
    base = 'smplx' if kind.startswith('smplx') else 'smpl'
    model = BodyModel(base, 'neutral', model_root=f'{root}/{kind}', num_betas=nb, device=dev)
    fitter = BodyFitter(model)
    J = model.num_joints
    rs = np.random.RandomState(42)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
    fw = model(t(rs.randn(B, 3 * J) * 0.1), t(rs.randn(B, nb) * 0.5), t(rs.randn(B, 3)))
    tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
    kw = dict(num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs', 'shape_betas', 'trans'])
    for _ in range(3):
        fitter.fit(tv, tj, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fitter.fit(tv, tj, **kw)
    torch.cuda.synchronize()
    out[kind] = dict(fits_per_s=round(B * 10 / (time.perf_counter() - t0)), skin_width=int(model._native(dev).info.skin_width),
                     num_betas=nb, kernel_path=model.kernel_path())
    del model, fitter, fw, tv, tj
    torch.cuda.empty_cache()
print(json.dumps(out))
