"""A/B of library builds on one box: fits/s of the default fit and the per-kernel times
(smplfit_time_kernel_f32) for the library SMPLFIT_LIB points at (default: the in-tree build).

    SMPLFIT_LIB=build_ab/libsmplfit_r1.so python tools/ab_fit.py [smpl|smplx] [B]
"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import _lib, synth
from smplfitter_amd.pt import BodyFitter, BodyModel

kind = sys.argv[1] if len(sys.argv) > 1 else 'smpl'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=(kind,))
model = BodyModel(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10, device=dev)
fitter = BodyFitter(model)
J = model.num_joints
rs = np.random.RandomState(42)
pose = torch.from_numpy((rs.randn(B, 3 * J) * 0.1).astype(np.float32)).to(dev)
betas = torch.from_numpy((rs.randn(B, 10) * 0.5).astype(np.float32)).to(dev)
trans = torch.from_numpy(rs.randn(B, 3).astype(np.float32)).to(dev)
fw = model(pose, betas, trans)
tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
h = model._native(dev)
ws = torch.empty(h.workspace_bytes(B), dtype=torch.uint8, device=dev)
kw = dict(num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs', 'shape_betas', 'trans'], _workspace=ws)
for _ in range(5):
    r = fitter.fit(tv, tj, **kw)
torch.cuda.synchronize()
rates = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        r = fitter.fit(tv, tj, **kw)
    torch.cuda.synchronize()
    rates.append(B * 20 / (time.perf_counter() - t0))
lib = _lib.load()
st = torch.cuda.current_stream(dev).cuda_stream
kt = {}
for name, kid in (('gemm', 2), ('accum', 3), ('solve', 4), ('lbs', 5), ('pair_gram', 6), ('layout', 7), ('tmpl_psum', 8),
                  ('joint', 9), ('refine', 10), ('gram_comb', 11), ('psum_comb', 12), ('jdT', 13), ('lbs_last', 15), ('mean', 14)):
    ms = C.c_float()
    if lib.smplfit_time_kernel_f32(h.ptr, kid, B, 10, C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(st), C.byref(ms)) == 0:
        kt[name] = round(ms.value * 1e3, 1)
chk = float(r['pose_rotvecs'].double().abs().sum().item())
print(json.dumps(dict(lib=os.environ.get('SMPLFIT_LIB', 'in-tree'), env={k: v for k, v in os.environ.items() if k.startswith('SMPLFIT_') and k != 'SMPLFIT_LIB'},
                      kind=kind, B=B, fits_per_s=[round(x) for x in rates], kernel_us=kt, checksum=chk)), flush=True)
