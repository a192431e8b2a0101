"""Time the individual heavy kernels through smplfit_time_kernel_f32 (after one fit)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import _lib, synth
from smplfitter_amd.pt import BodyFitter, BodyModel
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=('smpl',))
model = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev)
fitter = BodyFitter(model)
B = int(os.environ.get('B', 4096))
rs = np.random.RandomState(42)
pose = torch.from_numpy((rs.randn(B, 72) * 0.1).astype(np.float32)).to(dev)
betas = torch.from_numpy((rs.randn(B, 10) * 0.5).astype(np.float32)).to(dev)
trans = torch.from_numpy(rs.randn(B, 3).astype(np.float32)).to(dev)
fw = model(pose, betas, trans)
h = model._native(dev)
ws = torch.empty(h.workspace_bytes(B), dtype=torch.uint8, device=dev)
fitter.fit(fw['vertices'], fw['joints'], num_iter=3, _workspace=ws)
torch.cuda.synchronize()
lib = _lib.load()
st = torch.cuda.current_stream(dev).cuda_stream
out = {}
for name, kid in (('gemm', 2), ('accum', 3), ('solve', 4), ('lbs', 5), ('pair_gram', 6), ('transpose', 7)):
    ms = C.c_float()
    _lib.check(lib.smplfit_time_kernel_f32(h.ptr, kid, B, 10, C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(st), C.byref(ms)))
    out[name] = round(ms.value * 1e3, 1)
print(os.environ.get('SMPLFIT_EXP', '0'), out, flush=True)
