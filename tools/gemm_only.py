"""Run only the posedirs GEMM a few times (time hook, kernel id 2) — target of rocprofv3 --pmc passes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import _lib, synth
from smplfitter_amd.pt import BodyFitter, BodyModel
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=('smpl',))
m = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev)
f = BodyFitter(m); h = m._native(dev); lib = _lib.load()
B = 4096
rs = np.random.RandomState(42)
fw = m(torch.from_numpy((rs.randn(B, 72) * 0.1).astype(np.float32)).to(dev), torch.from_numpy((rs.randn(B, 10) * 0.5).astype(np.float32)).to(dev),
       torch.from_numpy(rs.randn(B, 3).astype(np.float32)).to(dev))
os.environ['SMPLFIT_CHUNKS'] = '1'
ws = torch.empty(h.workspace_bytes(B), dtype=torch.uint8, device=dev)
f.fit(fw['vertices'], fw['joints'], num_iter=1, _workspace=ws)
ms = C.c_float()
for kid in [int(k) for k in os.environ.get('KIDS', '2').split(',')]:
    _lib.check(lib.smplfit_time_kernel_f32(h.ptr, kid, B, 5, C.c_void_p(ws.data_ptr()), ws.numel(),
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), C.byref(ms)))
    print('kernel', kid, 'ms', ms.value)
