"""Event time of the shape solve of a default fit, k_solve_bm against k_gram_combine_bm + k_shape_solve, over batch sizes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import _lib, synth
from smplfitter_amd.pt import BodyFitter, BodyModel
kind = sys.argv[1] if len(sys.argv) > 1 else 'smpl'
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=(kind,))
model = BodyModel(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10, device=dev)
fitter = BodyFitter(model)
J = model.num_joints
lib = _lib.load()
st = torch.cuda.current_stream(dev).cuda_stream
for B in [int(x) for x in (sys.argv[2:] or [1024, 2048, 4096, 8192, 16384, 32768])]:
    rs = np.random.RandomState(42)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
    fw = model(t(rs.randn(B, 3 * J) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3)))
    tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
    h = model._native(dev)
    ws = torch.empty(h.workspace_bytes(B), dtype=torch.uint8, device=dev)
    out = {}
    for flag in ('0', '1'):
        os.environ['SMPLFIT_SOLVE_BM'] = flag
        _lib.reload_options()
        for _ in range(3):
            fitter.fit(tv, tj, num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs'], _workspace=ws)
        torch.cuda.synchronize()
        tot = 0.0
        for kid in (11, 4):  # combine (unsupported with the fused kernel), solve
            ms = C.c_float()
            if lib.smplfit_time_kernel_f32(h.ptr, kid, B, 20, C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(st), C.byref(ms)) == 0:
                tot += ms.value
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fitter.fit(tv, tj, num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs'], _workspace=ws)
        e1.record(); torch.cuda.synchronize()
        out[flag] = (tot * 1e3, e0.elapsed_time(e1) / 10)
    print(f'{kind} B {B:6d}: two kernels {out["0"][0]:7.1f} us, k_solve_bm {out["1"][0]:7.1f} us; fit {out["0"][1]:.3f} -> {out["1"][1]:.3f} ms')
    del ws, tv, tj, fw
