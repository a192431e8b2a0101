"""Where does k_solve_bm spend its time?  Debug build (-DSMPLFIT_SOLVE_STAMPS):
    tools/build_variant.sh sstamp -DSMPLFIT_SOLVE_STAMPS          (here)
    SMPLFIT_LIB=build_ab/libsstamp.so python tools/solve_stamps.py [B] [kind]   (on the GPU box)
Thread 0 of every workgroup records the 100 MHz wall clock at entry and behind every phase."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import _lib, synth
from smplfitter_amd.pt import BodyFitter, BodyModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else 'smpl'
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=(kind,))
model = BodyModel(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10, device=dev)
fitter = BodyFitter(model)
J = model.num_joints
rs = np.random.RandomState(42)
t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
fw = model(t(rs.randn(B, 3 * J) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3)))
tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
h = model._native(dev)
ws = torch.empty(h.workspace_bytes(B), dtype=torch.uint8, device=dev)
kw = dict(num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs'], _workspace=ws)
for _ in range(3):
    fitter.fit(tv, tj, **kw)
torch.cuda.synchronize()
lib = _lib.load()
st = torch.cuda.current_stream(dev).cuda_stream
ms = C.c_float()
assert lib.smplfit_time_kernel_f32(h.ptr, 4, B, 5, C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(st), C.byref(ms)) == 0
print(f'B {B} {kind}: solve {ms.value * 1e3:.1f} us by events')
nwg = min(4096, -(-B // 16))
buf = np.zeros((4096, 8), np.uint64)
lib.smplfit_debug_solve_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.smplfit_debug_solve_stamps(buf.ctypes.data_as(C.c_void_p), 4096) == 0
cyc = buf[:, 6:8].astype(np.int64)
tt = buf[:, :6].astype(np.int64)
ok = tt[:, 0] > 0
cyc = cyc[ok]
tt = tt[ok]
mhz = (cyc[:, 1] - cyc[:, 0]) / ((tt[:, 4] - tt[:, 0]) / 100.0)
print(f'shader clock over entry .. end of phase C: median {np.median(mhz):.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f})')
t0 = tt[:, 0].min()
us = (tt - t0) / 100.0
print(f'{len(tt)} workgroups; span by stamps {us[:, 5].max():.1f} us')
names = ['entry', 'A loads', 'B system', "B' rhs", 'C solve', 'D outputs']
for k, n in enumerate(names):
    q = np.percentile(us[:, k], [0, 10, 50, 90, 100])
    d = us[:, k] - us[:, k - 1] if k else us[:, 0]
    qd = np.percentile(d, [0, 10, 50, 90, 100])
    print(f'{n:>10}: at ' + ' '.join(f'{x:6.1f}' for x in q) + '   phase ' + ' '.join(f'{x:6.1f}' for x in qd) + '   (min p10 p50 p90 max, us)')
