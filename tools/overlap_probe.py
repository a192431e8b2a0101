"""How much of the pair-Gram kernel hides under the residual pass when both are in flight (two streams, two threads
calling the timing hook)?  An upper bound for what a fused launch of the two could gain."""
import ctypes as C, os, sys, threading, time
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
t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
fw = model(t(rs.randn(B, 3 * J) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3)))
tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
h = model._native(dev)
ws = torch.empty(h.workspace_bytes(B), dtype=torch.uint8, device=dev)
for _ in range(3):
    fitter.fit(tv, tj, num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs'], _workspace=ws)
torch.cuda.synchronize()
lib = _lib.load()
os.environ['SMPLFIT_TIME_NOPRE'] = '1'  # the kernel alone (no producer in front)
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
def run(kid, st, reps, out, key):
    ms = C.c_float()
    t0 = time.perf_counter()
    rc = lib.smplfit_time_kernel_f32(h.ptr, kid, B, reps, C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(st.cuda_stream), C.byref(ms))
    torch.cuda.synchronize()
    out[key] = (rc, ms.value * 1e3, (time.perf_counter() - t0) * 1e6 / reps)
R = 200
out = {}
run(3, streams[0], R, out, 'residual alone')
run(6, streams[1], R, out, 'pair-gram alone')
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(3, streams[0], R, out, 'residual (both)')), threading.Thread(target=run, args=(6, streams[1], R, out, 'pair-gram (both)'))]
for x in th: x.start()
for x in th: x.join()
both = (time.perf_counter() - t0) * 1e6 / R
for k, v in out.items(): print(f'{k:20s} rc {v[0]} event {v[1]:7.1f} us  wall/rep {v[2]:7.1f} us')
print(f'both in flight: wall per pair of launches {both:.1f} us')
