"""Direct check of the posedirs GEMM output (workspace region `vposed` after smplfit_forward_f32):
determinism across runs (also with a noisy neighbour stream) and bf16x3 vs f32 element-wise."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import _lib, synth
from smplfitter_amd.pt import BodyModel
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=('smpl',))
m = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev)
h = m._native(dev); lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Vp = h.info.padded_vertices; Mp = (B + 127) // 128 * 128
rs = np.random.RandomState(1)
pose = torch.from_numpy((rs.randn(B, 72) * 0.3).astype(np.float32)).to(dev)
betas = torch.from_numpy((rs.randn(B, 10) * 0.5).astype(np.float32)).to(dev)
trans = torch.from_numpy(rs.randn(B, 3).astype(np.float32)).to(dev)
verts = torch.empty((B, 6890, 3), device=dev); joints = torch.empty((B, 24, 3), device=dev)
nbytes = h.workspace_bytes(B)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
def au(x, a): return (x + a - 1) // a * a
off_vposed = au(au(B*3*Vp*4, 256) + B*Vp*4, 256)   # tvs, vws, then vposed
noise_a = torch.empty(64 << 20, dtype=torch.float32, device=dev); noise_b = torch.empty_like(noise_a)
side = torch.cuda.Stream(device=dev)
def run(noisy):
    st = torch.cuda.current_stream(dev).cuda_stream
    if noisy:
        with torch.cuda.stream(side):
            for _ in range(4): noise_b.copy_(noise_a)
    _lib.check(lib.smplfit_forward_f32(h.ptr, C.c_void_p(pose.data_ptr()), None, C.c_void_p(betas.data_ptr()), 10,
               C.c_void_p(trans.data_ptr()), None, B, C.c_void_p(verts.data_ptr()), C.c_void_p(joints.data_ptr()), None,
               C.c_void_p(ws.data_ptr()), nbytes, C.c_void_p(st)))
    torch.cuda.synchronize()
    return ws[off_vposed: off_vposed + Mp*3*Vp*4].view(torch.float32).clone()
out = {}
for mode in ('f32', 'bf16x3'):
    os.environ['SMPLFIT_GEMM'] = mode
    runs = [run(noisy=(i % 2 == 1)) for i in range(8)]
    nd = [int((r != runs[0]).sum()) for r in runs[1:]]
    print(mode, 'elements differing from run 0 in runs 1..7:', nd, flush=True)
    if any(nd):
        i = int(np.argmax(nd)) + 1
        idx = torch.nonzero(runs[i] != runs[0]).flatten()
        rows = (idx // (3*Vp)).unique(); cols = (idx % (3*Vp))
        print('   rows (instances):', rows[:20].tolist(), '... n tiles:', (cols // 32).unique()[:20].tolist(),
              'max abs diff', float((runs[i]-runs[0]).abs().max()))
    out[mode] = runs[0]
d = (out['bf16x3'][:B*3*Vp] - out['f32'][:B*3*Vp]).abs()
print('bf16x3 vs f32: max abs diff %.3e, > 1e-6: %d of %d' % (float(d.max()), int((d > 1e-6).sum()), d.numel()))
