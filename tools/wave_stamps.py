"""When do the waves of the residual pass start and finish?  Debug build (-DSMPLFIT_WAVE_STAMPS):
    tools/build_variant.sh wstamp -DSMPLFIT_WAVE_STAMPS          (here)
    SMPLFIT_LIB=build_ab/libwstamp.so python tools/wave_stamps.py [B]   (on the GPU box)
Every wave records the 100 MHz wall clock at entry, first step, last step, exit, and its HW_ID."""
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
os.environ.setdefault('SMPLFIT_CHUNKS', '1')
_lib.reload_options()
kw = dict(num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs'], _workspace=ws)
for _ in range(3):
    fitter.fit(tv, tj, **kw)
torch.cuda.synchronize()
lib = _lib.load()
st = torch.cuda.current_stream(dev).cuda_stream
ms = C.c_float()
assert lib.smplfit_time_kernel_f32(h.ptr, 3, B, 3, C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(st), C.byref(ms)) == 0
print(f'residual pass: {ms.value * 1e3:.1f} us by events')
mult = lib.smplfit_pick_share_mult(h.ptr, 0, B)
ncells = int(h.table('cell_counts')[0])
nw = (-(-B // 128) * 2) * (ncells // mult)
buf = np.zeros((nw, 5), np.uint64)
lib.smplfit_debug_wave_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.smplfit_debug_wave_stamps(buf.ctypes.data_as(C.c_void_p), nw) == 0
tt = buf[:, :4].astype(np.int64)
t0 = tt[:, 0].min()
us = (tt - t0) / 100.0
names = ['entry', 'first step', 'last step', 'exit']
print(f'B {B}: {nw} waves, mult {mult}; kernel span by stamps {us[:, 3].max():.1f} us')
for k, n in enumerate(names):
    q = np.percentile(us[:, k], [0, 1, 10, 50, 90, 99, 100])
    print(f'{n:>10}: ' + ' '.join(f'{x:7.1f}' for x in q) + '   (min p1 p10 p50 p90 p99 max, us from the first entry)')
d = us[:, 2] - us[:, 1]
print('loop time: ' + ' '.join(f'{x:7.1f}' for x in np.percentile(d, [0, 1, 10, 50, 90, 99, 100])))
pro = us[:, 1] - us[:, 0]
print('prologue : ' + ' '.join(f'{x:7.1f}' for x in np.percentile(pro, [0, 1, 10, 50, 90, 99, 100])))
hw = buf[:, 4]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64)
se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
for x in range(8):
    mk = xcc == x
    if mk.any():
        print(f'xcc {x}: {mk.sum():5d} waves, entry p50 {np.median(us[mk, 0]):6.1f} exit p50 {np.median(us[mk, 3]):6.1f} max {us[mk, 3].max():6.1f} loop p50 {np.median(d[mk]):6.1f}')
# running waves over time
edges = np.arange(0, us[:, 3].max() + 5, 5.0)
run = [(int(((us[:, 0] <= e) & (us[:, 3] > e)).sum()), int(((us[:, 1] <= e) & (us[:, 2] > e)).sum())) for e in edges]
print('t(us): resident / in-loop waves')
print(' '.join(f'{int(e)}:{a}/{b}' for e, (a, b) in zip(edges, run)))
# ---- loop time by share (cell cost model), by instance block, by CU
nblk = -(-B // 128) * 2
kbw = 4
w = np.arange(nw)
wave = w % kbw
wg = w // kbw
blk = wg % nblk
share = (wg // nblk) * kbw + wave
start = h.share_table(0, 0)
rec = h.share_table(0, 1).reshape(-1, 12)
print('share: pieces steps | loop mean  std  (us)')
rows = []
for s in range(ncells // mult):
    mk = share == s
    pcs = rec[start[s * mult]:start[(s + 1) * mult]]
    steps = int((pcs[:, 0] + (pcs[:, 0] & 1)).sum())
    rows.append((s, len(pcs), steps, d[mk].mean(), d[mk].std(), pro[mk].mean()))
for r in rows:
    print(f'{r[0]:4d}: {r[1]:3d} {r[2]:4d} | {r[3]:6.1f} {r[4]:5.1f}  prologue {r[5]:5.1f}')
A = np.array([[r[1], r[2], 1.0] for r in rows]); y = np.array([r[3] for r in rows])
coef, *_ = np.linalg.lstsq(A, y, rcond=None)
print(f'fit loop_us = {coef[0]:.2f} * pieces + {coef[1]:.3f} * steps + {coef[2]:.1f}; residual std {np.std(A @ coef - y):.2f} us; => piece cost {coef[0] / coef[1]:.1f} steps')
bm = np.array([d[blk == b].mean() for b in range(nblk)])
print('by instance block: mean loop', ' '.join(f'{x:.0f}' for x in bm))
cuid = xcc * 64 + se * 16 + cu
cm = np.array([d[cuid == c].mean() for c in np.unique(cuid)])
print(f'by CU ({len(cm)} ids): loop mean min {cm.min():.1f} p50 {np.median(cm):.1f} max {cm.max():.1f} std {cm.std():.1f}')
