"""Run-to-run determinism of chunked fits (two / three chunks: kernels of one chunk run beside the other chunk's GEMM):
    SMPLFIT_LIB=build_ab/libgemm4.so python tools/det_probe.py [smpl|smplx] [reps]
In a build that lets other kernels onto the split-bf16 GEMM's CUs (-DSMPLFIT_GEMM_SHARED_CU -DSMPLFIT_GEMM_WAVES=4) any
difference between repetitions is the neighbour interaction of DESIGN.md section 4."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import _lib, synth
from smplfitter_amd.pt import BodyFitter, BodyModel

kind = sys.argv[1] if len(sys.argv) > 1 else 'smpl'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=(kind,))
m = BodyModel(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10, device=dev)
f = BodyFitter(m)
J = m.num_joints
B = 4096
rs = np.random.RandomState(42)
t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
fw = m(t(rs.randn(B, 3 * J) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3)))
tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
for chunks in ('1', '2', '3'):
    os.environ['SMPLFIT_CHUNKS'] = chunks
    _lib.reload_options()
    kw = dict(num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs', 'shape_betas', 'trans'])
    ref = f.fit(tv, tj, **kw)
    torch.cuda.synchronize()
    bad = 0
    rows = collections.Counter()
    for _ in range(reps):
        r = f.fit(tv, tj, **kw)
        for k in ('pose_rotvecs', 'shape_betas', 'trans'):
            d = (r[k] != ref[k]).any(dim=1)
            n = int(d.sum())
            if n:
                bad += n
                rows.update((torch.nonzero(d).flatten().cpu().numpy() % 64).tolist())
    print(f'{kind} chunks {chunks}: {bad} result rows differ from the first fit in {reps} fits' + (f'; lanes (row % 64): {dict(sorted(rows.items()))}' if bad else ''), flush=True)
