"""Experiment: split the batch over several HIP streams (separate workspaces) and see whether the
MFMA-bound GEMM of one chunk overlaps the VALU-bound vertex kernels of another."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel

dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=('smpl',))
model = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev)
fitter = BodyFitter(model)
B = 4096
rs = np.random.RandomState(42)
pose = torch.from_numpy((rs.randn(B, 72) * 0.1).astype(np.float32)).to(dev)
betas = torch.from_numpy((rs.randn(B, 10) * 0.5).astype(np.float32)).to(dev)
trans = torch.from_numpy(rs.randn(B, 3).astype(np.float32)).to(dev)
fw = model(pose, betas, trans)
tv, tj = fw['vertices'].contiguous(), fw['joints'].contiguous()
h = model._native(dev)

def run(nchunks, stagger_ms=0.0, steps=20):
    cb = B // nchunks
    streams = [torch.cuda.Stream() for _ in range(nchunks)]
    wss = [torch.empty(h.workspace_bytes(cb), dtype=torch.uint8, device=dev) for _ in range(nchunks)]
    tvs = [tv[i * cb:(i + 1) * cb].contiguous() for i in range(nchunks)]
    tjs = [tj[i * cb:(i + 1) * cb].contiguous() for i in range(nchunks)]
    def step():
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                fitter.fit(tvs[i], tjs[i], num_iter=3, beta_regularizer=1.0, _workspace=wss[i])
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'chunks={nchunks}: {dt*1e3:.3f} ms/step  {B/dt:,.0f} fits/s', flush=True)

for n in (1, 2, 4, 8):
    run(n)
