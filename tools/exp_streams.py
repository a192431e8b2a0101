"""Experiment: process the batch in chunks, round-robin over several HIP streams (separate workspaces
per stream), to see (a) MFMA/VALU overlap across streams and (b) Infinity-Cache residency of a
chunk's streams when the chunk loop is outermost."""
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

def run(nstreams, chunk, steps=10):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    wss = [torch.empty(h.workspace_bytes(chunk), dtype=torch.uint8, device=dev) for _ in range(nstreams)]
    nch = B // chunk
    def step():
        for c in range(nch):
            s = c % nstreams
            with torch.cuda.stream(streams[s]):
                fitter.fit(tv[c * chunk:(c + 1) * chunk], tj[c * chunk:(c + 1) * chunk], num_iter=3,
                           beta_regularizer=1.0, _workspace=wss[s])
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'streams={nstreams} chunk={chunk}: {dt*1e3:.3f} ms/step  {B/dt:,.0f} fits/s', flush=True)

for ns, ch in ((1, 4096), (2, 1024), (4, 1024), (4, 512), (3, 512), (4, 256), (8, 256), (2, 512)):
    run(ns, ch)
