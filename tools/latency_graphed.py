import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel
dev = torch.device('cuda:0'); root = synth.ensure_model_root(kinds=('smpl',))
m = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev); f = BodyFitter(m)
for B, steps in ((1, 300), (32, 300), (256, 300), (1024, 100)):
    rs = np.random.RandomState(42); t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
    fw = m(t(rs.randn(B, 72) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3)))
    tv, tj = fw['vertices'].clone(), fw['joints'].clone()
    ws = torch.empty(m._native(dev).workspace_bytes(B), dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): f.fit(tv, tj, num_iter=3, _workspace=ws)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): f.fit(tv, tj, num_iter=3, _workspace=ws)
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f'hipGraph replay of one default fit call, B = {B}: {dt*1e3:.3f} ms per call, {B/dt:,.0f} fits/s', flush=True)
