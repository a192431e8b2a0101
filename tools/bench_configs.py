"""fits/s for BASELINE.json configs 2-4 (SMPL 4096, SMPL-X 4096, SMPL 1024-subset 16384) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel

dev = torch.device('cuda:0')
root = synth.ensure_model_root(kinds=('smpl', 'smplx'))

def run(name, kind, B, subset=None, steps=10):
    kw = {}
    if subset is not None:
        m0 = BodyModel(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10)
        part = m0.weights.argmax(1).numpy()
        kw['vertex_subset'] = synth.subset_indices(m0.num_vertices, part, subset, 8, seed=1)
    m = BodyModel(kind, 'neutral', model_root=f'{root}/{kind}', num_betas=10, device=dev, **kw)
    f = BodyFitter(m)
    rs = np.random.RandomState(42)
    J = m.num_joints
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
    fw = m(t(rs.randn(B, 3 * J) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3)))
    tv, tj = fw['vertices'], fw['joints']
    h = m._native(dev)
    ws = torch.empty(h.workspace_bytes(B), dtype=torch.uint8, device=dev)
    for _ in range(3):
        f.fit(tv, tj, num_iter=3, beta_regularizer=1.0, _workspace=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = f.fit(tv, tj, num_iter=3, beta_regularizer=1.0, _workspace=ws)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'{name}: B={B} V={m.num_vertices} J={J}: {dt*1e3:.2f} ms/step, {B/dt:,.0f} fits/s', flush=True)

run('C2 smpl', 'smpl', 4096)
run('C3 smplx', 'smplx', 4096)
run('C4 smpl-1024', 'smpl', 16384, subset=1024)
if os.getenv('C4_SUBSET'):
    n = int(os.environ['C4_SUBSET'])
    run(f'C4x smpl-{n}', 'smpl', 16384, subset=n)
run('plumbing smpl B=32', 'smpl', 32)


def run_graphed(B=32, steps=200):
    """Small batches are launch-bound (21+ kernels per fit): the same call captured into a HIP graph."""
    m = BodyModel('smpl', 'neutral', model_root=f'{root}/smpl', num_betas=10, device=dev)
    f = BodyFitter(m)
    rs = np.random.RandomState(42)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)
    fw = m(t(rs.randn(B, 72) * 0.1), t(rs.randn(B, 10) * 0.5), t(rs.randn(B, 3)))
    tv, tj = fw['vertices'].clone(), fw['joints'].clone()
    ws = torch.empty(m._native(dev).workspace_bytes(B), dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f.fit(tv, tj, num_iter=3, _workspace=ws)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f.fit(tv, tj, num_iter=3, _workspace=ws)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'hipGraph replay smpl B={B}: {dt*1e3:.3f} ms/step, {B/dt:,.0f} fits/s', flush=True)


run_graphed(32)
run_graphed(256)
if os.getenv('GRAPH_BIG'):
    run_graphed(4096, steps=30)
    run_graphed(1024, steps=50)
