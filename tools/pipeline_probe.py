"""Throughput of back-to-back default fits issued on ONE stream against the same fits issued alternately on TWO (or more)
streams with a workspace each: consecutive batches are independent, so the latency-bound third of one fit (joint stages,
solves, refinement) can run beside the bandwidth-bound passes of the next one.
    python tools/pipeline_probe.py [smpl|smplx] [B] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel

kind = sys.argv[1] if len(sys.argv) > 1 else 'smpl'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
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
kw = dict(num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs', 'shape_betas', 'trans'])
out = {}
ref = None
for ns in (1, 2, 3, 4):
    streams = [torch.cuda.current_stream(dev)] if ns == 1 else [torch.cuda.Stream(device=dev) for _ in range(ns)]
    wss = [torch.empty(h.workspace_bytes(B), dtype=torch.uint8, device=dev) for _ in range(ns)]
    res = [None] * ns

    def run(n):
        for i in range(n):
            k = i % ns
            with torch.cuda.stream(streams[k]):
                res[k] = fitter.fit(tv, tj, _workspace=wss[k], **kw)

    torch.cuda.synchronize()
    run(2 * ns + 2)
    torch.cuda.synchronize()
    rates = []
    for rep in range(3):
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        rates.append(B * steps / (time.perf_counter() - t0))
    chk = [float(r['pose_rotvecs'].double().abs().sum().item()) for r in res]
    if ref is None:
        ref = chk[0]
    out[f'streams_{ns}'] = dict(fits_per_s=[round(x) for x in rates], same_result=all(c == ref for c in chk))
    del wss
    torch.cuda.empty_cache()
print(json.dumps(dict(kind=kind, B=B, steps=steps, **out)))
