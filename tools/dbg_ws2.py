"""Which workspace regions differ between identical fits when a noisy neighbour perturbs the timing?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dbg_ws.py')).read().split("nbytes = h.workspace_bytes(B)")[0])
nbytes = h.workspace_bytes(B)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
NI = int(os.environ.get('NI', 1))
kw = dict(num_iter=NI, beta_regularizer=1.0, final_adjust_rots=False, _workspace=ws)
chunks = 2 if os.environ.get('SMPLFIT_CHUNKS', '2') != '1' else 1
regs, per = regions(B // chunks)
noise_a = torch.empty(48 << 20, dtype=torch.float32, device=dev); noise_b = torch.empty_like(noise_a)
side = torch.cuda.Stream(device=dev)
def run(mode):
    if mode == 1:
        with torch.cuda.stream(side):
            for _ in range(6): noise_b.copy_(noise_a)
    if mode == 2:
        ws[: 1 << 20] = 0   # a fill kernel in front, like the poison test
    r = f.fit(tv, tj, **kw); torch.cuda.synchronize()
    return ws.clone(), r['pose_rotvecs'].clone()
base = run(0)
for rep in range(1, 10):
    s = run(rep % 3)
    d = s[0] != base[0]
    names = []
    for c in range(chunks):
        for n, off, sz in regs:
            k = int(d[c*per + off: c*per + off + sz].sum())
            if k: names.append(f'c{c}.{n}:{k}')
    print(f'rep {rep} mode {rep % 3}: pose differs {int((s[1] != base[1]).any(1).sum())};', ' '.join(names) or 'identical', flush=True)
