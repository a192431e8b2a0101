"""Debug: options of fit on the general path against golden_general_opts.npz (prints the distances)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import util
from smplfitter_amd import synth
from smplfitter_amd.pt import BodyFitter, BodyModel
dev = torch.device('cuda:0')
gg = dict(np.load('tests/golden/golden_general.npz')); go = dict(np.load('tests/golden/golden_general_opts.npz'))
root = synth.ensure_model_root(kinds=tuple(util.GENERAL_OPT_KINDS), seed=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for kind, cases in util.GENERAL_OPT_KINDS.items():
    g, ge = util.general_view(gg, kind), util.general_view(go, kind)
    m = BodyModel('smpl', 'neutral', model_root=f'{root}/{kind}', num_betas=util.GENERAL_KINDS[kind], device=dev)
    om = util.general_oracle(root, kind, np.float64); om32 = util.general_oracle(root, kind)
    fit = {False: BodyFitter(m), True: BodyFitter(m, enable_kid=True)}
    ofit = {False: util.O.OracleFitter(om), True: util.O.OracleFitter(om, enable_kid=True)}
    tt = lambda kw: {k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
    for grp in ('scale', 'share', 'sharescale'):
        for case in cases[grp]:
            if grp == 'scale':
                kid_fit, tv, kw = util.scale_inputs(g, case); ex = {}
            elif grp == 'share':
                kid_fit, tv, kw = util.share_inputs(g, om32, case); ex = dict(share_beta=True)
            else:
                kid_fit, tv, kw = util.share_scale_inputs(g, om32, case); ex = dict(share_beta=True)
            keys = ['pose_rotvecs', 'shape_betas', 'trans'] + (['scale_corr'] if grp != 'share' else [])
            try:
                o = {k: v.cpu().numpy() for k, v in fit[kid_fit].fit(t(tv), requested_keys=keys, **ex, **tt(kw)).items()}
            except Exception as e:
                print(kind, grp, case, 'ERROR', type(e).__name__, str(e)[:100]); continue
            r = ofit[kid_fit].fit(tv, **ex, **kw)
            line = f'{kind} {grp} {case}:'
            for k in keys[1:]:
                line += f' {k} vs ref {np.abs(o[k] - ge[f"{grp}.{case}.{k}"]).max():.2e} vs oracle64 {np.abs(o[k] - r[k]).max():.2e};'
            print(line, flush=True)
