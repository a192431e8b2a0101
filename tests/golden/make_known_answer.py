"""Extract the DATA of the reference's only known-answer test (tests/test_forward.py:7-127: one pose,
one shape, expected `vertices[:, ::300]` and joints of the REAL SMPL neutral model) into a fixture.
Only the four array literals are evaluated; nothing else of the reference's test file is kept.

Usage (build container):  python tests/golden/make_known_answer.py
The fixture drives tests/test_real_model.py, which runs only when a real SMPL model file is supplied
(SMPLFITTER_BODY_MODELS / DATA_ROOT), as SURVEY.md §8c asks."""
import ast
import os.path as osp

import numpy as np

HERE = osp.dirname(osp.abspath(__file__))
src = open('/root/reference/tests/test_forward.py').read()
fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'test_smpl')
want = {'rots': 'pose_rotvecs', 'betas': 'shape_betas', 'verts_expect': 'vertices_every_300th',
        'joints_expect': 'joints'}
out = {}
for node in fn.body:
    if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id in want:
        value = eval(compile(ast.Expression(node.value), '<literal>', 'eval'), {'np': np})
        out[want[node.targets[0].id]] = np.asarray(value, np.float32)
assert set(out) == set(want.values()), sorted(out)
path = osp.join(HERE, 'reference_known_answer_smpl.npz')
np.savez_compressed(path, **out)
print(path, {k: v.shape for k, v in out.items()})
