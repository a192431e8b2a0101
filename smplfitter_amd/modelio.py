"""Model-file loader: reads the official SMPL-family files into float arrays.

Host-side glue (not on the hot path).  It accepts the same files, the same search order and produces
the same derived arrays as the reference loader (reference: src/smplfitter/common.py:219-395,
``initialize``), so licensed SMPL / SMPL-X files and the synthetic fixture drop in unchanged:

* search order ``model_root`` arg, ``$SMPLFITTER_BODY_MODELS/<name>``, ``$DATA_ROOT/body_models/<name>``,
  ``./body_models/<name>`` (common.py:228-240; the platformdirs fallback is replaced by an error —
  there is no downloader here);
* ``J_template = J_regressor @ v_template``, ``J_shapedirs = J_regressor . shapedirs`` unless the file
  carries them (common.py:336-344);
* the kid blend shape from ``kid_template.npy`` (common.py:317-334);
* the identity-pose pose-corrective is folded into ``v_template`` so that
  ``v_posed = v_template + posedirs . vec(R_rel)`` with unshifted rotation matrices (common.py:346-350);
* ``vertex_subset`` slices every per-vertex array (common.py:368-393).

Official ``.pkl`` files hold chumpy objects; they are unpickled through a restricted ``Unpickler``
whose ``find_class`` maps any ``chumpy.*`` class to a tiny array-like stand-in (the reference instead
registers stub modules in ``sys.modules``, common.py:432-475).
"""

from __future__ import annotations

import os
import os.path as osp
import pickle
from dataclasses import dataclass
from typing import Optional

import numpy as np

_FILE_PATTERNS = {
    # model name -> (gender letter -> token, filename template)
    'smpl': (dict(f='f', m='m', n='neutral'), 'basicmodel_{g}_lbs_10_207_0_v1.1.0.pkl'),
    'smplx': (dict(f='FEMALE', m='MALE', n='NEUTRAL'), 'SMPLX_{g}.npz'),
    'smplxlh': (dict(f='FEMALE', m='MALE', n='NEUTRAL'), 'SMPLX_{g}.npz'),
    'smplxmoyo': (dict(f='FEMALE', m='MALE', n='NEUTRAL'), 'SMPLX_{g}.npz'),
    'smplh': (dict(f='female', m='male'), 'SMPLH_{g}.pkl'),
    'smplh16': (dict(f='female', m='male', n='neutral'), '{g}/model.npz'),
    'mano': ({}, 'MANO_RIGHT.pkl'),
}


@dataclass
class ModelData:
    """Arrays of one body model (float64 / int as loaded; BodyModel casts to float32)."""

    v_template: np.ndarray  # (V, 3)
    shapedirs: np.ndarray  # (V, 3, S)
    posedirs: np.ndarray  # (V, 3, 9(J-1))
    J_regressor_post_lbs: np.ndarray  # (J, V')
    J_template: np.ndarray  # (J, 3)
    J_shapedirs: np.ndarray  # (J, 3, S)
    kid_shapedir: np.ndarray  # (V, 3)
    kid_J_shapedir: np.ndarray  # (J, 3)
    weights: np.ndarray  # (V, J)
    kintree_parents: list
    faces: np.ndarray
    num_joints: int
    num_vertices: int
    vertex_subset: np.ndarray
    joint_names: list


class _ChumpyStandIn:
    """Array-like stand-in for pickled ``chumpy`` nodes (``Ch`` stores ``.x``; ``Select`` stores
    ``.a``/``.idxs``).  Only ``__array__`` is needed."""

    def __setstate__(self, state):
        self.__dict__.update(state)

    def __array__(self, dtype=None, copy=None):
        if hasattr(self, 'idxs'):
            out = np.asarray(self.a, dtype=dtype).ravel()[self.idxs]
            shp = getattr(self, 'preferred_shape', None)
            return out.reshape(shp) if shp is not None else out
        return np.asarray(self.x, dtype=dtype)


class _ModelUnpickler(pickle.Unpickler):
    """Restricted unpickler for body-model / deformation-transfer files: numpy array reconstruction, scipy
    sparse matrices, chumpy stand-ins and plain builtins are allowed; every other global raises
    ``UnpicklingError`` (a crafted model file cannot import arbitrary callables)."""

        # protocol-2 pickles of numpy arrays decode their byte payload with ``_codecs.encode(str, 'latin1')``:
    # a pure string -> bytes function, safe to allow.
    _ALLOWED = {
        ('_codecs', 'encode'),
        ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
        ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
        ('numpy', 'ndarray'), ('numpy', 'dtype'), ('numpy', 'matrix'),
        ('numpy.core.numeric', '_frombuffer'), ('numpy._core.numeric', '_frombuffer'),
        ('collections', 'OrderedDict'), ('builtins', 'dict'), ('builtins', 'list'), ('builtins', 'tuple'),
        ('builtins', 'set'), ('builtins', 'frozenset'), ('builtins', 'slice'), ('builtins', 'complex'),
        ('builtins', 'bytearray'), ('__builtin__', 'dict'), ('__builtin__', 'list'), ('__builtin__', 'tuple'),
        ('__builtin__', 'set'), ('__builtin__', 'object'), ('builtins', 'object'), ('copy_reg', '_reconstructor'),
        ('copyreg', '_reconstructor'),
    }

    def find_class(self, module, name):
        if module.split('.')[0] == 'chumpy':
            return _ChumpyStandIn
        if module == 'scipy.sparse' or module.startswith('scipy.sparse.'):  # incl. old scipy.sparse.csc.csc_matrix
            import scipy.sparse

            if name.endswith(('_matrix', '_array')) and hasattr(scipy.sparse, name):
                return getattr(scipy.sparse, name)
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f'global {module}.{name} is not allowed in a body-model file')


def restricted_load(fileobj, **kw):
    """``pickle.load`` through ``_ModelUnpickler`` (latin1 for the python-2 pickles SMPL ships as)."""
    return _ModelUnpickler(fileobj, **kw).load()


def _resolve_root(model_name, model_root):
    if model_root is not None:
        return model_root
    base = os.getenv('SMPLFITTER_BODY_MODELS')
    if base is None:
        data_root = os.getenv('DATA_ROOT')
        if data_root is not None:
            base = f'{data_root}/body_models'
        elif osp.isdir('body_models'):
            base = 'body_models'
        else:
            raise FileNotFoundError(
                'No body-model directory configured. Pass model_root=..., or set '
                'SMPLFITTER_BODY_MODELS or DATA_ROOT (same conventions as smplfitter).'
            )
    return f'{base}/{model_name}'


def _joint_names(model_name, J):
    smpl = [
        'pelvis', 'left_hip', 'right_hip', 'spine1', 'left_knee', 'right_knee', 'spine2',
        'left_ankle', 'right_ankle', 'spine3', 'left_foot', 'right_foot', 'neck', 'left_collar',
        'right_collar', 'head', 'left_shoulder', 'right_shoulder', 'left_elbow', 'right_elbow',
        'left_wrist', 'right_wrist', 'left_hand', 'right_hand',
    ]  # fmt: skip
    fingers = [f'{f}{k}' for f in ('index', 'middle', 'pinky', 'ring', 'thumb') for k in (1, 2, 3)]
    hands = [f'left_{n}' for n in fingers] + [f'right_{n}' for n in fingers]
    if model_name == 'smpl':
        return smpl
    if model_name in ('smplh', 'smplh16'):
        return smpl[:22] + hands
    if model_name.startswith('smplx'):
        return smpl[:22] + ['jaw', 'left_eye_smplhf', 'right_eye_smplhf'] + hands
    if model_name == 'mano':
        return ['wrist'] + fingers
    return [f'joint{i}' for i in range(J)]


def load_model(
    model_name: str = 'smpl',
    gender: str = 'neutral',
    model_root: Optional[str] = None,
    num_betas: Optional[int] = None,
    vertex_subset_size: Optional[int] = None,
    vertex_subset=None,
    faces=None,
    joint_regressor_post_lbs=None,
) -> ModelData:
    """Load one SMPL-family model file (same arguments as the reference's ``initialize``)."""
    if model_name not in _FILE_PATTERNS:
        raise ValueError(f'Unknown model name: {model_name}')
    gmap, template = _FILE_PATTERNS[model_name]
    if model_name != 'mano':
        key = gender[0].lower()
        if key not in gmap:
            names = dict(f='female', m='male', n='neutral')
            available = ', '.join(repr(names[k]) for k in gmap)
            raise ValueError(
                f"Gender '{gender}' is not available for model '{model_name}'. "
                f'Available: {available}.'
            )
        filename = template.format(g=gmap[key])
    else:
        filename = template
    root = _resolve_root(model_name, model_root)
    path = osp.join(root, filename)
    if not osp.exists(path):
        raise FileNotFoundError(
            f'Body model file not found: {path}\n'
            f"Set the location with BodyModel(..., model_root='/path/body_models/{model_name}'), "
            f'SMPLFITTER_BODY_MODELS or DATA_ROOT.'
        )
    if path.endswith('.npz'):
        raw = np.load(path)
    else:
        with open(path, 'rb') as f:
            raw = _ModelUnpickler(f, encoding='latin1').load()

    f64 = np.float64
    shapedirs = np.array(raw['shapedirs'], dtype=f64)
    posedirs = np.array(raw['posedirs'], dtype=f64)
    v_template = np.array(raw['v_template'], dtype=f64)
    jr = raw['J_regressor']
    J_regressor = (np.asarray(jr.toarray()) if hasattr(jr, 'toarray') else np.asarray(jr)).astype(f64)
    weights = np.array(raw['weights'])
    file_faces = np.array(np.asarray(raw['f']).astype(np.int32))
    parents = np.array(np.asarray(raw['kintree_table'])[0]).astype(np.int64)
    parents = parents.astype(np.int32).tolist()  # 2**32-1 in the official files wraps to -1
    J = len(parents)
    V = len(v_template)

    if model_name.lower().startswith('smpl'):
        kid_path = osp.join(root, 'kid_template.npy')
        if not osp.exists(kid_path):
            raise FileNotFoundError(f'Kid template not found: {kid_path}')
        smil = np.load(kid_path).astype(f64)
        kid_shapedir = smil - smil.mean(axis=0) - v_template
        kid_J_shapedir = J_regressor @ kid_shapedir
    else:
        kid_shapedir = np.zeros_like(v_template)
        kid_J_shapedir = np.zeros((J, 3))

    keys = raw.files if hasattr(raw, 'files') else raw.keys()
    if 'J_shapedirs' in keys:
        J_shapedirs = np.array(raw['J_shapedirs'], dtype=f64)
    else:
        J_shapedirs = np.einsum('jv,vcs->jcs', J_regressor, shapedirs)
    if 'J_template' in keys:
        J_template = np.array(raw['J_template'], dtype=f64)
    else:
        J_template = J_regressor @ v_template

    # fold the rest-pose (identity rotations) pose-corrective into the template
    eye_feature = np.tile(np.eye(3, dtype=f64), [J - 1, 1]).reshape(-1)
    v_template = v_template - posedirs @ eye_feature

    if vertex_subset_size is not None:
        subset_path = f'{root}/vertex_subset_{vertex_subset_size}.npz'
        if not osp.exists(subset_path):
            raise FileNotFoundError(
                f'{subset_path} not found (mesh decimation is an offline tool of the reference, '
                f'src/smplfitter/decimation/; pass vertex_subset=<indices> instead).'
            )
        sub = np.load(subset_path)
        vertex_subset = sub['i_verts']
        faces = sub['faces']
        reg_path = f'{root}/vertex_subset_joint_regr_post_lbs_{vertex_subset_size}.npy'
        if osp.exists(reg_path):
            joint_regressor_post_lbs = np.load(reg_path)
        else:
            joint_regressor_post_lbs = J_regressor[:, vertex_subset]

    if vertex_subset is None:
        vertex_subset = np.arange(V, dtype=np.int64)
    else:
        vertex_subset = np.array(vertex_subset, dtype=np.int64)
    if faces is None:
        faces = file_faces
    if joint_regressor_post_lbs is None:
        joint_regressor_post_lbs = J_regressor

    return ModelData(
        v_template=v_template[vertex_subset],
        shapedirs=shapedirs[vertex_subset, :, :num_betas],
        posedirs=posedirs[vertex_subset],
        J_regressor_post_lbs=np.asarray(joint_regressor_post_lbs),
        J_template=J_template,
        J_shapedirs=J_shapedirs[:, :, :num_betas],
        kid_shapedir=kid_shapedir[vertex_subset],
        kid_J_shapedir=kid_J_shapedir,
        weights=weights[vertex_subset],
        kintree_parents=parents,
        faces=faces,
        num_joints=J,
        num_vertices=len(vertex_subset),
        vertex_subset=vertex_subset,
        joint_names=_joint_names(model_name, J),
    )
