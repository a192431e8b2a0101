"""Build + ctypes-load the TEST-ONLY host emulation (tests/hostemu/hostemu.cpp)."""

import ctypes as C
import os
import os.path as osp
import subprocess

import numpy as np

from smplfitter_amd import _lib

HERE = osp.dirname(osp.abspath(__file__))
SRC = osp.join(HERE, 'hostemu', 'hostemu.cpp')
CSRC = osp.join(HERE, '..', 'smplfitter_amd', 'csrc')
BUILD = osp.join(HERE, 'hostemu', '_build')
SO = osp.join(BUILD, 'libhostemu.so')

_cache = None


def load():
    global _cache
    if _cache is not None:
        return _cache
    deps = [SRC] + [osp.join(CSRC, f) for f in ('sf_math.h', 'sf_stages.h', 'sf_tables.h', 'sf_tables.cpp')]
    # HOSTEMU_SANITIZE=1: the same sources (the product's table builder and stage code) under
    # AddressSanitizer + UBSan; run as  LD_PRELOAD=$(gcc -print-file-name=libasan.so) HOSTEMU_SANITIZE=1
    # ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_hostemu.py   (tools/asan_hostemu.sh)
    san = os.getenv('HOSTEMU_SANITIZE') == '1'
    so = SO.replace('.so', '_asan.so') if san else SO
    if not osp.exists(so) or any(osp.getmtime(d) > osp.getmtime(so) for d in deps):
        os.makedirs(BUILD, exist_ok=True)
        tmp = so + f'.tmp{os.getpid()}'
        flags = ['-O1', '-g', '-fsanitize=address,undefined', '-fno-omit-frame-pointer',
                 '-fno-sanitize-recover=undefined'] if san else ['-O2']
        subprocess.run(
            ['g++', *flags, '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', SRC,
             osp.join(CSRC, 'sf_tables.cpp'), '-o', tmp],
            check=True,
        )
        os.replace(tmp, so)
    lib = C.CDLL(so)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    lib.hostemu_fit.argtypes = [C.POINTER(_lib.ModelDesc), vp, vp, vp, vp, i32, i32, f32, f32, f32, i32, vp, vp, vp, vp, vp, vp]
    lib.hostemu_fit.restype = i32
    lib.hostemu_forward.argtypes = [C.POINTER(_lib.ModelDesc), vp, vp, vp, i32, vp, vp, i32, vp, vp, vp]
    lib.hostemu_forward.restype = i32
    lib.hostemu_fit_warm.argtypes = [C.POINTER(_lib.ModelDesc), vp, vp, vp, vp, i32, i32, f32, f32, f32, i32, vp, vp, i32, vp, i32, i32, f32, vp, vp, vp, vp, vp, vp]
    lib.hostemu_fit_warm.restype = i32
    lib.hostemu_set_share_allreduce.argtypes = [_lib.ShareAllreduceFn, vp]
    lib.hostemu_set_share_allreduce.restype = None
    lib.hostemu_fit_known_shape.argtypes = [C.POINTER(_lib.ModelDesc), vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.hostemu_fit_known_shape.restype = i32
    lib.hostemu_last_error.restype = C.c_char_p
    for fn in ('hostemu_proj_so3', 'hostemu_mat2rotvec', 'hostemu_rotvec2mat'):
        getattr(lib, fn).argtypes = [vp, vp, i32]
    lib.hostemu_align.argtypes = [vp, vp, vp, i32]
    _cache = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def desc_from_md(md, kind='smpl', enable_kid=False):
    return _lib.make_desc(
        md.v_template, md.shapedirs, md.posedirs, md.weights, md.J_template, md.J_shapedirs,
        md.kintree_parents,
        md.J_regressor_post_lbs if md.J_regressor_post_lbs.shape[1] == md.num_vertices else None,
        is_smpl_family=kind.startswith('smpl'),
        kid_shapedir=md.kid_shapedir if enable_kid else None,
        kid_J_shapedir=md.kid_J_shapedir if enable_kid else None,
    )


def fit(md, kind, tv, tj=None, vw=None, jw=None, num_iter=1, beta_regularizer=1.0,
        beta_regularizer2=0.0, final_adjust_rots=True, enable_kid=False, kid_regularizer=None):
    lib = load()
    desc, keep = desc_from_md(md, kind, enable_kid)
    if kid_regularizer is None:
        kid_regularizer = beta_regularizer
    f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)  # noqa: E731
    tv, tj, vw, jw = f(tv), f(tj), f(vw), f(jw)
    B, J, S = tv.shape[0], md.num_joints, md.shapedirs.shape[2]
    pose = np.zeros((B, 3 * J), np.float32)
    betas = np.zeros((B, S), np.float32)
    trans = np.zeros((B, 3), np.float32)
    orient = np.zeros((B, J, 3, 3), np.float32)
    G0 = np.zeros((B, J, 3, 3), np.float32)
    kid = np.zeros((B,), np.float32)
    rc = lib.hostemu_fit(C.byref(desc), _p(tv), _p(tj), _p(vw), _p(jw), B, num_iter, beta_regularizer,
                         beta_regularizer2, kid_regularizer, int(final_adjust_rots), _p(pose), _p(betas),
                         _p(trans), _p(kid), _p(orient), _p(G0))
    if rc != 0:
        raise RuntimeError(lib.hostemu_last_error().decode())
    out = dict(pose_rotvecs=pose, shape_betas=betas, trans=trans, orientations=orient, glob_rotmats_iter0=G0)
    if enable_kid:
        out['kid_factor'] = kid
    return out


def forward(md, kind, pose=None, betas=None, trans=None, glob=None, kid=None):
    lib = load()
    desc, keep = desc_from_md(md, kind, kid is not None)
    f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)  # noqa: E731
    pose, betas, trans, glob, kid = f(pose), f(betas), f(trans), f(glob), f(kid)
    B = (pose if pose is not None else glob).shape[0]
    J, V = md.num_joints, md.num_vertices
    verts = np.zeros((B, V, 3), np.float32)
    joints = np.zeros((B, J, 3), np.float32)
    orient = np.zeros((B, J, 3, 3), np.float32)
    rc = lib.hostemu_forward(C.byref(desc), _p(pose), _p(glob), _p(betas), 0 if betas is None else betas.shape[1],
                             _p(trans), _p(kid), B, _p(verts), _p(joints), _p(orient))
    if rc != 0:
        raise RuntimeError(lib.hostemu_last_error().decode())
    return dict(vertices=verts, joints=joints, orientations=orient)


def fit_known_shape(md, kind, betas, tv, target_joints=None, vertex_weights=None, joint_weights=None,
                    kid_factor=None, num_iter=1, final_adjust_rots=True, initial_pose_rotvecs=None,
                    scale_fit=False):
    lib = load()
    desc, keep = desc_from_md(md, kind, kid_factor is not None)
    f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)  # noqa: E731
    betas, tv, tj, vw, jw = f(betas), f(tv), f(target_joints), f(vertex_weights), f(joint_weights)
    kid, init = f(kid_factor), f(initial_pose_rotvecs)
    B, J = tv.shape[0], md.num_joints
    pose = np.zeros((B, 3 * J), np.float32)
    trans = np.zeros((B, 3), np.float32)
    scale = np.ones((B,), np.float32)
    orient = np.zeros((B, J, 3, 3), np.float32)
    rc = lib.hostemu_fit_known_shape(C.byref(desc), _p(betas), betas.shape[1], _p(kid), _p(init), _p(tv),
                                     _p(tj), _p(vw), _p(jw), B, num_iter, int(final_adjust_rots),
                                     int(scale_fit), _p(pose), _p(trans), _p(scale), _p(orient))
    if rc != 0:
        raise RuntimeError(lib.hostemu_last_error().decode())
    out = dict(pose_rotvecs=pose, trans=trans, orientations=orient)
    if scale_fit:
        out['scale_corr'] = scale
    return out


def fit_warm(md, kind, tv, target_joints=None, vertex_weights=None, joint_weights=None, num_iter=1,
             beta_regularizer=1.0, beta_regularizer2=0.0, final_adjust_rots=True, enable_kid=False,
             kid_regularizer=None, initial_pose_rotvecs=None, initial_shape_betas=None,
             initial_kid_factor=None, share_beta=False, scale_target=False, scale_fit=False,
             scale_regularizer=0.0, share_allreduce=None):
    """``share_allreduce(sums: float64 ndarray)`` sums the array in place over the ranks of a sharded
    ``share_beta`` fit (the host-memory counterpart of ``smplfit_fit_args.share_allreduce``)."""
    lib = load()
    desc, keep = desc_from_md(md, kind, enable_kid)
    if kid_regularizer is None:
        kid_regularizer = beta_regularizer

    def _cb(_user, ptr, count, _stream):
        share_allreduce(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(count,)))
        return 0

    cb = _lib.ShareAllreduceFn(_cb) if share_allreduce is not None else _lib.ShareAllreduceFn()
    lib.hostemu_set_share_allreduce(cb, None)
    f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)  # noqa: E731
    tv, tj, vw, jw = f(tv), f(target_joints), f(vertex_weights), f(joint_weights)
    ip, ib, ik = f(initial_pose_rotvecs), f(initial_shape_betas), f(initial_kid_factor)
    B, J, S = tv.shape[0], md.num_joints, md.shapedirs.shape[2]
    pose = np.zeros((B, 3 * J), np.float32)
    betas = np.zeros((B, S), np.float32)
    trans = np.zeros((B, 3), np.float32)
    orient = np.zeros((B, J, 3, 3), np.float32)
    kid = np.zeros((B,), np.float32)
    scale = np.ones((B,), np.float32)
    scale_mode = 1 if scale_target else 2 if scale_fit else 0
    rc = lib.hostemu_fit_warm(C.byref(desc), _p(tv), _p(tj), _p(vw), _p(jw), B, num_iter, beta_regularizer,
                              beta_regularizer2, kid_regularizer, int(final_adjust_rots), _p(ip), _p(ib),
                              0 if ib is None else ib.shape[1], _p(ik), int(share_beta), scale_mode, scale_regularizer,
                              _p(scale), _p(pose), _p(betas), _p(trans),
                              _p(kid), _p(orient))
    lib.hostemu_set_share_allreduce(_lib.ShareAllreduceFn(), None)
    if rc != 0:
        raise RuntimeError(lib.hostemu_last_error().decode())
    out = dict(pose_rotvecs=pose, shape_betas=betas, trans=trans, orientations=orient)
    if enable_kid:
        out['kid_factor'] = kid
    if scale_mode:
        out['scale_corr'] = scale
    return out
