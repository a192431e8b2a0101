"""ctypes binding of ``libsmplfit_hip.so`` (the C-ABI of ``include/smplfit.h``).

The shared library is built in-tree by ``smplfitter_amd.build`` (``hipcc --offload-arch=gfx950``).
There is NO fallback: if the library is missing or a call fails, an exception is raised — the product
path never routes through a CPU implementation.
"""

from __future__ import annotations

import ctypes as C
import os
import os.path as osp

import numpy as np

_HERE = osp.dirname(osp.abspath(__file__))
# SMPLFIT_LIB: another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.getenv('SMPLFIT_LIB') or osp.join(_HERE, 'libsmplfit_hip.so')

SMPLFIT_OK = 0
SMPLFIT_ERR_BAD_ARG = -1
SMPLFIT_ERR_UNSUPPORTED = -2
SMPLFIT_PATH_WAVE, SMPLFIT_PATH_BATCH_MAJOR, SMPLFIT_PATH_GENERAL = 0, 1, 2  # smplfit_info.vertex_path
SMPLFIT_ERR_WORKSPACE = -3
SMPLFIT_ERR_HIP = -4
SMPLFIT_CREATE_HOST_ONLY = 1
SMPLFIT_ABI_VERSION = 5  # include/smplfit.h; checked against smplfit_abi_version() when the library is loaded

TABLE_IDS = dict(
    part_assignment=0, sort_perm=1, part_type=2, fk_order=3, fk_level_start=4, adj_flag=5,
    used_part=6, segments=7, vertex_pieces=8, cell_counts=9, joint_pairs=10,
)

# every symbol include/smplfit.h declares
EXPORTED_SYMBOLS = [
    'smplfit_create', 'smplfit_destroy', 'smplfit_last_error', 'smplfit_version',
    'smplfit_get_info', 'smplfit_get_table', 'smplfit_workspace_bytes', 'smplfit_fit_f32',
    'smplfit_forward_f32', 'smplfit_part_rotations_f32', 'smplfit_shape_solve_f32',
    'smplfit_shape_solve_ex_f32', 'smplfit_fit_known_shape_f32', 'smplfit_fit_warm_f32', 'smplfit_fit_ex_f32',
    'smplfit_time_kernel_f32', 'smplfit_primitives_f32', 'smplfit_forward_ex_f32',
    'smplfit_transfer_create', 'smplfit_transfer_destroy', 'smplfit_transfer_f32',
    'smplfit_convert_plan_create', 'smplfit_convert_plan_destroy', 'smplfit_convert_workspace_bytes',
    'smplfit_convert_f32', 'smplfit_reload_options', 'smplfit_get_share_table', 'smplfit_pick_share_mult',
    'smplfit_abi_version',
]  # fmt: skip

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


# smplfit_share_allreduce_fn: int (*)(void* user, double* sums, int32_t count, void* hip_stream)
ShareAllreduceFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)


class FitArgs(C.Structure):
    """smplfit_fit_args (include/smplfit.h)."""
    _fields_ = [
        ('target_vertices', C.c_void_p), ('target_joints', C.c_void_p), ('vertex_weights', C.c_void_p),
        ('joint_weights', C.c_void_p), ('batch', C.c_int32), ('num_iter', C.c_int32),
        ('beta_regularizer', C.c_float), ('beta_regularizer2', C.c_float), ('kid_regularizer', C.c_float),
        ('final_adjust_rots', C.c_int32), ('initial_pose_rotvecs', C.c_void_p),
        ('initial_shape_betas', C.c_void_p), ('num_initial_betas', C.c_int32),
        ('initial_kid_factor', C.c_void_p), ('share_beta', C.c_int32), ('scale_mode', C.c_int32),
        ('scale_regularizer', C.c_float), ('pose_rotvecs', C.c_void_p),
        ('shape_betas', C.c_void_p), ('trans', C.c_void_p), ('kid_factor', C.c_void_p),
        ('orientations', C.c_void_p), ('relative_orientations', C.c_void_p), ('scale_corr', C.c_void_p),
        ('workspace', C.c_void_p),
        ('workspace_bytes', C.c_size_t), ('hip_stream', C.c_void_p),
        ('share_allreduce', ShareAllreduceFn), ('share_user', C.c_void_p),
    ]


class ShapeSolveArgs(C.Structure):
    """smplfit_shape_solve_args (include/smplfit.h)."""
    _fields_ = [
        ('glob_rotmats', C.c_void_p), ('target_vertices', C.c_void_p), ('target_joints', C.c_void_p),
        ('vertex_weights', C.c_void_p), ('joint_weights', C.c_void_p), ('batch', C.c_int32),
        ('beta_regularizer', C.c_float), ('beta_regularizer2', C.c_float), ('kid_regularizer', C.c_float),
        ('add_mean', C.c_int32), ('beta_regularizer_reference', C.c_void_p),
        ('num_reference_betas', C.c_int32), ('kid_regularizer_reference', C.c_void_p),
        ('share_beta', C.c_int32), ('scale_mode', C.c_int32), ('scale_regularizer', C.c_float),
        ('shape_betas', C.c_void_p), ('trans', C.c_void_p), ('kid_factor', C.c_void_p),
        ('scale_corr', C.c_void_p), ('vertices_out', C.c_void_p), ('joints_out', C.c_void_p),
        ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('hip_stream', C.c_void_p),
        ('share_allreduce', ShareAllreduceFn), ('share_user', C.c_void_p),
    ]


class ForwardArgs(C.Structure):
    """smplfit_forward_args (include/smplfit.h)."""
    _fields_ = [
        ('pose_rotvecs', C.c_void_p), ('glob_rotmats', C.c_void_p), ('rel_rotmats', C.c_void_p),
        ('shape_betas', C.c_void_p), ('num_betas_given', C.c_int32), ('trans', C.c_void_p),
        ('kid_factor', C.c_void_p), ('batch', C.c_int32), ('vertices', C.c_void_p), ('joints', C.c_void_p),
        ('orientations', C.c_void_p), ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t),
        ('hip_stream', C.c_void_p),
    ]


class ConvertArgs(C.Structure):
    """smplfit_convert_args (include/smplfit.h)."""
    _fields_ = [
        ('pose_rotvecs', C.c_void_p), ('shape_betas', C.c_void_p), ('num_betas_given', C.c_int32),
        ('trans', C.c_void_p), ('batch', C.c_int32), ('num_iter', C.c_int32),
        ('beta_regularizer', C.c_float), ('beta_regularizer2', C.c_float), ('kid_regularizer', C.c_float),
        ('final_adjust_rots', C.c_int32), ('out_pose_rotvecs', C.c_void_p), ('out_shape_betas', C.c_void_p),
        ('out_trans', C.c_void_p), ('out_kid_factor', C.c_void_p), ('out_orientations', C.c_void_p),
        ('out_relative_orientations', C.c_void_p), ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t),
        ('hip_stream', C.c_void_p),
    ]


class ModelDesc(C.Structure):
    _fields_ = [
        ('num_vertices', C.c_int32),
        ('num_joints', C.c_int32),
        ('num_betas', C.c_int32),
        ('is_smpl_family', C.c_int32),
        ('v_template', _fp),
        ('shapedirs', _fp),
        ('posedirs', _fp),
        ('weights', _fp),
        ('J_template', _fp),
        ('J_shapedirs', _fp),
        ('parents', _ip),
        ('J_regressor_post_lbs', _fp),
        ('regressor_num_vertices', C.c_int32),
        ('enable_kid', C.c_int32),
        ('kid_shapedir', _fp),
        ('kid_J_shapedir', _fp),
    ]


class Info(C.Structure):
    _fields_ = [
        (n, C.c_int32)
        for n in (
            'num_vertices', 'num_joints', 'num_betas', 'has_kid', 'padded_vertices', 'num_used_vertices',
            'skin_width', 'num_segments', 'num_fk_levels', 'adj_last_level', 'has_device', 'gemm_vgprs',
            'vertex_path', 'share_fallback',
        )
    ]  # fmt: skip


class SmplfitError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not osp.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: build it with `python -m smplfitter_amd.build` '
            '(hipcc, gfx950).  smplfitter_amd has no CPU fallback.'
        )
    # PyTorch is the container for device memory and streams, so the kernels must run on the SAME
    # HIP runtime instance torch uses: torch bundles a libamdhip64.so.7 with the same SONAME as the
    # system one, and whichever is loaded first serves both.  Import torch first.
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    vp, sz, i32, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    lib.smplfit_create.argtypes = [C.POINTER(ModelDesc), i32, C.POINTER(vp)]
    lib.smplfit_create.restype = i32
    lib.smplfit_destroy.argtypes = [vp]
    lib.smplfit_destroy.restype = None
    lib.smplfit_last_error.restype = C.c_char_p
    lib.smplfit_version.restype = C.c_char_p
    lib.smplfit_get_info.argtypes = [vp, C.POINTER(Info)]
    lib.smplfit_get_info.restype = i32
    lib.smplfit_get_table.argtypes = [vp, i32, _ip, sz, C.POINTER(sz)]
    lib.smplfit_get_table.restype = i32
    lib.smplfit_workspace_bytes.argtypes = [vp, i32]
    lib.smplfit_workspace_bytes.restype = sz
    lib.smplfit_fit_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, f32, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.smplfit_fit_f32.restype = i32
    lib.smplfit_fit_warm_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, f32, f32, f32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.smplfit_fit_warm_f32.restype = i32
    lib.smplfit_fit_ex_f32.argtypes = [vp, C.POINTER(FitArgs)]
    lib.smplfit_fit_ex_f32.restype = i32
    lib.smplfit_forward_f32.argtypes = [vp, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, sz, vp]
    lib.smplfit_forward_f32.restype = i32
    lib.smplfit_part_rotations_f32.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, sz, vp]
    lib.smplfit_part_rotations_f32.restype = i32
    lib.smplfit_shape_solve_f32.argtypes = [vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, i32, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.smplfit_shape_solve_f32.restype = i32
    lib.smplfit_shape_solve_ex_f32.argtypes = [vp, C.POINTER(ShapeSolveArgs)]
    lib.smplfit_shape_solve_ex_f32.restype = i32
    lib.smplfit_fit_known_shape_f32.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.smplfit_fit_known_shape_f32.restype = i32
    lib.smplfit_time_kernel_f32.argtypes = [vp, i32, i32, i32, vp, sz, vp, C.POINTER(C.c_float)]
    lib.smplfit_time_kernel_f32.restype = i32
    lib.smplfit_primitives_f32.argtypes = [i32, vp, vp, vp, i32, vp]
    lib.smplfit_primitives_f32.restype = i32
    lib.smplfit_forward_ex_f32.argtypes = [vp, C.POINTER(ForwardArgs)]
    lib.smplfit_forward_ex_f32.restype = i32
    lib.smplfit_transfer_create.argtypes = [i32, i32, _ip, _ip, _fp, i32, C.POINTER(vp)]
    lib.smplfit_transfer_create.restype = i32
    lib.smplfit_transfer_destroy.argtypes = [vp]
    lib.smplfit_transfer_destroy.restype = None
    lib.smplfit_transfer_f32.argtypes = [vp, vp, i32, vp, vp]
    lib.smplfit_transfer_f32.restype = i32
    lib.smplfit_convert_plan_create.argtypes = [vp, vp, vp, C.POINTER(vp)]
    lib.smplfit_convert_plan_create.restype = i32
    lib.smplfit_convert_plan_destroy.argtypes = [vp]
    lib.smplfit_convert_plan_destroy.restype = None
    lib.smplfit_convert_workspace_bytes.argtypes = [vp, i32]
    lib.smplfit_convert_workspace_bytes.restype = sz
    lib.smplfit_convert_f32.argtypes = [vp, C.POINTER(ConvertArgs)]
    lib.smplfit_convert_f32.restype = i32
    lib.smplfit_reload_options.argtypes = []
    lib.smplfit_reload_options.restype = i32
    if os.environ.get('SMPLFIT_LIB') and not hasattr(lib, 'smplfit_abi_version'):
        _lib = lib  # an older build loaded by the A/B tools (tools/ab_fit.py): no version / share-table exports
        return lib
    lib.smplfit_abi_version.argtypes = []
    lib.smplfit_abi_version.restype = i32
    if lib.smplfit_abi_version() != SMPLFIT_ABI_VERSION:
        raise ImportError(f'{LIB_PATH} was built for ABI {lib.smplfit_abi_version()}, this package expects '
                          f'{SMPLFIT_ABI_VERSION}: rebuild it (python -m smplfitter_amd.build --force)')
    lib.smplfit_get_share_table.argtypes = [vp, i32, i32, _ip, sz, C.POINTER(sz)]
    lib.smplfit_get_share_table.restype = i32
    lib.smplfit_pick_share_mult.argtypes = [vp, i32, i32]
    lib.smplfit_pick_share_mult.restype = i32
    _lib = lib
    return lib


def check(status: int):
    """Map a C-ABI status to the exception types the reference raises for the same condition."""
    if status == SMPLFIT_OK:
        return
    msg = load().smplfit_last_error().decode()
    if status == SMPLFIT_ERR_BAD_ARG:
        raise ValueError(msg)
    if status == SMPLFIT_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise SmplfitError(f'smplfit status {status}: {msg}')


def make_desc(v_template, shapedirs, posedirs, weights, J_template, J_shapedirs, parents,
              J_regressor_post_lbs=None, is_smpl_family=True, kid_shapedir=None, kid_J_shapedir=None):
    """Build a ``ModelDesc`` from numpy arrays; returns (desc, keepalive list)."""
    f32c = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float32)  # noqa: E731
    arrs = dict(
        v_template=f32c(v_template), shapedirs=f32c(shapedirs), posedirs=f32c(posedirs),
        weights=f32c(weights), J_template=f32c(J_template), J_shapedirs=f32c(J_shapedirs),
    )
    par = np.ascontiguousarray(np.asarray(parents), dtype=np.int32)
    V, J = arrs['weights'].shape
    S = arrs['shapedirs'].shape[2]
    assert arrs['v_template'].shape == (V, 3)
    assert arrs['shapedirs'].shape == (V, 3, S)
    assert arrs['posedirs'].shape == (V, 3, 9 * (J - 1))
    assert arrs['J_template'].shape == (J, 3) and arrs['J_shapedirs'].shape == (J, 3, S)
    d = ModelDesc()
    d.num_vertices, d.num_joints, d.num_betas = V, J, S
    d.is_smpl_family = 1 if is_smpl_family else 0
    for k, a in arrs.items():
        setattr(d, k, a.ctypes.data_as(_fp))
    d.parents = par.ctypes.data_as(_ip)
    keep = list(arrs.values()) + [par]
    if J_regressor_post_lbs is not None:
        reg = f32c(J_regressor_post_lbs)
        d.J_regressor_post_lbs = reg.ctypes.data_as(_fp)
        d.regressor_num_vertices = reg.shape[1]
        keep.append(reg)
    else:
        d.J_regressor_post_lbs = None
        d.regressor_num_vertices = 0
    if kid_shapedir is not None:
        ks, kj = f32c(kid_shapedir), f32c(kid_J_shapedir)
        assert ks.shape == (V, 3) and kj.shape == (J, 3)
        d.enable_kid = 1
        d.kid_shapedir = ks.ctypes.data_as(_fp)
        d.kid_J_shapedir = kj.ctypes.data_as(_fp)
        keep += [ks, kj]
    else:
        d.enable_kid = 0
        d.kid_shapedir = None
        d.kid_J_shapedir = None
    return d, keep


class Handle:
    """Owns a ``smplfit_handle*``."""

    def __init__(self, desc: ModelDesc, host_only: bool = False):
        lib = load()
        self._h = C.c_void_p()
        check(lib.smplfit_create(C.byref(desc), SMPLFIT_CREATE_HOST_ONLY if host_only else 0, C.byref(self._h)))
        self.info = Info()
        check(lib.smplfit_get_info(self._h, C.byref(self.info)))

    @property
    def ptr(self):
        return self._h

    def table(self, name: str) -> np.ndarray:
        lib = load()
        n = C.c_size_t()
        check(lib.smplfit_get_table(self._h, TABLE_IDS[name], None, 0, C.byref(n)))
        out = np.zeros(n.value, np.int32)
        check(lib.smplfit_get_table(self._h, TABLE_IDS[name], out.ctypes.data_as(_ip), n.value, C.byref(n)))
        return out

    def share_table(self, kind: int, what: int) -> np.ndarray:
        """Cell tables of the batch-major vertex kernels (``smplfit_get_share_table``)."""
        lib = load()
        n = C.c_size_t()
        check(lib.smplfit_get_share_table(self._h, kind, what, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.int32)
        check(lib.smplfit_get_share_table(self._h, kind, what, out.ctypes.data_as(_ip), n.value, C.byref(n)))
        return out

    def workspace_bytes(self, batch: int) -> int:
        return int(load().smplfit_workspace_bytes(self._h, int(batch)))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            load().smplfit_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reload_options():
    """Re-read the SMPLFIT_* tuning variables (they are read once, at first use): for tests and A/B tools that
    switch kernel paths inside one process."""
    check(load().smplfit_reload_options())


class Transfer:
    """Owns a ``smplfit_transfer*``: the (V_out x V_in) CSR topology-transfer matrix on the current device."""

    def __init__(self, num_vertices_in: int, num_vertices_out: int, indptr, indices, values, host_only: bool = False):
        lib = load()
        ip = np.ascontiguousarray(indptr, dtype=np.int32)
        ix = np.ascontiguousarray(indices, dtype=np.int32)
        va = np.ascontiguousarray(values, dtype=np.float32)
        if ip.shape != (num_vertices_out + 1,) or ix.shape != va.shape or ix.ndim != 1 or int(ip[-1]) != ix.shape[0]:
            raise ValueError('Transfer: indptr / indices / values do not describe a (V_out x V_in) CSR matrix')
        self.shape = (int(num_vertices_out), int(num_vertices_in))
        self._t = C.c_void_p()
        check(lib.smplfit_transfer_create(
            int(num_vertices_in), int(num_vertices_out), ip.ctypes.data_as(_ip), ix.ctypes.data_as(_ip),
            va.ctypes.data_as(_fp), SMPLFIT_CREATE_HOST_ONLY if host_only else 0, C.byref(self._t)))

    @property
    def ptr(self):
        return self._t

    def close(self):
        if getattr(self, '_t', None) is not None and self._t.value:
            load().smplfit_transfer_destroy(self._t)
            self._t = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ConvertPlan:
    """Owns a ``smplfit_convert_plan*`` and keeps the two handles (and the matrix) it borrows alive.  Raises
    ``NotImplementedError`` when the fused conversion does not apply to the two models."""

    def __init__(self, h_in: Handle, h_out: Handle, transfer: 'Transfer | None'):
        self._keep = (h_in, h_out, transfer)
        self._p = C.c_void_p()
        check(load().smplfit_convert_plan_create(h_in.ptr, h_out.ptr, transfer.ptr if transfer is not None else None,
                                                 C.byref(self._p)))

    @property
    def ptr(self):
        return self._p

    def workspace_bytes(self, batch: int) -> int:
        return int(load().smplfit_convert_workspace_bytes(self._p, int(batch)))

    def close(self):
        if getattr(self, '_p', None) is not None and self._p.value:
            load().smplfit_convert_plan_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
