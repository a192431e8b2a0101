"""The C-ABI library loads here (no GPU) and exports every symbol include/smplfit.h declares;
host-only handles expose the tables BodyFitter.__init__ of the reference would build."""

import os.path as osp
import re

import numpy as np
import pytest

import util
from smplfitter_amd import _lib, build

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    build.build(verbose=False)
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(osp.join(ROOT, 'include', 'smplfit.h')).read()
    declared = set(re.findall(r'\b(smplfit_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    for s in declared:
        assert getattr(lib, s) is not None


def test_build_id_covers_csrc():
    """Every file under csrc/ (and the public header) enters needs_build() and the build id smplfit_version() reports:
    an edit to any kernel file must change the id the profiles are tied to (round 5: kernels_gen.inc was missing)."""
    import os
    import re

    from smplfitter_amd import build as b

    hashed = {os.path.normpath(os.path.join(b.CSRC, f)) for f in b.SOURCES + b._headers()}
    for f in os.listdir(b.CSRC):
        assert os.path.normpath(os.path.join(b.CSRC, f)) in hashed, f
    # every #include "..." of the translation units resolves to a hashed file
    for src in b.SOURCES:
        for inc in re.findall(r'^#include "([^"]+)"', open(os.path.join(b.CSRC, src)).read(), re.M):
            assert os.path.normpath(os.path.join(b.CSRC, inc)) in hashed, (src, inc)
    assert os.path.normpath(os.path.join(b.CSRC, '../../include/smplfit.h')) in hashed
    assert re.fullmatch(r'[0-9a-f]{12}', b.source_id())


def test_no_device_fails_loudly(lib, model_root):
    import torch

    if torch.cuda.is_available():
        pytest.skip('GPU present')
    kind, md = util.load_md(model_root, 'smpl')
    import hostemu_util as H

    desc, keep = H.desc_from_md(md, kind)
    with pytest.raises(_lib.SmplfitError):
        _lib.Handle(desc)  # uploading without a device must raise, never fall back
    from smplfitter_amd.pt import BodyModel

    m = BodyModel('smpl', 'neutral', model_root=f'{model_root}/smpl', num_betas=10)
    with pytest.raises(RuntimeError):
        m(pose_rotvecs=torch.zeros(1, 72))


@pytest.mark.parametrize('name', ['smpl', 'smplx', 'smpl1024', 'smpl_rnd', 'smpl_w6'])
def test_host_tables_match_reference_structure(lib, name, model_root, golden):
    g = golden(name)
    kind, md = util.load_md(model_root, name, g)
    import hostemu_util as H

    desc, keep = H.desc_from_md(md, kind)
    h = _lib.Handle(desc, host_only=True)
    om, of = util.make_oracle(md, kind)
    assert (h.table('part_assignment') == of.part).all()
    ptype = h.table('part_type')
    assert sorted(np.where(ptype == 1)[0]) == of.multi
    assert sorted(np.where(ptype == 2)[0]) == of.bone
    assert sorted(np.where(ptype == 3)[0]) == of.leaf
    assert sorted(np.where(h.table('adj_flag') == 1)[0]) == sorted(of.adjustable)
    assert sorted(np.where(h.table('used_part') == 1)[0]) == of.used_parts
    starts = h.table('fk_level_start')
    order = h.table('fk_order')
    levels = [sorted(order[starts[i]:starts[i + 1]].tolist()) for i in range(len(starts) - 1)]
    assert levels == [sorted(lv) for lv in of.levels]
    if name == 'smpl':  # structural facts of SURVEY.md §8a
        assert of.multi == [0, 9] and of.leaf == [15, 22, 23] and len(of.bone) == 17
        assert [len(lv) for lv in levels] == [3, 3, 3, 5, 3, 2, 2, 2]
    if name == 'smplx':
        assert of.multi == [0, 9, 15, 20, 21] and len(of.bone) == 35 and len(of.leaf) == 13
    perm = h.table('sort_perm')
    V = md.num_vertices
    assert sorted(perm[perm >= 0].tolist()) == list(range(V))
    seg = h.table('segments').reshape(-1, 3)
    used = np.isin(of.part, of.used_parts)
    assert h.info.num_used_vertices == used.sum() == seg[:, 1].sum()
    for s, c, p in seg:  # every segment is one part, at most one wave wide
        assert 0 < c <= 64 and (of.part[perm[s:s + c]] == p).all()
    assert h.info.padded_vertices % 128 == 0
    nj = 8 if name.endswith('_w6') else 4  # six weights per vertex: eight pairs, pieces of up to eight joints (round 5)
    assert h.info.skin_width == nj
    # pieces (what the batch-major vertex kernels walk): a partition of the sorted slots into runs of one part
    # with at most four skinning joints each, used parts first; and at least one padding slot behind the
    # vertices whenever those kernels can apply
    pcs = h.table('vertex_pieces').reshape(-1, 5)
    assert pcs[0, 0] == 0 and (pcs[1:, 0] == pcs[:-1, 0] + pcs[:-1, 1]).all()
    assert pcs[-1, 0] + pcs[-1, 1] == V
    assert (pcs[:, 1] > 0).all() and (pcs[:, 4] >= 1).all() and (pcs[:, 4] <= nj).all()
    for s, c, p, u, nq in pcs:
        assert (of.part[perm[s:s + c]] == p).all() and u == int(p in of.used_parts)
        joints = np.unique(np.nonzero(md.weights[perm[s:s + c]])[1])
        assert len(joints) == nq
    assert (np.diff(pcs[:, 3]) <= 0).all()  # used pieces first
    check_share_tables(h, of, md, perm, V, plain=name != 'smpl_rnd', nj=nj)
    if V >= 1024:
        assert h.info.padded_vertices > V
    assert h.workspace_bytes(64) > 0
    h.close()


@pytest.mark.parametrize('kind', list(util.GENERAL_KINDS))
def test_general_models_create(lib, kind, model_root):
    """More than 16 betas / more than 8 skinning weights per vertex: the handle is created (rounds 1-4 refused these at
    smplfit_create), reports the general path, the caller's own num_betas, and the part tables of the reference."""
    import hostemu_util as H

    md = util.load_general_md(model_root, kind)
    desc, keep = H.desc_from_md(md, 'smpl')
    h = _lib.Handle(desc, host_only=True)
    _, of = util.make_oracle(md, 'smpl')
    assert h.info.vertex_path == _lib.SMPLFIT_PATH_GENERAL
    assert h.info.num_betas == md.shapedirs.shape[2] and h.info.has_kid == 0
    assert h.info.skin_width == (12 if kind == 'smpl_w12' else 4)
    assert (h.table('part_assignment') == of.part).all()
    assert len(h.table('vertex_pieces')) == 0 and len(h.table('cell_counts')) == 0 and len(h.table('joint_pairs')) == 0
    perm = h.table('sort_perm')
    assert sorted(perm[perm >= 0].tolist()) == list(range(md.num_vertices))
    assert h.workspace_bytes(8) > 0
    h.close()
    desc, keep = H.desc_from_md(md, 'smpl', enable_kid=True)
    hk = _lib.Handle(desc, host_only=True)
    assert hk.info.has_kid == 1 and hk.info.num_betas == md.shapedirs.shape[2]
    hk.close()


def check_share_tables(h, of, md, perm, V, plain=True, nj=4):
    """Every cell table deals its domain (all slots / used parts / adjustable parts) exactly once to its cells, in
    cells of nearly equal cost, and its rows say where the partial sums go.  ``nj`` = 8: pieces of up to eight joints,
    whose record is a PAIR of 12-int records (joints / local slots 4..7 in the second one, the other fields repeated)."""
    ncells = h.table('cell_counts')
    assert len(ncells) == 8 and all(n >= 8 and n & (n - 1) == 0 for n in ncells)  # powers of two; (coarse, fine) x 4 kinds
    assert all(ncells[4 + k] >= ncells[k] for k in range(4))
    slot_part = of.part[perm[:V]]
    domains = {0: np.ones(V, bool), 1: np.ones(V, bool), 2: np.isin(slot_part, of.used_parts),
               3: np.isin(slot_part, sorted(of.adjustable))}
    for table in range(8):
        kind, fine = table % 4, table >= 4
        nc = int(ncells[table])
        start = h.share_table(table, 0)
        rec = h.share_table(table, 1).reshape(-1, 12 * (nj // 4))
        if nj == 8:  # fold the pair into one logical record: count, joints[8], slots[8], first slot, row, tail
            assert (rec[:, [0, 9, 10, 11]] == rec[:, [12, 21, 22, 23]]).all()
            rec = np.concatenate([rec[:, 0:1], rec[:, 1:5], rec[:, 13:17], rec[:, 5:9], rec[:, 17:21], rec[:, 9:12]], 1)
        rows = h.share_table(table, 2)
        assert len(start) == nc + 1 and start[0] == 0 and start[-1] == len(rec) - 1
        assert (rec[-1] == 0).all()  # sentinel
        seen = np.zeros(V, np.int32)
        costs, row = [], 0
        for k in range(nc):
            cost, slots = 0, {}
            assert start[k + 1] > start[k]  # no empty cell
            for r in rec[start[k]:start[k + 1]]:
                cnt, js, loc, s0, close, tail = r[0], r[1:1 + nj], r[1 + nj:1 + 2 * nj], r[-3], r[-2], r[-1]
                assert cnt > 0
                seen[s0:s0 + cnt] += 1
                cost += cnt + (cnt & 1) + 3
                assert len(set(slot_part[s0:s0 + cnt])) == 1
                used_j = set(np.unique(np.nonzero(md.weights[perm[s0:s0 + cnt]])[1]))
                assert used_j <= set(js.tolist())
                if kind == 0:
                    for j, q in zip(js, loc):  # local slots are stable inside a segment
                        assert 0 <= q < 12 and slots.setdefault(int(j), int(q)) == q
                if close >= 0:
                    assert close == row
                    if kind == 0:
                        rj = rows.reshape(-1, 12)[row]
                        assert (tail & 0xff) == len(slots) == (rj >= 0).sum()
                        assert all(rj[q] == j for j, q in slots.items())
                        slots = {}
                    else:
                        assert rows[row] == slot_part[s0]
                    row += 1
            last = rec[start[k + 1] - 1]
            assert last[-2] >= 0  # a cell always closes its last row: the rows do not depend on the multiplier
            if kind == 0:
                assert (last[-1] >> 8) == k + 1 and all((r[-1] >> 8) == 0 for r in rec[start[k]:start[k + 1] - 1])
            costs.append(cost)
        assert (seen == domains[kind].astype(np.int32)).all()
        assert row == (len(rows) // 12 if kind == 0 else len(rows))
        # balanced: the longest cell is within a few steps of the mean (the last one may be shorter)
        assert max(costs) <= np.mean(costs) + 8, (table, max(costs), np.mean(costs))
        assert np.mean(costs) >= (10 if fine else 60)  # cells of ~75+ steps (fine tables: 12+)
    # multiplier: one round of the chip where the batch allows it
    lib = _lib.load()
    if V > 6000 and plain:  # (the random-joint variant has a piece per vertex: 256 cells)
        m4k, m32k = lib.smplfit_pick_share_mult(h.ptr, 0, 4096), lib.smplfit_pick_share_mult(h.ptr, 0, 32768)
        assert (4096 // 64) * ncells[0] // m4k == 4096 and (32768 // 64) * ncells[0] // m32k == 4096


def test_create_rejects_bad_models(lib, model_root):
    kind, md = util.load_md(model_root, 'smpl')
    import hostemu_util as H

    desc, keep = H.desc_from_md(md, kind)
    desc.num_joints = 100
    with pytest.raises(NotImplementedError):
        _lib.Handle(desc, host_only=True)
    desc, keep = H.desc_from_md(md, kind)
    desc.v_template = None
    with pytest.raises(ValueError):
        _lib.Handle(desc, host_only=True)


def test_torch_library_operators_registered(model_root):
    """`smplfitter_amd::fit` / `::forward` exist with shape functions: under FakeTensorMode (what
    torch.compile / export trace with) they return the result shapes without touching a GPU."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode

    from smplfitter_amd.pt import BodyModel, ops

    m = BodyModel('smpl', 'neutral', model_root=f'{model_root}/smpl', num_betas=10)
    mid = ops.register_model(m)
    assert ops.register_model(m) == mid
    with FakeTensorMode():
        tv = torch.empty((5, m.num_vertices, 3))
        out = torch.ops.smplfitter_amd.fit(mid, False, tv, None, None, None, 3, 1.0, 0.0, 1.0, True, None,
                                           None, None, False, 0, 0.0)
        assert [tuple(t.shape) for t in out] == [(5, 72), (5, 10), (5, 3), (5,), (5, 24, 3, 3), (5, 24, 3, 3), (0,)]
        fw = torch.ops.smplfitter_amd.forward(mid, torch.empty((5, 72)), None, None, None, None, None, True)
        assert [tuple(t.shape) for t in fw] == [(5, 24, 3), (5, 24, 3, 3), (5, m.num_vertices, 3)]


def test_model_unpickler_is_restricted(tmp_path):
    """A body-model / deftrafo pickle may only reconstruct arrays, sparse matrices and plain containers:
    any other global (here ``os.system``) is refused instead of imported."""
    import io
    import pickle

    import scipy.sparse as sp

    from smplfitter_amd import modelio

    class Evil:
        def __reduce__(self):
            import os

            return (os.system, ('true',))

    with pytest.raises(pickle.UnpicklingError):
        modelio.restricted_load(io.BytesIO(pickle.dumps({'mtx': Evil()}, protocol=2)))
    m = sp.random(5, 7, density=0.4, format='csc', dtype=np.float64, random_state=0)
    back = modelio.restricted_load(io.BytesIO(pickle.dumps({'mtx': m, 'a': np.arange(4.0)}, protocol=2)))
    assert (back['mtx'] != m).nnz == 0 and np.array_equal(back['a'], np.arange(4.0))


def test_transfer_create_validates(lib):
    """smplfit_transfer_create (host-only: no GPU needed) rejects malformed CSR input with the BAD_ARG status
    (ValueError through _lib.check) and accepts empty rows."""
    import ctypes as C

    from smplfitter_amd import _lib

    indptr = np.array([0, 2, 2, 3], np.int32)
    indices = np.array([0, 4, 1], np.int32)
    values = np.array([0.5, 0.5, 1.0], np.float32)
    tr = _lib.Transfer(5, 3, indptr, indices, values, host_only=True)
    assert tr.shape == (3, 5)
    # a host-only matrix cannot compute
    with pytest.raises(_lib.SmplfitError):
        _lib.check(lib.smplfit_transfer_f32(tr.ptr, C.c_void_p(16), 1, C.c_void_p(16), None))
    tr.close()
    with pytest.raises(ValueError):  # column index outside the input vertices
        _lib.Transfer(4, 3, indptr, indices, values, host_only=True)
    with pytest.raises(ValueError):  # decreasing indptr
        _lib.Transfer(5, 3, np.array([0, 2, 1, 3], np.int32), indices, values, host_only=True)
    with pytest.raises(ValueError):  # indptr[0] != 0
        _lib.Transfer(5, 3, np.array([1, 2, 2, 3], np.int32), indices, values, host_only=True)
    with pytest.raises(ValueError):  # shapes that are not a CSR matrix (caught before the call)
        _lib.Transfer(5, 3, indptr[:-1], indices, values, host_only=True)
    out = C.c_void_p()
    assert lib.smplfit_convert_plan_create(None, None, None, C.byref(out)) == _lib.SMPLFIT_ERR_BAD_ARG
    assert lib.smplfit_convert_workspace_bytes(None, 8) == 0
    assert lib.smplfit_convert_f32(None, None) == _lib.SMPLFIT_ERR_BAD_ARG


def test_reload_options(lib, monkeypatch, model_root, golden):
    """The tuning variables are read once; smplfit_reload_options() picks up a change: the cells a wave of the vertex
    passes walks at B = 4096 follow SMPLFIT_BM_SLOTS (one round of 4096 waves by default, two cells per wave when the
    pass is dealt for 2048 resident waves)."""
    import hostemu_util as H

    from smplfitter_amd import _lib

    kind, md = util.load_md(model_root, 'smpl', golden('smpl'))
    desc, keep = H.desc_from_md(md, kind)
    h = _lib.Handle(desc, host_only=True)
    monkeypatch.delenv('SMPLFIT_BM_SLOTS', raising=False)
    _lib.reload_options()
    assert lib.smplfit_pick_share_mult(h.ptr, 0, 4096) == 1
    monkeypatch.setenv('SMPLFIT_BM_SLOTS', '2048')
    assert lib.smplfit_pick_share_mult(h.ptr, 0, 4096) == 1  # not re-read yet
    _lib.reload_options()
    assert lib.smplfit_pick_share_mult(h.ptr, 0, 4096) == 2
    monkeypatch.delenv('SMPLFIT_BM_SLOTS')
    _lib.reload_options()
    assert lib.smplfit_pick_share_mult(h.ptr, 0, 4096) == 1
    h.close()
