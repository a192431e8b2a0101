"""-m gpu: the parity evidence VERDICT round 1 asked for, on the device code that ships.

* the rotation primitives of the joint-level kernels evaluated ON THE DEVICE (smplfit_primitives_f32)
  against the reference's primitive goldens, degenerate inputs included (pt/rotation.py:100-110, 210-289);
* pose_rotvecs / shape_betas / trans / vertex statistics (max, p99, median) against the reference's
  outputs and against the fp64 arbiter, with the tight pose gate (<= 3e-4) on the
  well-conditioned fixtures (SMPL, fat-part SMPL-X); the thin-finger SMPL-X fixture is judged on vertices;
* full-size batches (BASELINE.json configs 2 and 5's per-GPU shard) compared DIRECTLY with the fp64 oracle on
  instances sampled across the batch (first, last, chunk boundary, instance-block boundaries);
* bench.py's N > 1 leg (self-launched ranks, result all-gather) driven with world 2 on the one GPU.
"""

import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import util
from test_gpu_parity import get_model, make_targets, t, to_np

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    return torch.device('cuda:0')


def _prim(op, a, b, out_shape, dev):
    from smplfitter_amd import _lib

    lib = _lib.load()
    ta = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    tb = None if b is None else torch.from_numpy(np.ascontiguousarray(b, np.float32)).to(dev)
    out = torch.full(out_shape, float('nan'), dtype=torch.float32, device=dev)
    _lib.check(lib.smplfit_primitives_f32(op, C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr() if tb is not None else 0),
                                          C.c_void_p(out.data_ptr()), len(a),
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_device_primitives(golden, dev):
    """Same assertions as tests/test_hostemu.py::test_primitives, but on the DEVICE build of sf_math.h
    (hardware v_rsq_f64 / v_rcp_f64 seeds + Newton steps instead of libm)."""
    g = golden('primitives')
    n = int(g['proj_n_random'])
    A = g['proj_in']
    R = _prim(0, A, None, A.shape, dev)
    assert np.isfinite(R).all()
    assert np.abs(R[:n] - g['proj_out'][:n]).max() < 2e-5
    det = np.linalg.det(R.astype(np.float64))
    assert np.abs(det - 1).max() < 1e-5  # every output is a proper rotation, degenerate inputs included
    assert np.abs(R @ np.swapaxes(R, -1, -2) - np.eye(3)).max() < 1e-5
    # degenerate inputs whose projection is well defined: both reflections, tiny / huge scale, nearly rank 2,
    # negative determinant (rank 1 / rank 2 inputs only have to give a proper rotation, checked above)
    for i in (n + 2, n + 3, n + 5, n + 6, n + 7, n + 8):
        assert np.abs(R[i] - g['proj_out'][i]).max() < 1e-4, i
    assert np.abs(R[n + 4] - np.eye(3)).max() == 0  # zero matrix -> identity
    rv = g['rotvec_in']
    M = _prim(1, rv, None, (len(rv), 3, 3), dev)
    assert np.abs(M - g['rotvec2mat_out']).max() < 1e-6
    out = _prim(2, g['rotvec2mat_out'], None, (len(rv), 3), dev)
    assert np.abs(out - g['mat2rotvec_out']).max() < 1e-5  # all four branches, angles near 0 and pi
    a, b = g['align_a'], g['align_b']
    Ra = _prim(3, a, b, (len(a), 3, 3), dev)
    assert np.abs(Ra[:-4] - g['align_out'][:-4]).max() < 1e-6
    # exactly antiparallel pairs: the formula is 0/0 there.  The reference's own CPU answer (golden) is a
    # half turn about a rounding-noise axis ("arbitrary", pt/rotation.py:217) that does not even map a to b;
    # the host build of sf_math.h gives the identity (no FMA contraction: the cross product is exactly 0),
    # the device build again a half turn about its own noise axis (v_fma leaves a 1e-8 residue).  Pinned:
    # a proper rotation that is the identity or a half turn (symmetric), nothing more — as in
    # tests/test_oracle_golden.py::test_primitive_goldens.
    Rap = Ra[-4:].astype(np.float64)
    assert np.abs(Rap @ np.swapaxes(Rap, -1, -2) - np.eye(3)).max() < 1e-5
    assert np.abs(np.linalg.det(Rap) - 1).max() < 1e-5
    assert np.abs(Rap - np.swapaxes(Rap, -1, -2)).max() < 1e-5
    # NaN input propagates (the reference's SVD would raise / return NaN; never a silent rotation)
    bad = A[:2].copy()
    bad[0, 1, 1] = np.nan
    Rn = _prim(0, bad, None, bad.shape, dev)
    assert np.isnan(Rn[0]).all() and np.isfinite(Rn[1]).all()
    # swing-twist of a bone part against the oracle's restatement (pt/bodyfitter.py:1389-1412)
    rs = np.random.RandomState(3)
    m = 128
    bref = rs.randn(m, 3).astype(np.float32)
    btgt = rs.randn(m, 3).astype(np.float32)
    bref[0] = 0  # divide_no_nan path
    Acov = (rs.randn(m, 3, 3) + 2 * np.eye(3)).astype(np.float32)
    Rst = _prim(4, bref, np.concatenate([btgt, Acov.reshape(m, 9)], 1), (m, 3, 3), dev)
    O = util.O
    br = O.divide_no_nan(bref, np.linalg.norm(bref, axis=-1, keepdims=True))
    bt = O.divide_no_nan(btgt, np.linalg.norm(btgt, axis=-1, keepdims=True))
    Rsw = O.align_unit_vectors(br.astype(np.float32), bt.astype(np.float32))
    Hm = Rsw @ np.swapaxes(Acov, -1, -2)
    trH = Hm[:, 0, 0] + Hm[:, 1, 1] + Hm[:, 2, 2]
    bHb = np.einsum('br,brc,bc->b', bt, Hm, bt)
    vee = np.stack([Hm[:, 1, 2] - Hm[:, 2, 1], Hm[:, 2, 0] - Hm[:, 0, 2], Hm[:, 0, 1] - Hm[:, 1, 0]], -1)
    ang = np.arctan2((bt * vee).sum(-1), trH - bHb)
    ref = O.rotvec2mat((bt * ang[:, None]).astype(np.float32)) @ Rsw
    assert np.abs(Rst - ref).max() < 5e-6


@pytest.mark.parametrize('name', ['smpl', 'smplxfat', 'smplx'])
def test_parity_statistics(name, model_root, golden, dev, capsys):
    """max / p99 / median of |ours - reference| for pose_rotvecs, shape_betas, trans and the re-forwarded
    vertices, against the reference's own outputs (goldens, CPU fp32 PyTorch) and against the fp64 arbiter.
    Gates: vertices <= 1e-4 m everywhere; pose_rotvecs <= 3e-4 on the well-conditioned fixtures (SMPL and the
    fat-part SMPL-X model; host emulation of this arithmetic: 2.2e-4 / 2.3e-4), betas /
    trans <= 1e-4 there; the thin-finger SMPL-X fixture (the reference's own fp32 noise: 5e-4 against fp64)
    is reported and judged on vertices."""
    g = golden(name)
    kind, md = util.load_md(model_root, name, g)
    om64, of64 = util.make_oracle(md, kind, np.float64)
    from smplfitter_amd.pt import BodyFitter, BodyModel

    m = BodyModel(kind, 'neutral', model_root=f'{model_root}/{util.model_dir(name)}', num_betas=10, device=dev)
    f = BodyFitter(m)
    tv, tj = g['target_vertices'], g['target_joints']
    rows = []
    for c in [c for c in util.fit_configs(g) if '_j_nw_' in c]:
        cfg = util.cfg_from_name(c)
        o = to_np(f.fit(t(tv, dev), t(tj, dev), num_iter=cfg['num_iter'], beta_regularizer=cfg['beta_regularizer'],
                        final_adjust_rots=cfg['final_adjust_rots'], requested_keys=['pose_rotvecs']))
        ref = {k: g[f'fit.{c}.{k}'] for k in ('pose_rotvecs', 'shape_betas', 'trans')}
        r64 = of64.fit(tv, tj, num_iter=cfg['num_iter'], beta_regularizer=cfg['beta_regularizer'],
                       final_adjust_rots=cfg['final_adjust_rots'])
        for tag, other in (('pt', ref), ('f64', r64)):
            va = om64.forward(o['pose_rotvecs'], o['shape_betas'], o['trans'])['vertices']
            vb = om64.forward(other['pose_rotvecs'], other['shape_betas'], other['trans'])['vertices']
            vl2 = np.linalg.norm(va - vb, axis=-1)
            row = dict(fixture=name, config=c, against=tag,
                       pose=util.stats(o['pose_rotvecs'], other['pose_rotvecs']),
                       betas=util.stats(o['shape_betas'], other['shape_betas']),
                       trans=util.stats(o['trans'], other['trans']),
                       vertex_l2=dict(max=float(vl2.max()), p99=float(np.percentile(vl2, 99)),
                                      median=float(np.median(vl2))))
            rows.append(row)
            assert row['vertex_l2']['max'] < 1e-4, row
            if name != 'smplx':
                assert row['pose']['max'] < 3e-4, row
                assert row['betas']['max'] < 1e-4 and row['trans']['max'] < 1e-5, row
        # the reference's own distance to the arbiter, for scale
        rows.append(dict(fixture=name, config=c, against='pt-vs-f64', pose=util.stats(ref['pose_rotvecs'], r64['pose_rotvecs'])))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', f'parity_stats_{name}.json'), 'w') as fh:
        json.dump(rows, fh, indent=1)
    with capsys.disabled():
        for r in rows:
            p = r['pose']
            print(f"\n[parity] {r['fixture']:9s} {r['config']:20s} vs {r['against']:9s} pose max {p['max']:.2e} "
                  f"p99 {p['p99']:.2e} med {p['median']:.2e}" +
                  (f" | betas {r['betas']['max']:.1e} trans {r['trans']['max']:.1e} vtx {r['vertex_l2']['max']:.2e}"
                   if 'betas' in r else ''), end='')


def test_gemm_split_precision(model_root, golden, dev, smplfit_env):
    """The posedirs contraction on the bf16 matrix cores (error-free bf16 split of both fp32 operands, three products
    per k-step + the bias row's third term, fp32 accumulate: k_posedirs_gemm_bf16x3, the default) is fp32-class at the
    vertex level: against the fp64 oracle its forward mesh is as accurate as the one computed with the fp32 MFMA
    (SMPLFIT_GEMM=f32; gate: <= 1.5x its error + 5e-8 m, and < 2e-6 m absolute), on small, on large and on EXTREME
    rotations (3 rad per component on every joint: rotation angles up to pi and beyond, pose features of order 2 — the
    pose-corrective offsets, where the dropped 2^-18 terms live, are then as large as they get), and whole fits agree
    to the last digits."""
    g = golden('smpl')
    kind, md = util.load_md(model_root, 'smpl', g)
    om64, _ = util.make_oracle(md, kind, np.float64)
    m, f = get_model(model_root, 'smpl', g, dev)
    rs = np.random.RandomState(11)
    B = 48
    errs = {}
    for scale in (0.1, 1.0, 3.0):
        pose = (rs.randn(B, 72) * scale).astype(np.float32)
        betas = (rs.randn(B, 10) * 0.5).astype(np.float32)
        trans = rs.randn(B, 3).astype(np.float32)
        ref = om64.forward(pose, betas, trans)['vertices']
        for mode in ('bf16x3', 'f32'):
            smplfit_env('SMPLFIT_GEMM', mode)
            v = m(t(pose, dev), t(betas, dev), t(trans, dev))['vertices'].cpu().numpy()
            errs[(scale, mode)] = float(np.abs(v - ref).max())
    smplfit_env('SMPLFIT_GEMM', None)
    print('\n[gemm] max |forward - fp64| (m):', {f'{k[1]}@{k[0]}': f'{v:.2e}' for k, v in errs.items()})
    for scale in (0.1, 1.0, 3.0):
        assert errs[(scale, 'f32')] < 2e-6 and errs[(scale, 'bf16x3')] < 2e-6
        assert errs[(scale, 'bf16x3')] <= 1.5 * errs[(scale, 'f32')] + 5e-8, errs
    tv, tj = make_targets(m, 256, 9, dev)
    out = {}
    for mode in ('bf16x3', 'f32'):
        smplfit_env('SMPLFIT_GEMM', mode)
        out[mode] = to_np(f.fit(tv, tj, num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs']))
    smplfit_env('SMPLFIT_GEMM', None)
    assert np.abs(out['bf16x3']['shape_betas'] - out['f32']['shape_betas']).max() < 2e-5
    assert np.abs(out['bf16x3']['trans'] - out['f32']['trans']).max() < 2e-6
    assert util.vertex_l2(om64, out['bf16x3'], out['f32']) < 2e-5


@pytest.mark.parametrize('B,pose_scale', [(300, 0.1), (1024, 0.1), (300, 1.0)])
def test_gemm_tiled_split_smplx(B, pose_scale, model_root, golden, dev, smplfit_env):
    """SMPL-X (K = 487): the batch-major path runs the tiled split-bf16 GEMM (k_posedirs_gemm_bf16x3_tiled, feature
    images by k_split_features); whole fits agree with the fp32-MFMA GEMM (SMPLFIT_GEMM=f32) to the last digits —
    B = 300 leaves the second 256-instance tile mostly empty, 1024 runs as two chunks; pose_scale 1.0: rotations of
    ~1 rad per component on all 55 joints (pose features of order 1)."""
    g = golden('smplx')
    kind, md = util.load_md(model_root, 'smplx', g)
    om64, _ = util.make_oracle(md, kind, np.float64)
    m, f = get_model(model_root, 'smplx', g, dev)
    tv, tj = make_targets(m, B, 5, dev, pose_scale=pose_scale)
    out = {}
    for mode in ('bf16x3', 'f32'):
        smplfit_env('SMPLFIT_GEMM', mode)
        out[mode] = to_np(f.fit(tv, tj, num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs']))
    smplfit_env('SMPLFIT_GEMM', None)
    assert np.isfinite(out['bf16x3']['pose_rotvecs']).all()
    assert np.abs(out['bf16x3']['shape_betas'] - out['f32']['shape_betas']).max() < 5e-5
    assert np.abs(out['bf16x3']['trans'] - out['f32']['trans']).max() < 2e-6
    n = min(B, 64)
    a = {k: v[:n] for k, v in out['bf16x3'].items()}
    b = {k: v[:n] for k, v in out['f32'].items()}
    assert util.vertex_l2(om64, a, b) < 6e-5  # the thin-finger fixture amplifies last-digit differences
    a = {k: v[-n:] for k, v in out['bf16x3'].items()}
    b = {k: v[-n:] for k, v in out['f32'].items()}
    assert util.vertex_l2(om64, a, b) < 6e-5  # the thin-finger fixture amplifies last-digit differences


def test_gemm_tiled_split_smplx_subset(model_root, golden, dev, smplfit_env):
    """A vertex subset of the SMPL-X-shaped model (1100 vertices: nine 128-vertex tiles, padded to ten for the
    256-column tiles of the tiled GEMM): fits agree between the split-bf16 and the fp32-MFMA GEMM."""
    from smplfitter_amd.pt import BodyFitter, BodyModel

    rs = np.random.RandomState(3)
    subset = np.sort(rs.choice(10475, 1100, replace=False))
    m = BodyModel('smplx', 'neutral', model_root=f'{model_root}/smplx', num_betas=10, device=dev, vertex_subset=subset)
    f = BodyFitter(m)
    B = 512
    tv, tj = make_targets(m, B, 9, dev)
    out = {}
    for mode in ('bf16x3', 'f32'):
        smplfit_env('SMPLFIT_GEMM', mode)
        out[mode] = to_np(f.fit(tv, tj, num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs']))
    smplfit_env('SMPLFIT_GEMM', None)
    assert np.isfinite(out['bf16x3']['pose_rotvecs']).all()
    # (1100 vertices condition the shape less well than the full mesh: last-digit differences of v_posed show in the betas)
    assert np.abs(out['bf16x3']['shape_betas'] - out['f32']['shape_betas']).max() < 5e-4
    assert np.abs(out['bf16x3']['trans'] - out['f32']['trans']).max() < 2e-5
    fw = {k: m(t(v['pose_rotvecs'], dev), t(v['shape_betas'], dev), t(v['trans'], dev))['vertices'] for k, v in out.items()}
    # few vertices per finger part: a part rotation estimated from a handful of them amplifies last-digit differences
    # of v_posed on single vertices, so the bulk is gated tightly and the worst vertex loosely; both
    # fits must also sit equally close to their targets
    d = (fw['bf16x3'] - fw['f32']).norm(dim=-1).flatten()
    ds = d[:: max(1, d.numel() // 1000000)]
    assert ds.median().item() < 2e-5 and torch.quantile(ds, 0.999).item() < 1.5e-4, (ds.median().item(), d.max().item())
    assert d.max().item() < 1e-3
    err = {k: (v - tv).norm(dim=-1).mean().item() for k, v in fw.items()}
    assert abs(err['bf16x3'] - err['f32']) < 1e-3 * err['f32'], err


def _sample_rows(B):
    """Instances spread over the batch: both ends, the chunk boundary (B/2), instance-block (64) and
    GEMM-tile (128) boundaries, and a seeded scatter."""
    rs = np.random.RandomState(0)
    fixed = [0, 1, 63, 64, 127, 128, B // 2 - 1, B // 2, B // 2 + 1, B // 2 + 63, B // 2 + 64, B - 129, B - 65, B - 64,
             B - 2, B - 1]
    extra = rs.choice(B, size=64 - len(fixed), replace=False).tolist()
    return np.array(sorted(set(fixed + extra)))


@pytest.mark.parametrize('B', [4096, 32768])
def test_full_size_vs_oracle_samples(B, model_root, golden, dev):
    """BASELINE.json config 2 (B = 4096) and the per-GPU shard of config 5 (B = 32768 = 262144 / 8) at full
    size: ~64 instances sampled across the batch are compared one by one with the fp64 oracle run on exactly
    the same targets (vertex gate 1e-4 m, betas 1e-4, trans 1e-5, pose 3e-4); plus the size-independent
    properties of test_full_size_properties (determinism, slice independence) on the shard."""
    g = golden('smpl')
    kind, md = util.load_md(model_root, 'smpl', g)
    om64, of64 = util.make_oracle(md, kind, np.float64)
    m, f = get_model(model_root, 'smpl', g, dev)
    tv, tj = make_targets(m, B, 42, dev)
    r = f.fit(tv, tj, num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs'])
    assert all(torch.isfinite(r[k]).all() for k in ('pose_rotvecs', 'shape_betas', 'trans'))
    idx = _sample_rows(B)
    ti = torch.from_numpy(idx).to(dev)
    o = {k: r[k][ti].cpu().numpy() for k in ('pose_rotvecs', 'shape_betas', 'trans')}
    ref = of64.fit(tv[ti].cpu().numpy(), tj[ti].cpu().numpy(), num_iter=3, beta_regularizer=1.0)
    assert util.vertex_l2(om64, o, ref) < 1e-4
    assert np.abs(o['shape_betas'] - ref['shape_betas']).max() < 1e-4
    assert np.abs(o['trans'] - ref['trans']).max() < 1e-5
    assert np.abs(o['pose_rotvecs'] - ref['pose_rotvecs']).max() < 3e-4
    if B == 32768:
        r2 = f.fit(tv, tj, num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs'])
        for k in ('pose_rotvecs', 'shape_betas', 'trans'):
            assert torch.equal(r[k], r2[k]), k  # run-to-run determinism at the shard size
        s = slice(B // 2 - 450, B // 2 + 450)  # (above 768 instances: the coarse cell tables, as the full batch)
        r3 = f.fit(tv[s], tj[s], num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs'])
        for k in ('pose_rotvecs', 'shape_betas', 'trans'):
            assert torch.equal(r[k][s], r3[k]), k  # an instance's result does not depend on its batch
        fw = m(r['pose_rotvecs'], r['shape_betas'], r['trans'])
        assert (fw['vertices'] - tv).norm(dim=-1).mean().item() < 1e-2  # round trip (beta_regularizer = 1)


@pytest.mark.parametrize('name,B', [('smplxfat', 4096), ('smpl1024', 16384)])
def test_full_size_vs_oracle_samples_c3_c4(name, B, model_root, golden, dev):
    """BASELINE.json configs 3 and 4 at full size, compared DIRECTLY with the fp64 oracle: the SMPL-X-shaped model
    (10475 vertices, 55 joints; the well-conditioned fat-part variant, so that the pose gate means something) at
    B = 4096 and the 1024-vertex SMPL subset at B = 16384 — instances sampled across the batch (both ends, the chunk
    boundary, instance-block and GEMM-tile boundaries).  Gates as test_full_size_vs_oracle_samples."""
    g = golden(name)
    kind, md = util.load_md(model_root, name, g)
    om64, of64 = util.make_oracle(md, kind, np.float64)
    m, f = get_model(model_root, name, g, dev)
    tv, tj = make_targets(m, B, 42, dev)
    r = f.fit(tv, tj, num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs'])
    assert all(torch.isfinite(r[k]).all() for k in ('pose_rotvecs', 'shape_betas', 'trans'))
    idx = _sample_rows(B)[::2] if name == 'smplxfat' else _sample_rows(B)  # the SMPL-X oracle is 4x the work
    ti = torch.from_numpy(idx).to(dev)
    o = {k: r[k][ti].cpu().numpy() for k in ('pose_rotvecs', 'shape_betas', 'trans')}
    ref = of64.fit(tv[ti].cpu().numpy(), tj[ti].cpu().numpy(), num_iter=3, beta_regularizer=1.0)
    errs = dict(vertex=util.vertex_l2(om64, o, ref), betas=float(np.abs(o['shape_betas'] - ref['shape_betas']).max()),
                trans=float(np.abs(o['trans'] - ref['trans']).max()),
                pose=float(np.abs(o['pose_rotvecs'] - ref['pose_rotvecs']).max()))
    print(f'\n[full-size] {name}@{B}: ' + ' '.join(f'{k} {v:.2e}' for k, v in errs.items()))
    assert errs['vertex'] < 1e-4 and errs['betas'] < 1e-4 and errs['trans'] < 1e-5 and errs['pose'] < 3e-4, errs


@pytest.mark.parametrize('name', ['smpl', 'smplx'])
def test_workspace_guards(name, model_root, golden, dev):
    """Stand-in for a device memcheck (the image has no ASAN-instrumented ROCm runtime, profiles/r03_asan_device.txt):
    the per-call workspace is embedded between two 1 MB guard regions whose byte pattern must survive every call, and
    is itself filled with NaN bit patterns before each call — a kernel reading a cell nobody wrote first would carry
    NaN into the results, which must be finite and bit-identical to a run on a zeroed workspace.  Covers the default
    fit with a partial last instance block, the kid unknown, joints omitted, vertex weights (the accumulate kernel), a
    warm start, the scale options (the accumulate kernel's extras), share_beta (the two-level reduction), a small batch
    (the fine cell tables), fit_with_known_shape (alignment kernels) and fit_with_known_pose (shape solve entry)."""
    from smplfitter_amd.pt import BodyFitter

    g = golden(name)
    m, f = get_model(model_root, name, g, dev)
    fk = BodyFitter(m, enable_kid=True)
    B = 1100 if name == 'smpl' else 300
    tv, tj = make_targets(m, B, 3, dev)
    guard = 1 << 20
    cases = [
        (f, False, dict(num_iter=2, beta_regularizer=1.0)),
        (fk, False, dict(num_iter=2, beta_regularizer=1.0)),
        (f, True, dict(num_iter=2, beta_regularizer=1.0, target_joints=None)),
        (f, False, dict(num_iter=2, vertex_weights=torch.rand(B, m.num_vertices, device=dev) + 0.5,
                        joint_weights=torch.rand(B, m.num_joints, device=dev) + 0.5)),
        (f, False, dict(num_iter=2, initial_pose_rotvecs=torch.zeros(B, 3 * m.num_joints, device=dev),
                        initial_shape_betas=torch.zeros(B, 10, device=dev))),
        (f, False, dict(num_iter=2, scale_target=True)),
        (f, False, dict(num_iter=2, scale_fit=True, vertex_weights=torch.rand(B, m.num_vertices, device=dev) + 0.5,
                        joint_weights=torch.rand(B, m.num_joints, device=dev) + 0.5)),
        (f, False, dict(num_iter=2, share_beta=True)),
        (f, False, dict(num_iter=2, beta_regularizer=1.0, _rows=37)),
        (f, False, dict(_call='known_shape', num_iter=2, scale_fit=True)),
        (f, True, dict(_call='known_shape', num_iter=1, target_joints=None,
                       vertex_weights=torch.rand(B, m.num_vertices, device=dev) + 0.5)),
        (f, False, dict(_call='known_pose')),
    ]
    zeros_pose = torch.zeros(B, 3 * m.num_joints, device=dev)
    zeros_betas = torch.zeros(B, 10, device=dev)
    for fitter, no_joints, kw in cases:
        kw = dict(kw)
        kw.pop('target_joints', None)
        call, rows = kw.pop('_call', 'fit'), kw.pop('_rows', B)
        h = m._native(dev, kid=fitter.enable_kid)
        n = h.workspace_bytes(rows)
        buf = torch.empty(n + 2 * guard, dtype=torch.uint8, device=dev)
        ws = buf[guard:guard + n]
        assert ws.data_ptr() % 256 == 0
        out = {}
        for fill in ('zero', 'nan'):
            buf.fill_(0xA5)
            if fill == 'zero':
                ws.zero_()
            else:
                ws.view(torch.int32).fill_(0x7FC00000 | 0x1234)  # quiet NaN pattern in every float / half a double
            kwr = {k: (v[:rows] if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
            tjr = None if no_joints else tj[:rows]
            if call == 'fit':
                r = fitter.fit(tv[:rows], tjr, _workspace=ws, **kwr)
            else:  # (these entries take the model's cached workspace: it is swapped for the guarded one)
                keep = m._workspace
                m._workspace = lambda h_, B_, device_, ws=ws: ws
                try:
                    if call == 'known_shape':
                        r = fitter.fit_with_known_shape(zeros_betas[:rows], tv[:rows], tjr, **kwr)
                    else:
                        r = fitter.fit_with_known_pose(zeros_pose[:rows], tv[:rows], tjr, **kwr)
                finally:
                    m._workspace = keep
            torch.cuda.synchronize()
            assert bool((buf[:guard] == 0xA5).all()) and bool((buf[guard + n:] == 0xA5).all()), 'guard region written'
            out[fill] = {k: v.clone() for k, v in r.items()}
            assert all(torch.isfinite(v).all() for v in r.values()), 'a result depends on uninitialised workspace'
        for k in out['zero']:
            assert torch.equal(out['zero'][k], out['nan'][k]), k


@pytest.mark.parametrize('kind,nb', [('smpl_w6', 10), ('smpl_b16', 16), ('smpl_b32', 32), ('smpl_w12', 10)])
def test_workspace_guards_round5_paths(kind, nb, model_root, dev):
    """The same guard / NaN-poison check on the kernels of round 5: pieces of eight joints (smpl_w6) and 16 betas (smpl_b16)
    on the batch-major kernels, and the general path (smpl_b32, smpl_w12: k_gen_accum_mfma / k_gen_lbs, stage scratch in the
    workspace) — default fit with a partial last block, the kid unknown, joints omitted, weights, a small batch, known
    shape, known pose."""
    from smplfitter_amd.pt import BodyFitter, BodyModel

    m = BodyModel('smpl', 'neutral', model_root=f'{model_root}/{kind}', num_betas=nb, device=dev)
    f, fk = BodyFitter(m), BodyFitter(m, enable_kid=True)
    B = 300
    rs = np.random.RandomState(3)
    fw = m(t((rs.randn(B, 3 * m.num_joints) * 0.1).astype(np.float32), dev), t((rs.randn(B, nb) * 0.3).astype(np.float32), dev),
           t(rs.randn(B, 3).astype(np.float32), dev))
    tv, tj = fw['vertices'], fw['joints']
    guard = 1 << 20
    vw, jw = torch.rand(B, m.num_vertices, device=dev) + 0.5, torch.rand(B, m.num_joints, device=dev) + 0.5
    cases = [
        (f, False, dict(num_iter=2, beta_regularizer=1.0)),
        (fk, False, dict(num_iter=2, beta_regularizer=1.0)),
        (f, True, dict(num_iter=2, beta_regularizer=1.0)),
        (f, False, dict(num_iter=2, vertex_weights=vw, joint_weights=jw)),
        (f, False, dict(num_iter=2, beta_regularizer=1.0, _rows=37)),
        (f, False, dict(_call='known_shape', num_iter=2)),
        (f, False, dict(_call='known_pose')),
    ]
    if m.kernel_path() == 'general':  # the options the general path gained last: the scaled solve's fp64 extras, the
        cases += [                    # chunked share sums (300 instances: two chunks), both together
            (f, False, dict(num_iter=2, beta_regularizer=1.0, scale_target=True)),
            (f, False, dict(num_iter=2, beta_regularizer=1.0, scale_fit=True, vertex_weights=vw, joint_weights=jw)),
            (f, False, dict(num_iter=2, beta_regularizer=1.0, share_beta=True)),
            (fk, False, dict(num_iter=2, beta_regularizer=1.0, share_beta=True, scale_fit=True)),
        ]
    zeros_pose, zeros_betas = torch.zeros(B, 3 * m.num_joints, device=dev), torch.zeros(B, nb, device=dev)
    for fitter, no_joints, kw in cases:
        kw = dict(kw)
        call, rows = kw.pop('_call', 'fit'), kw.pop('_rows', B)
        h = m._native(dev, kid=fitter.enable_kid)
        n = h.workspace_bytes(rows)
        buf = torch.empty(n + 2 * guard, dtype=torch.uint8, device=dev)
        ws = buf[guard:guard + n]
        out = {}
        for fill in ('zero', 'nan'):
            buf.fill_(0xA5)
            if fill == 'zero':
                ws.zero_()
            else:
                ws.view(torch.int32).fill_(0x7FC00000 | 0x1234)
            kwr = {k: (v[:rows] if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
            tjr = None if no_joints else tj[:rows]
            if call == 'fit':
                r = fitter.fit(tv[:rows], tjr, _workspace=ws, **kwr)
            else:
                keep = m._workspace
                m._workspace = lambda h_, B_, device_, ws=ws: ws
                try:
                    if call == 'known_shape':
                        r = fitter.fit_with_known_shape(zeros_betas[:rows], tv[:rows], tjr, **kwr)
                    else:
                        r = fitter.fit_with_known_pose(zeros_pose[:rows], tv[:rows], tjr, **kwr)
                finally:
                    m._workspace = keep
            torch.cuda.synchronize()
            assert bool((buf[:guard] == 0xA5).all()) and bool((buf[guard + n:] == 0xA5).all()), ('guard region written', kind, call)
            out[fill] = {k: v.clone() for k, v in r.items()}
            assert all(torch.isfinite(v).all() for v in r.values()), ('a result depends on uninitialised workspace', kind, call, kw.keys())
        for k in out['zero']:
            assert torch.equal(out['zero'][k], out['nan'][k]), (kind, call, k)


@pytest.mark.usefixtures('two_chunks')
@pytest.mark.parametrize('name,B,reps', [('smpl', 4096, 200), ('smplx', 2048, 60)])
def test_neighbour_stress(name, B, reps, model_root, golden, dev):
    """The split-bf16 GEMMs must never share a CU with another kernel (k_posedirs_gemm_bf16x3 and, for the SMPL-X-shaped
    model, k_posedirs_gemm_bf16x3_tiled: "exclusive CU"): beside their LDS-fed bf16 MFMAs, other kernels' waves were
    seen to read wrong lanes.  The guard is an occupancy one (256 VGPRs x 8 waves), so it is stressed: default two-chunk
    fits (200 at B = 4096 / 60 at 2048) while a SECOND handle fits on its own stream from another thread and torch
    streams elementwise kernels on a third — every result must be bit-identical to the first.  Also checks the kernels
    really allocate the whole register file."""
    import threading

    from smplfitter_amd.pt import BodyFitter, BodyModel

    g = golden(name)
    m, f = get_model(model_root, name, g, dev)
    tv, tj = make_targets(m, B, 42, dev)
    h = m._native(dev)
    assert h.info.gemm_vgprs >= 256, h.info.gemm_vgprs  # the occupancy guard: 8 waves x 256 registers = one CU
    ws = torch.empty(h.workspace_bytes(B), dtype=torch.uint8, device=dev)
    kw = dict(num_iter=3, beta_regularizer=1.0, requested_keys=['pose_rotvecs'], _workspace=ws)
    ref = f.fit(tv, tj, **kw)
    ref = {k: ref[k].clone() for k in ('pose_rotvecs', 'shape_betas', 'trans')}
    torch.cuda.synchronize()
    # foreign work: another handle of the same model (its own constants, streams, workspace) + torch elementwise
    m2 = BodyModel(name, 'neutral', model_root=f'{model_root}/{name}', num_betas=10, device=dev)
    f2 = BodyFitter(m2)
    B2 = B // 2
    tv2, tj2 = make_targets(m2, B2, 7, dev)
    stop = threading.Event()
    errors = []

    def foreign_fit():
        try:
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                ws2 = torch.empty(m2._native(dev).workspace_bytes(B2), dtype=torch.uint8, device=dev)
                while not stop.is_set():
                    f2.fit(tv2, tj2, num_iter=2, beta_regularizer=1.0, _workspace=ws2)
                    s.synchronize()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    def foreign_elementwise():
        try:
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                x = torch.randn(1 << 24, device=dev)
                while not stop.is_set():
                    for _ in range(8):
                        x = torch.sin(x) * 1.0001 + 0.5
                    s.synchronize()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=foreign_fit), threading.Thread(target=foreign_elementwise)]
    for th in threads:
        th.start()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    try:
        for _ in range(reps):
            r = f.fit(tv, tj, **kw)
            for k, v in ref.items():
                bad += (r[k] != v).sum()
        torch.cuda.synchronize()
    finally:
        stop.set()
        for th in threads:
            th.join()
    assert not errors, errors
    assert int(bad.item()) == 0, f'{int(bad.item())} result values differed from the first fit under foreign load'


@pytest.mark.parametrize('launcher', ['self', 'torchrun'])
def test_bench_world2(launcher, model_root):
    """bench.py's N > 1 leg: `python bench.py --gpus 2` starts its two ranks itself (and `torch.distributed.run`
    starts them for it), every rank fits its own shard, the packed rows are all-gathered and rank 0 prints
    n_gpus = 2 with the aggregate rate.  The box has ONE GPU and RCCL refuses two ranks on one device, so the
    collective runs on gloo here (--backend gloo --oversubscribe, test-only switches); the nccl path is the
    same code with the default backend."""
    env = dict(os.environ, SMPLFIT_SYNTH_ROOT=model_root)
    common = ['--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '256', '--no-cpu-baseline',
              '--backend', 'gloo', '--oversubscribe']
    if launcher == 'self':
        cmd = [sys.executable, 'bench.py'] + common
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
               '--master-addr', '127.0.0.1', '--master-port', str(29000 + os.getpid() % 500), 'bench.py'] + common
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-2000:] + p.stderr[-3000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['nccl_world_size'] == 2 and out['scaling'] == 'weak'
    assert out['config']['batch_per_gpu'] == 256 and out['value'] > 0
    assert 'all_gather_into_tensor' in out['collective']


def test_bench_refuses_missing_devices():
    """`python bench.py --gpus 64` on a box with fewer devices fails loudly instead of printing n_gpus = 1."""
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', '64', '--steps', '1', '--warmup', '0'], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and 'device' in (p.stderr + p.stdout)


def test_bench_world1_rccl_gather(model_root):
    """The RCCL leg of bench.py with ONE rank (`--collective-always`): a world-1 nccl process group, the packed result
    rows of every step all-gathered by `all_gather_into_tensor` — the same call N ranks make over xGMI (the box has one
    GPU: this is the part of the collective path that can run here)."""
    env = dict(os.environ, SMPLFIT_SYNTH_ROOT=model_root)
    cmd = [sys.executable, 'bench.py', '--steps', '2', '--warmup', '1', '--batch', '512', '--no-cpu-baseline',
           '--collective-always']
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-2000:] + p.stderr[-3000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['nccl_world_size'] == 1 and out['value'] > 0
    assert out['collective'].startswith('nccl all_gather_into_tensor') and 'side stream' in out['collective']
    assert out['roofline']['stream_bytes_per_launch'] and out['roofline']['frac_of_section8d_bytes'] > 0
    # round 6: the gather of step k runs behind the fit of step k + 1 (dist.OverlappedGather); the line carries the same
    # steps with the gather in line and without any collective beside the headline
    mg = out['multi_gpu']
    assert mg['gather'].startswith('overlapped') and mg['ms_per_step_no_collective'] > 0 and mg['ms_per_step_gather_inline'] > 0
    assert out['ms_per_step'] < 1.5 * mg['ms_per_step_no_collective']  # (2 steps of a small batch: a loose bound)
