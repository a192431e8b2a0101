"""Build ``libsmplfit_hip.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m smplfitter_amd.build [--force]
"""

from __future__ import annotations

import os
import os.path as osp
import shutil
import subprocess
import sys

HERE = osp.dirname(osp.abspath(__file__))
CSRC = osp.join(HERE, 'csrc')
OUT = osp.join(HERE, 'libsmplfit_hip.so')
SOURCES = ['smplfit_hip.hip', 'sf_tables.cpp']


def _headers():
    """Every file under csrc/ that is not a translation unit, plus the public header: the list is read
    from the directory, so a new ``*.inc`` / ``*.h`` is part of ``needs_build()`` and of the build id the
    moment it exists (``tests/test_cabi.py::test_build_id_covers_csrc``)."""
    own = sorted(f for f in os.listdir(CSRC)
                 if f not in SOURCES and f.endswith(('.h', '.inc', '.hip', '.cpp', '.hpp')))
    return own + ['../../include/smplfit.h']


HEADERS = _headers()


def _hipcc():
    for c in (os.getenv('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and osp.exists(c):
            return c
    raise RuntimeError('hipcc not found (needed to build libsmplfit_hip.so)')


def needs_build():
    if not osp.exists(OUT):
        return True
    t = osp.getmtime(OUT)
    return any(osp.getmtime(osp.join(CSRC, f)) > t for f in SOURCES + _headers())


def source_id():
    """sha256 over the sources the library is built from (12 hex digits): the build id smplfit_version()
    reports, so that profiles can be tied to the tree they were measured on."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(SOURCES + _headers()):
        with open(osp.join(CSRC, f), 'rb') as fh:
            h.update(f.encode() + b'\0' + fh.read())
    return h.hexdigest()[:12]


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    cmd = [
        _hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
        '-Wno-unused-value', f'-DSMPLFIT_BUILD_ID="{source_id()}"',
        *[osp.join(CSRC, s) for s in SOURCES], '-o', OUT + '.tmp',
    ]  # fmt: skip
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(OUT + '.tmp', OUT)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(OUT)
