"""pytest configuration: the ``gpu`` marker and shared fixtures.

``-m "not gpu"`` tests run in the build container (no GPU): the oracle against the golden vectors,
host logic, C-ABI symbol export, the host-emulated kernel math.  ``-m gpu`` tests are the parity
tests proper and call the HIP kernels through the C-ABI on an MI355X.
"""

import os
import os.path as osp
import sys

import numpy as np
import pytest

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def model_root():
    from smplfitter_amd import synth

    return synth.ensure_model_root(kinds=('smpl', 'smplx', 'smplx_fat', 'smpl_b16', 'smpl_w6_b16', 'smpl_w6', 'smplx_w6', 'smpl_rnd', 'smpl_b32', 'smpl_b300', 'smpl_w12', 'smpl_b100', 'smpl_b400'),
                                   seed=0)


@pytest.fixture(scope='session')
def golden():
    gdir = osp.join(ROOT, 'tests', 'golden')

    def load(name):
        return dict(np.load(osp.join(gdir, f'golden_{name}.npz'), allow_pickle=False))

    return load


@pytest.fixture(scope='session')
def data_root():
    """Synthetic topology-transfer / mirror files in the official layout ($DATA_ROOT/body_models)."""
    from smplfitter_amd import synth

    return synth.write_transfer_files(os.getenv('SMPLFIT_SYNTH_DATA', '/tmp/smplfit_synth_data_seed0'))


@pytest.fixture(scope='session')
def data_root_fat():
    """The same for SMPL <-> the fat-part SMPL-X variant (the cross-topology BodyConverter fixture)."""
    from smplfitter_amd import synth

    return synth.write_transfer_files(os.getenv('SMPLFIT_SYNTH_DATA_FAT', '/tmp/smplfit_synth_data_fat_seed0'),
                                      smplx_kind='smplx_fat')


@pytest.fixture
def smplfit_env(monkeypatch):
    """Set / clear a SMPLFIT_* tuning variable for one test: the library reads them once, so every change is followed
    by smplfit_reload_options(); restored (and re-read) afterwards."""
    from smplfitter_amd import _lib

    def set_var(name, value):
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, value)
        _lib.reload_options()

    yield set_var
    monkeypatch.undo()
    _lib.reload_options()


@pytest.fixture
def two_chunks(smplfit_env):
    """Run a test's fits as two concurrent chunks (fork / join over the handle's side stream) whatever the model's
    default is — one chunk for the SMPL-shaped model since round 4."""
    smplfit_env('SMPLFIT_CHUNKS', '2')
