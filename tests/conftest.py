import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE_DIR = os.path.join(ROOT, "oracle")
if ORACLE_DIR not in sys.path:
    sys.path.insert(0, ORACLE_DIR)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# shared by test_gpu_parity.py and test_zz_gpu_multiprocess.py (the latter is named to be collected LAST: the tests that start child
# processes / torch.distributed.run must never stand between the suite and the kernel / model parity tests)
@pytest.fixture(scope="session")
def lib():
    import torch
    from manipose_amd import _lib
    assert torch.cuda.is_available(), "the gpu-marked tests need an MI355X"
    return _lib.load()


@pytest.fixture(scope="session")
def raw_dataset_dir(tmp_path_factory):
    import numpy as np
    from helpers import write_raw_dataset_files
    d = tmp_path_factory.mktemp("raw_datasets")
    fx = np.load(GOLDEN + "/datasets.npz")
    write_raw_dataset_files(str(d), fx)
    return str(d), fx
