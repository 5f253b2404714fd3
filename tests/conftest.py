import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def dev():
    import os
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    tile = os.environ.get('VM_TEST_GEMM_TILE')      # test plumbing only (the child run of test_gemm_bf16_suite_with_forced_big_tiles)
    if tile:
        from mmmm_amd import hip
        assert hip.lib().vm_gemm_force_tile_(int(tile)) == 0
    return torch.device('cuda:0')
