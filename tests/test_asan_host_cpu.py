"""Host-side AddressSanitizer pass over the C ABI (CPU only: GPU ASan / xnack+ code objects are not available on this pool).
tools/asan_host.sh compiles the GEMM / row-wise / LoRA translation units with -fsanitize=address for the host and runs
tests/asan/abi_host_asan.cpp: argument validation of every entry point that validates before it launches, workspace helpers and the
event-profiling bookkeeping — no GPU needed, every call returns before a kernel launch."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.timeout(900)
def test_host_code_of_the_c_abi_is_clean_under_address_sanitizer(tmp_path):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not Path(hipcc).exists() or shutil.which('bash') is None:
        pytest.skip('no hipcc')
    r = subprocess.run(['bash', str(ROOT / 'tools' / 'asan_host.sh'), str(tmp_path)], capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert 'OK: 0 failed checks' in r.stdout
    assert 'AddressSanitizer' not in r.stderr
    libs = subprocess.run(['ldd', str(tmp_path / 'abi_host_asan')], capture_output=True, text=True).stdout
    assert 'asan' in libs, 'the driver was not linked against the ASan runtime'
