"""Build recipe for libvividmed_hip.so (gfx950 only).

`python -m mmmm_amd.build` cross-compiles every HIP translation unit under mmmm_amd/csrc with
hipcc --offload-arch=gfx950 and links them into mmmm_amd/lib/libvividmed_hip.so (in-tree, so the
built library travels with the repository snapshot to the GPU box).
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
from pathlib import Path
import subprocess
import sys

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / 'csrc'
LIB_DIR = ROOT / 'lib'
LIB_PATH = LIB_DIR / 'libvividmed_hip.so'
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
# VM_BUILD_DEFINES: extra -D flags of diagnostic builds (VM_GEMM_DEBUG_BUILD: the GEMM timing-experiment branches, tools/README.md)
CFLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function', '-DVM_KEEP_DENORMS',
          *os.environ.get('VM_BUILD_DEFINES', '').split()]


def _sources() -> list[Path]:
    return sorted(CSRC.glob('*.hip'))


def _digest(src: Path) -> str:
    h = hashlib.sha256()
    h.update(src.read_bytes())
    for hdr in sorted(CSRC.glob('*.hpp')) + [ROOT.parent / 'include' / 'vividmed_hip.h']:
        h.update(hdr.read_bytes())
    h.update(' '.join(CFLAGS).encode())
    return h.hexdigest()


def _compile(src: Path, obj: Path) -> None:
    stamp = obj.with_suffix('.sha')
    dig = _digest(src)
    if obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return
    cmd = [HIPCC, *CFLAGS, '-c', str(src), '-o', str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    stamp.write_text(dig)


def build(verbose: bool = True) -> Path:
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = LIB_DIR / 'obj'
    obj_dir.mkdir(exist_ok=True)
    srcs = _sources()
    objs = [obj_dir / (s.stem + '.o') for s in srcs]
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        list(ex.map(lambda a: _compile(*a), zip(srcs, objs)))
    newest = max(o.stat().st_mtime for o in objs)
    if not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < newest:
        cmd = [HIPCC, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', str(LIB_PATH), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    for tmp in LIB_DIR.glob(LIB_PATH.name + '.*'):        # hipcc leaves its offload-bundle temporaries beside the output
        tmp.unlink()
    if verbose:
        print(f'built {LIB_PATH} from {len(srcs)} sources')
    return LIB_PATH


if __name__ == '__main__':
    build()
