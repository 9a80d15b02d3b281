"""Builds libdam_hip.so (the C-ABI library of include/dam_hip.h) for gfx950 with hipcc.

In-tree build: objects under csrc/_build/, the shared library next to this file, so it
travels with the repo snapshot to the GPU box.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_build')
LIB = os.path.join(HERE, 'libdam_hip.so')
# the same library with dam_conv_strip.hip's device-side check of its geometry tables compiled in (tests/test_strip_diag_gpu.py)
LIB_DIAG = os.path.join(HERE, 'libdam_hip_diag.so')
DIAG_SOURCE, DIAG_FLAGS = 'dam_conv_strip.hip', ['-DDAM_STRIP_DIAG_TAGS']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function', '-mllvm', '-amdgpu-mfma-vgpr-form=1',
         '-I', os.path.join(ROOT, 'include'), '-I', CSRC]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, extra=(), suffix=''):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + suffix + '.o')
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(ROOT, 'include', 'dam_hip.h'))
    if _stale(obj, [src] + headers):
        cmd = [HIPCC] + FLAGS + list(extra) + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build_lib(force=False, jobs=None):
    """Compile every csrc/*.hip for gfx950 and link libdam_hip.so.  Returns its path."""
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = sources()
    diag_src = os.path.join(CSRC, DIAG_SOURCE)
    with ThreadPoolExecutor(max_workers=jobs or min(8, len(srcs) + 1)) as ex:
        diag = ex.submit(_compile, diag_src, DIAG_FLAGS, '.diag')
        objs = list(ex.map(_compile, srcs))
        diag_obj = diag.result()
    for lib, members in ((LIB, objs), (LIB_DIAG, [diag_obj if os.path.basename(o) == DIAG_SOURCE[:-4] + '.o' else o for o in objs])):
        if force or _stale(lib, members):
            cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + members
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    return LIB


if __name__ == '__main__':
    print(build_lib(force='--force' in sys.argv))
