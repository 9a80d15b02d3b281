"""Builds libdam_hip.so (the C-ABI library of include/dam_hip.h) for gfx950 with hipcc.

In-tree build: objects under csrc/_build/, the shared library next to this file, so it
travels with the repo snapshot to the GPU box.  hipcc cross-compiles without a GPU.
Staleness is decided by CONTENT: every object and library carries a `.sig` file with the
sha256 of the sources, headers and flags it was built from.
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


def _digest(paths, extra=()):
    """sha256 over the flags and the CONTENT of the given files (not their mtimes: a snapshot copied to another box, a
    checkout, a touch all change mtimes without changing what would be compiled -- and an edit restored from a backup
    changes content under an old mtime)."""
    import hashlib
    h = hashlib.sha256()
    for x in extra:
        h.update(str(x).encode() + b'\0')
    for p in paths:
        h.update(os.path.basename(p).encode() + b'\0')
        with open(p, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def _current(target, sig):
    """Was `target` built from exactly this signature?  (recorded in target + '.sig' after a successful build)"""
    try:
        with open(target + '.sig') as fh:
            return os.path.exists(target) and fh.read().strip() == sig
    except OSError:
        return False


def _record(target, sig):
    with open(target + '.sig', 'w') as fh:
        fh.write(sig + '\n')


def _compile(src, extra=(), suffix=''):
    """-> (object path, signature of what it was compiled from)"""
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + suffix + '.o')
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h'))
    headers.append(os.path.join(ROOT, 'include', 'dam_hip.h'))
    flags = [a for a in FLAGS if not os.path.isabs(a)] + list(extra)          # (include paths differ between boxes)
    sig = _digest([src] + headers, [HIPCC] + flags)
    if not _current(obj, sig):
        cmd = [HIPCC] + FLAGS + list(extra) + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
        _record(obj, sig)
    return obj, sig


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
        built = list(ex.map(_compile, srcs))
        diag_built = diag.result()
    diag_members = [diag_built if os.path.basename(o) == DIAG_SOURCE[:-4] + '.o' else (o, sig) for o, sig in built]
    for lib, members in ((LIB, built), (LIB_DIAG, diag_members)):
        import hashlib
        sig = hashlib.sha256(''.join(sig for _, sig in members).encode()).hexdigest()
        if force or not _current(lib, sig):
            cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + [o for o, _ in members]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
            _record(lib, sig)
    return LIB


if __name__ == '__main__':
    print(build_lib(force='--force' in sys.argv))
