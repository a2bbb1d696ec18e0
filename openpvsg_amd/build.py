"""In-tree build of the gfx950 HIP library (no JIT cache: the .so must travel with the tree).

`python -m openpvsg_amd.build` or `__graft_entry__.build()` compiles every `csrc/*.hip` with
`hipcc --offload-arch=gfx950` and links `openpvsg_amd/lib/libopenpvsg_hip.so`.
hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(LIBDIR, 'obj')
LIBNAME = 'libopenpvsg_hip.so'
ARCH = 'gfx950'
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
CFLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC',
          '-Wall', '-Wno-unused-function']
# per-file additions.  winograd3x3: packed-f32 VALU (v_pk_add_f32 built by the SLP vectoriser, with the v_movs that feed
# it) beside f32 MFMAs costs issue slots the scalar adds do not.
EXTRA_CFLAGS = {'winograd3x3.hip': ['-fno-slp-vectorize']}


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip_lib(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src[:-4] + '.o')
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + CFLAGS + EXTRA_CFLAGS.get(os.path.basename(s), []) + ['-c', s, '-o', o]
        if verbose:
            print('[build]', ' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (s, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    # objects of sources that no longer exist: drop them and relink, or their symbols live on in the library
    orphans = [os.path.join(OBJDIR, f) for f in os.listdir(OBJDIR) if f.endswith('.o') and os.path.join(OBJDIR, f) not in objs]
    for o in orphans:
        os.remove(o)
    out = lib_path()
    if force or jobs or orphans or _stale(out, objs):
        cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC'] + objs + ['-o', out]
        if verbose:
            print('[build]', ' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    return out


if __name__ == '__main__':
    print(build_hip_lib(force='--force' in sys.argv))
