"""Builds env_build_amd/lib/libenvbuild_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

-ffp-contract=off is part of the numerical contract: every fp32 op of the reference is one IEEE
rounding, so no FMA contraction.

Staleness is decided by CONTENT, not by mtime: the SHA-256 of every source, header and compiler flag is
written next to the library (libenvbuild_hip.so.srchash) when it is built, and `needs_build()` /
`check_fresh()` compare it with the hash of the sources present.  A library that was copied somewhere
together with different sources (e.g. a snapshot pushed to a GPU box) is therefore never silently reused:
it is rebuilt when hipcc is there, and `_capi.hip_api()` refuses to load it otherwise."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'lib', 'libenvbuild_hip.so')
HASH_FILE = LIB + '.srchash'
SOURCES = ['eb_capi.hip', 'eb_kernels.hip', 'eb_rollout.hip', 'eb_env_kernels.hip', 'eb_env_step.hip', 'eb_env_step_t1.hip', 'eb_env_step_t2.hip',
           'eb_policy.hip']   # (the env step's kernels: one translation unit per task — the three compile side by side)
HEADERS = ['eb_device.h', 'eb_kernels.h', 'eb_env_device.h', 'eb_env_step_body.h', os.path.join('..', '..', 'include', 'envbuild.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-fast-math',
         '-fPIC', '-Wno-unused-value', '-Wno-pass-failed',
         '-mllvm', '-amdgpu-kernarg-preload-count=14']   # the rollout kernel's leading arguments (14 dwords: all of FusedHot) arrive in SGPRs
LINK_FLAGS = ['--offload-arch=gfx950', '-shared', '-fPIC']


def source_hash():
    """SHA-256 over the compiler flags and the bytes of every source and header, in a fixed order."""
    h = hashlib.sha256()
    h.update('\0'.join(FLAGS + ['|'] + LINK_FLAGS).encode())
    for f in SOURCES + HEADERS:
        h.update(b'\0' + f.encode() + b'\0')
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


KERNEL_SOURCES = {   # what a kernel's machine code depends on (its translation unit and the headers it includes) — pmc_traffic.py
    'rollout': ['eb_rollout.hip', 'eb_device.h', 'eb_kernels.h'],             # records these next to the HBM bytes it measures,
    'env_step': ['eb_env_step_body.h', 'eb_env_step.hip', 'eb_env_device.h', 'eb_device.h', 'eb_kernels.h'],   # bench.py refuses the bytes of other code
}


def kernel_hash(which):
    """SHA-256 over the compiler flags and the sources of one kernel family ('rollout' / 'env_step')."""
    h = hashlib.sha256()
    h.update('\0'.join(FLAGS).encode())
    for f in KERNEL_SOURCES[which]:
        h.update(b'\0' + f.encode() + b'\0')
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def built_hash():
    try:
        with open(HASH_FILE) as fh:
            return fh.read().strip()
    except OSError:
        return None


def needs_build():
    return not os.path.isfile(LIB) or built_hash() != source_hash()


def check_fresh():
    """(ok, message): does the library on disk come from the sources on disk?"""
    if not os.path.isfile(LIB):
        return False, 'HIP extension missing: %s' % LIB
    if built_hash() != source_hash():
        return False, ('%s was built from different sources or flags than the ones in %s (hash %s, sources %s)'
                       % (LIB, CSRC, built_hash(), source_hash()))
    return True, ''


def build(force=False, verbose=False):
    """Compile the translation units in parallel (one hipcc per .hip file: a fresh GPU box builds the library in the time
    of its slowest file), then link; the library and its source hash are replaced atomically at the end."""
    if not force and not needs_build():
        return LIB
    import fcntl
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    # one builder at a time: the ranks of a multi-GPU job on a fresh box all arrive here; the first one compiles, the
    # others wait for the lock and find the library fresh
    with open(LIB + '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return LIB
        return _build_locked(verbose)


def _build_locked(verbose):
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    digest = source_hash()           # of what the compiler is about to read
    objdir = tempfile.mkdtemp(prefix='eb_build_', dir=os.path.dirname(LIB))
    try:
        def compile_one(f):
            obj = os.path.join(objdir, os.path.splitext(f)[0] + '.o')
            cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, f), '-o', obj]
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
            return obj
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        tmp = LIB + '.tmp%d' % os.getpid()
        cmd = [hipcc] + LINK_FLAGS + objs + ['-o', tmp]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
    with open(HASH_FILE, 'w') as fh:
        fh.write(digest + '\n')
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
