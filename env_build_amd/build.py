"""Builds env_build_amd/lib/libenvbuild_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

-ffp-contract=off is part of the numerical contract: every fp32 op of the reference is one IEEE
rounding, so no FMA contraction."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'lib', 'libenvbuild_hip.so')
SOURCES = ['eb_capi.hip', 'eb_kernels.hip', 'eb_rollout.hip', 'eb_env_kernels.hip', 'eb_policy.hip']
HEADERS = ['eb_device.h', 'eb_kernels.h', os.path.join('..', '..', 'include', 'envbuild.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-fast-math',
         '-fPIC', '-shared', '-Wno-unused-value', '-Wno-pass-failed',
         '-mllvm', '-amdgpu-kernarg-preload-count=12']   # the rollout kernel's leading arguments arrive in SGPRs


def needs_build():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, f) for f in SOURCES] + ['-o', LIB]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
