"""Build the in-tree gfx950 shared library with hipcc (cross-compiles without
a GPU).  `python -m dolfin_navier_scipy_amd.build`"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['dns_amd.hip']
LIB = os.path.join(CSRC, 'libdnsamd.so')


def dependencies():
    """every file the translation unit can include: all headers / .inc files
    next to it and the public header (derived, not hand-kept: a stale list once
    let an edit of trap_capi.inc ship an old library)"""
    deps = sorted(glob.glob(os.path.join(CSRC, '*.hpp'))
                  + glob.glob(os.path.join(CSRC, '*.inc'))
                  + glob.glob(os.path.join(CSRC, '*.hip')))
    deps.append(os.path.normpath(os.path.join(CSRC, '..', '..', 'include',
                                              'dns_amd.h')))
    return deps


def _hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found')


def needs_build():
    if not os.path.exists(LIB):
        return True
    libtime = os.path.getmtime(LIB)
    for path in dependencies():
        if os.path.getmtime(path) > libtime:
            return True
    return False


def build_library(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    # kernel arguments preloaded into SGPRs by the hardware (gfx950): the
    # first 16 dwords are there when a wave starts -- latency-bound kernels of
    # 5 us began with two to five dependent scalar loads of their arguments
    flags = os.environ.get('DNS_HIPCC_FLAGS',
                           '-mllvm -amdgpu-kernarg-preload-count=16').split()
    cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] \
        + flags + ['-shared', '-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES] \
        + ['-L/opt/rocm/lib', '-lrccl', '-lpthread',
           '-Wl,-rpath,/opt/rocm/lib']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == '__main__':
    build_library(force='--force' in sys.argv)
