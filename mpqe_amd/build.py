"""Builds mpqe_amd/lib/libmpqe_amd.so: every .hip source under mpqe_amd/csrc compiled by
hipcc for gfx950 (MI355X) only and linked into one C-ABI shared library. hipcc
cross-compiles without a GPU, so this also runs in the authoring container.
    python -m mpqe_amd.build [--force]
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = os.path.join(PKG, 'csrc')
LIB_DIR = os.path.join(PKG, 'lib')
LIB = os.path.join(LIB_DIR, 'libmpqe_amd.so')
OBJ_DIR = os.path.join(PKG, 'lib', 'obj')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-fPIC', '-std=c++17', '-Wno-unused-result',
         '-I' + os.path.join(ROOT, 'include')]
# per-source extras. step.hip: MFMA accumulators in VGPRs (gfx90a+ allows it; 167 VGPRs leave the room) -- with
# AGPR accumulators hipcc rotates the chain kernel's 16 accumulator registers through VGPRs at the top of
# every K-loop item and reads them back one by one in every node-update epilogue
EXTRA = {'step.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1']}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(SRC, '*.hip')))
    hdrs = glob.glob(os.path.join(SRC, '*.h')) + [os.path.join(ROOT, 'include', 'mpqe_amd.h')]
    if not srcs:
        raise RuntimeError('no HIP sources under %s' % SRC)
    if not os.path.exists(HIPCC):
        raise RuntimeError('hipcc not found at %s' % HIPCC)
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ_DIR, os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        if force or _stale(o, [s, os.path.abspath(__file__)] + hdrs):
            jobs.append([HIPCC] + FLAGS + EXTRA.get(os.path.basename(s), []) + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC'] + objs + ['-o', LIB])
    build_pyhost(force, verbose)
    build_autograd_node(force, verbose)
    return LIB


def autograd_node_path():
    import sysconfig
    return os.path.join(LIB_DIR, '_autograd_node' + (sysconfig.get_config_var('EXT_SUFFIX') or '.so'))


def build_autograd_node(force=False, verbose=False):
    """The drop-in margin_loss call's autograd node as a C++ torch::autograd::Node (csrc/host/autograd_node.cpp): g++ against
    the installed torch's headers and libraries, no device code. Returns its path."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension
    src = os.path.join(SRC, 'host', 'autograd_node.cpp')
    out = autograd_node_path()
    os.makedirs(LIB_DIR, exist_ok=True)
    if force or _stale(out, [src, os.path.abspath(__file__)]):
        libdir = os.path.join(os.path.dirname(torch.__file__), 'lib')
        cmd = [os.environ.get('CXX', 'g++'), '-O2', '-shared', '-fPIC', '-std=c++17', '-DTORCH_EXTENSION_NAME=_autograd_node',
               '-DTORCH_API_INCLUDE_EXTENSION_H', '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch.compiled_with_cxx11_abi())] + \
            ['-I' + i for i in cpp_extension.include_paths()] + ['-I' + sysconfig.get_paths()['include'], src, '-o', out,
                                                                 '-L' + libdir, '-ltorch', '-ltorch_cpu', '-lc10', '-ltorch_python',
                                                                 '-Wl,-rpath,' + libdir]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return out


def pyhost_path():
    import sysconfig
    return os.path.join(LIB_DIR, '_pyhost' + (sysconfig.get_config_var('EXT_SUFFIX') or '.so'))


def build_pyhost(force=False, verbose=False):
    """The CPython extension of the drop-in entry points' host path (csrc/host/pyhost.c): plain C against Python.h, no
    device code. Returns its path."""
    import sysconfig
    src = os.path.join(SRC, 'host', 'pyhost.c')
    out = pyhost_path()
    os.makedirs(LIB_DIR, exist_ok=True)
    if force or _stale(out, [src, os.path.join(ROOT, 'include', 'mpqe_amd.h'), os.path.abspath(__file__)]):
        cmd = [os.environ.get('CC', 'gcc'), '-O2', '-shared', '-fPIC', '-Wall', '-I' + sysconfig.get_paths()['include'],
               '-I' + os.path.join(ROOT, 'include'), src, '-o', out]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
