"""TEST INFRASTRUCTURE ONLY -- builds tests/emu/_build/libmpqe_emu.so: the kernel sources
of mpqe_amd/csrc compiled for the HOST against the fiber emulator (tests/emu/include),
so kernel logic can be checked without a GPU. Never loaded by the product."""
import fcntl
import glob
import os
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# MPQE_EMU_EXPERIMENTS=1: the variant with the launch forms that were taken out of the shipped library (csrc/common.h:
# MPQE_EXPERIMENTS) -- their tests in tests/test_step.py run against it and skip otherwise
EXPERIMENTS = os.environ.get('MPQE_EMU_EXPERIMENTS', '') not in ('', '0')
OUT = os.path.join(HERE, '_build', 'libmpqe_emu_exp.so' if EXPERIMENTS else 'libmpqe_emu.so')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'


def sources():
    return sorted(glob.glob(os.path.join(ROOT, 'mpqe_amd', 'csrc', '*.hip')))


def build_emu(force=False):
    # a prebuilt variant (e.g. the AddressSanitizer build of tools/emu_asan.sh) can be injected
    if os.environ.get('MPQE_EMU_LIB'):
        return os.environ['MPQE_EMU_LIB']
    deps = sources() + glob.glob(os.path.join(ROOT, 'mpqe_amd', 'csrc', '*.h')) + \
        glob.glob(os.path.join(HERE, 'include', '*', '*.h*')) + \
        glob.glob(os.path.join(HERE, 'include', '*', '*', '*.h*')) + \
        [os.path.join(HERE, 'emu_runtime.cpp'), os.path.join(ROOT, 'include', 'mpqe_amd.h')]
    def fresh():
        return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)
    if not force and fresh():
        return OUT
    if not os.path.exists(CLANG):
        raise RuntimeError('clang++ not found at %s' % CLANG)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT + '.lock', 'w') as lock:            # pytest-xdist workers build at most once
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not fresh():
            tmp = OUT + '.tmp.%d' % os.getpid()
            started = time.time()
            cmd = [CLANG, '-x', 'c++', '-std=c++17', '-O1', '-g', '-fPIC', '-shared'] + \
                (['-DMPQE_EXPERIMENTS'] if EXPERIMENTS else []) + \
                ['-I' + os.path.join(HERE, 'include'), '-I' + os.path.join(ROOT, 'include')] + \
                sources() + [os.path.join(HERE, 'emu_runtime.cpp'), '-o', tmp]
            subprocess.check_call(cmd)
            os.utime(tmp, (started, started))       # a source edited while this ran is newer than the result
            os.replace(tmp, OUT)
    return OUT


if __name__ == '__main__':
    print(build_emu(force=True))
