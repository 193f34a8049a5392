#!/bin/bash
# Kernel logic under AddressSanitizer, on the CPU (GPU sanitizer runs are not available on this pool): the emulator build
# of the kernel sources (tests/emu) compiled with -fsanitize=address, loaded by the tests through MPQE_EMU_LIB. The test
# arrays are numpy buffers from the sanitizer's malloc, so a kernel that indexes outside an operand is reported with the
# kernel's source line. Fibers switch with ucontext here (-DEMU_UCONTEXT: the sanitizer knows swapcontext), ~10x slower
# than the default emulator build: pick tests with -k.
#   tools/emu_asan.sh tests/test_kernels.py -k "general or dense"
#   tools/emu_asan.sh tests/test_step.py -k "learned_readout and mlp-add"
set -e
root=$(cd $(dirname $0)/.. && pwd)
CL=/opt/rocm/lib/llvm/bin/clang++
rt=$($CL -print-file-name=libclang_rt.asan-x86_64.so)
out=${TMPDIR:-/tmp}/mpqe_emu_asan
mkdir -p $out
lib=$out/libmpqe_emu_asan.so
newest=$(ls -t $root/mpqe_amd/csrc/*.hip $root/mpqe_amd/csrc/*.h $root/tests/emu/emu_runtime.cpp $root/include/mpqe_amd.h | head -1)
if [ ! -f $lib ] || [ $newest -nt $lib ]; then
    $CL -x c++ -std=c++17 -O1 -g -fPIC -shared -fsanitize=address -DEMU_UCONTEXT -I$root/tests/emu/include -I$root/include \
        $root/mpqe_amd/csrc/*.hip $root/tests/emu/emu_runtime.cpp -o $lib
fi
cd $root
LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 MPQE_EMU_LIB=$lib \
    python -m pytest -x -q -s -m "not gpu" -p no:cacheprovider "$@"      # (-s: a report must not die in pytest's capture)
