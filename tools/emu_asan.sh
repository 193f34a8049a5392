#!/bin/bash
# Kernel logic under a sanitizer, on the CPU (GPU sanitizer runs are not available on this pool): the emulator build of
# the kernel sources (tests/emu) compiled with -fsanitize=address (default) or =undefined (MPQE_SAN=undefined), loaded by
# the tests through MPQE_EMU_LIB. The test arrays are numpy buffers from the sanitizer's malloc, so a kernel -- or the host
# planner -- that indexes outside an operand is reported with its source line. ~3x slower than the default emulator build.
#   tools/emu_asan.sh tests/test_kernels.py -k "general or dense"
#   MPQE_SAN=undefined tools/emu_asan.sh tests/test_step.py -k "learned_readout and mlp-add"
set -e
root=$(cd $(dirname $0)/.. && pwd)
CL=/opt/rocm/lib/llvm/bin/clang++
san=${MPQE_SAN:-address}
if [ $san = address ]; then
    flags="-fsanitize=address"        # (the fibers' stack switches are announced to the sanitizer: emu_runtime.cpp)
    rt=$($CL -print-file-name=libclang_rt.asan-x86_64.so)
    export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0
else
    flags="-fsanitize=undefined -fno-sanitize=vptr,function"
    rt=$($CL -print-file-name=libclang_rt.ubsan_standalone-x86_64.so)
    export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
fi
out=${TMPDIR:-/tmp}/mpqe_emu_$san
mkdir -p $out
lib=$out/libmpqe_emu_$san.so
newest=$(ls -t $root/mpqe_amd/csrc/*.hip $root/mpqe_amd/csrc/*.h $root/tests/emu/emu_runtime.cpp $root/include/mpqe_amd.h | head -1)
if [ ! -f $lib ] || [ $newest -nt $lib ]; then
    $CL -x c++ -std=c++17 -O1 -g -fPIC -shared $flags -I$root/tests/emu/include -I$root/include \
        $root/mpqe_amd/csrc/*.hip $root/tests/emu/emu_runtime.cpp -o $lib
fi
cd $root
# (-s: a report must not die in pytest's capture)
LD_PRELOAD=$rt MPQE_EMU_LIB=$lib python -m pytest -x -q -s -m "not gpu" -p no:cacheprovider "$@"
