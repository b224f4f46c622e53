#!/bin/bash
# Builds librwr with AddressSanitizer + UBSan on its HOST code (the gfx950 device code stays uninstrumented) into
# build/asan/ and links tests/cpp/abi_sanitize.cpp against it.  Runs in the build container (no GPU needed); the binaries
# travel to the GPU box with the snapshot, where tools/asan_run.sh executes them.  (Listed in .gpurunignore: the GPU box
# only runs the result.)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/asan
mkdir -p $out
for f in api build iterate spmv sweep small chain_scan rank sort; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero \
      -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
      -c $root/recommendersystems_amd/csrc/$f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-sanitize -fsanitize=address -fsanitize=undefined -o $out/librwr.so $out/*.o
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fno-gpu-sanitize -fsanitize=address -fsanitize=undefined -fno-omit-frame-pointer \
    $root/tests/cpp/abi_sanitize.cpp -o $out/abi_sanitize -L$out -lrwr -Wl,-rpath,'$ORIGIN'
ls -la $out/abi_sanitize
