#!/bin/bash
# Runs the host-sanitized C-ABI driver built by tools/asan_build.sh (build/asan/) on this box's GPU.
root=$(cd "$(dirname "$0")/.." && pwd)
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 $root/build/asan/abi_sanitize
