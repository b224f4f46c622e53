#!/bin/bash
# Builds and runs tools/eval_graphs_threads.cpp on the GPU box:  bash tools/eval_graphs_threads.sh [graphs per call] [calls per thread]
set -e
root=$(pwd)
g++ -O2 -std=c++17 -pthread -I$root/include $root/tools/eval_graphs_threads.cpp -o /tmp/egt $root/recommendersystems_amd/librwr.so -Wl,-rpath,$root/recommendersystems_amd
/tmp/egt "$@"
