#!/bin/bash
# usage: tools/build_kbench.sh <name> [extra hipcc flags...]   -> tools/kb/<name>
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/kb
n=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Wno-unused-function -Wno-unused-value "$@" tools/kbench.cpp -o tools/kb/$n
