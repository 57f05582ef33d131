#!/bin/bash
# build_ab.sh NAME SRC "FLAGS": a copy of libkx_hip.so with csrc/SRC.hip rebuilt under extra flags,
# as lib/ab/libkx_hip_NAME.so (for A/B runs through KX_HIP_LIB)
set -e
cd "$(dirname "$0")/../llm-inference-engine_amd"
mkdir -p lib/ab build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../include $3 -c csrc/$2.hip -o build/ab/$2_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o lib/ab/libkx_hip_$1.so $(ls build/*.o | grep -v "build/$2.o") build/ab/$2_$1.o -ldl
