#!/bin/bash
# build experiment variants of the Huffman kernels: tools/huf_variants.sh NAME "-DFLAG ..." -> sprintz_amd/ab/libNAME.so
cd $(dirname $0)/../sprintz_amd/csrc
mkdir -p ../ab build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -c huf.hip -o build/huf_$1.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/lib$1.so build/api.o build/huf_$1.o build/decode_w8.o build/decode_w16.o build/encode_w8.o build/encode_w16.o
