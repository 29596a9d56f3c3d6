#!/bin/bash
# Builds the library variant whose staged epilogue masks out-of-image lanes by an out-of-range buffer offset (round 5's bug) into
# tools/bin/libpocr_masked.so.  Run on the GPU box:  POCR_TMP_LIB=tools/bin/libpocr_masked.so python tools/three_in_flight.py 6 [--prealloc]
cd "$(dirname "$0")/.." && mkdir -p tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -DPOCR_EPI_MASKED_STORE=1 -I include \
    -o tools/bin/libpocr_masked.so pero_ocr_amd/csrc/pocr_hip.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/masked_store_probe tools/masked_store_probe.hip
