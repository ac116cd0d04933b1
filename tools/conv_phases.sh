#!/bin/bash
# builds libhat_stamps.so (hat_conv.hip with -DHAT_CONV_STAMPS, the other objects as shipped) and runs tools/conv_phases.py
set -eo pipefail
mkdir -p tools/bin gpurun_out
CS=super_resolution_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DHAT_CONV_STAMPS -Iinclude -c $CS/hat_conv.hip -o tools/bin/hat_conv_stamps.o
OBJS=$(ls $CS/*.o | grep -v "hat_conv.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libhat_stamps.so tools/bin/hat_conv_stamps.o $OBJS
HAT_MI355X_LIB=$PWD/tools/bin/libhat_stamps.so timeout -k 10 300 python tools/conv_phases.py | tee gpurun_out/conv_phases.txt
