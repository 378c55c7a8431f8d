#!/bin/bash
# usage: tools/build_stamps_variant.sh   -> tools/experiments/libron_hip_stamps.so
# The product library with tools/experiments/kloop_clock_stamps.patch applied to a COPY of csrc (the tree is not touched): the four-wave
# tiles stamp s_memtime / s_memrealtime around their assembly K loop into a ring that `ron_debug_stamps(ptr)` hands to the library
# (tools/kloop_clock.py allocates and reads it).  Diagnostic only: RON_HIP_LIB=<that file> selects it; the shipped kernels execute no stamp.
set -e
cd "$(dirname "$0")/.."
C=ron_tensorflow_amd/csrc
W=/tmp/ron_stamps_build
rm -rf $W && mkdir -p $W && cp -r $C $W/b && rm -rf $W/b/build $W/b/build_asan
(cd $W && patch -p0 -s < "$OLDPWD/tools/experiments/kloop_clock_stamps.patch")
make -C $C -j8 > /dev/null            # the unpatched objects the variant links against
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -I."
R=$PWD
mkdir -p $W/o
(cd $W/b && sed -i "s#\"../../include/ron_hip.h\"#\"$R/include/ron_hip.h\"#" common.h; \
 /opt/rocm/bin/hipcc $FLAGS -c conv_mfma.hip -o $W/o/conv_mfma.o & \
 /opt/rocm/bin/hipcc $FLAGS -c conv_mfma4w.hip -o $W/o/conv_mfma4w.o & wait)
OBJS=$(ls $C/build/*.o | grep -v -e '/conv_mfma.o' -e '/conv_mfma4w.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/libron_hip_stamps.so $OBJS $W/o/conv_mfma.o $W/o/conv_mfma4w.o
ls -la tools/experiments/libron_hip_stamps.so
