#!/bin/bash
# usage: tools/build_kloop_variant.sh <name> <KLOOP_OPTS>   -> tools/experiments/libron_hip_<name>.so (the product library with another
# K-loop schedule from tools/gen_kloop4w.py; csrc/kloop4w.inc is restored afterwards).  RON_HIP_LIB=<that file> selects it.
set -e
cd "$(dirname "$0")/.."
NAME=$1; OPTS=$2
C=ron_tensorflow_amd/csrc
cp $C/kloop4w.inc /tmp/kloop4w.inc.keep
KLOOP_OPTS=$OPTS python3 tools/gen_kloop4w.py > /dev/null
mkdir -p /tmp/kv_$NAME
(cd $C && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -I. -c conv_mfma.hip -o /tmp/kv_$NAME/conv_mfma.o 2>/dev/null)
cp /tmp/kloop4w.inc.keep $C/kloop4w.inc
OBJS=$(ls $C/build/*.o | grep -v conv_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/experiments/libron_hip_$NAME.so $OBJS /tmp/kv_$NAME/conv_mfma.o
ls -la tools/experiments/libron_hip_$NAME.so
