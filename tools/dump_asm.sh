#!/bin/bash
# usage: tools/dump_asm.sh <mangled-name-substring>   -> /tmp/t/kernel.s (one kernel) + resource usage
mkdir -p /tmp/t && cd /root/repo/ron_tensorflow_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S conv_mfma.hip -o /tmp/t/conv_dev.s 2>/dev/null
awk "/^_ZN.*$1.*:/,/s_endpgm/" /tmp/t/conv_dev.s > /tmp/t/kernel.s
wc -l /tmp/t/kernel.s
grep -E "vgpr_count|sgpr_count|lds_size|vgpr_spill" /tmp/t/conv_dev.s | head -0
