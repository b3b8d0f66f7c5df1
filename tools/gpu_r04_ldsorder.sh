#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_order.hip -o /tmp/lds_order && /tmp/lds_order | tee gpurun_out/r04/lds_atomic_order.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only tools/lds_atomic_order.hip -o - | grep -i "ds_min\|ds_cmpst\|ds_.*rtn" | head
