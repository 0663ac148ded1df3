#!/bin/bash
# compile one csrc/*.hip with the resource-usage remarks (VGPRs / spills per kernel), e.g. tools/lab/cc.sh gemm_kpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I../../include -Wall -Wno-unused-function ${EXTRA} -c /root/repo/lafs_cvpr2024_amd/csrc/$1.hip -I/root/repo/include -o /root/repo/lafs_cvpr2024_amd/csrc/build/$1.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning|Function Name|VGPRs:|Spill" | sed 's/.*remark: *//' | paste - - - - | sed 's/\[-Rpass[^]]*\]//g'
