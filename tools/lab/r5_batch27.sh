#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b27; mkdir -p $O; cd $R
for v in "" nohash; do echo "== variant '$v'"; LAFS_LIB_VARIANT=$v timeout 900 python tools/lab/t_big_ab.py 2>&1 | grep "^M=44160" | grep "GELU pair\|resid\|dgelu" | cut -c1-330; done
