#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b33; mkdir -p $O; cd $R
bash tools/lab/pmc_mem.sh tools/lab/t_big_pmc.py 2>&1 | grep "gemm_big\|gemm_nt\|no counters" > $O/mem.txt; cat $O/mem.txt | cut -c1-330
