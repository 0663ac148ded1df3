#!/bin/bash
# GPU box, one gpurun call: everything profiles/round<N>_* is condensed from (tools/publish_profiles.py round<N> afterwards).
#   step (graph, two streams) kernel stats | step serialised | roofline kernels alone: stats + FETCH/WRITE/SQ passes |
#   HBM bytes of one step | SQ counters of every GEMM / weight-gradient / attention instance
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
# fingerprint of the kernel sources these profiles are measured on (bench.py prints read-back numbers only while it matches)
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.csrc_fingerprint())" > $O/csrc_sha.txt
rocprofv3 --kernel-trace --stats -d $O/step -o step --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline > $O/step_bench.json 2> $O/step.err
rocprofv3 --kernel-trace --stats -d $O/dom -o dom --output-format csv -- python3 $R/bench.py --roofline-only > $O/dom_bench.json 2> $O/dom.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --roofline-only > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 $R/bench.py --roofline-only > /dev/null 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq -o s --output-format csv -- python3 $R/bench.py --roofline-only > /dev/null 2> $O/pmc_sq.err
bash $R/tools/profile_serial.sh > $R/gpurun_out/serial.log 2>&1
bash $R/tools/profile_step_hbm.sh > $R/gpurun_out/stephbm.log 2>&1
bash $R/tools/profile_gemm.sh > $R/gpurun_out/gemm.log 2>&1
bash $R/tools/profile_attn.sh > $R/gpurun_out/attn.log 2>&1
tail -3 $R/gpurun_out/serial.log; tail -4 $R/gpurun_out/stephbm.log | cut -c1-600; tail -30 $R/gpurun_out/gemm.log; tail -6 $R/gpurun_out/attn.log
cat $O/step_bench.json | cut -c1-400; cat $O/dom_bench.json | cut -c1-1200
