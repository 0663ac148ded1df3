#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats for the whole benchmark and for the dominant kernel alone, plus HBM PMC passes.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/step -o step --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/step_bench.json 2> $O/step.err
rocprofv3 --kernel-trace --stats -d $O/dom -o dom --output-format csv -- python3 $R/bench.py --roofline-only > $O/dom_bench.json 2> $O/dom.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --roofline-only > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 $R/bench.py --roofline-only > /dev/null 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq -o s --output-format csv -- python3 $R/bench.py --roofline-only > /dev/null 2> $O/pmc_sq.err
ls -R $O | head -40
