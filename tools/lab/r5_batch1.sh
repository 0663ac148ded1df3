#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/b1; mkdir -p $O; cd $R
timeout 300 tools/lab/repro_graph_events static > $O/repro_static.log 2>&1; echo "repro static rc=$?"; tail -2 $O/repro_static.log
timeout 300 tools/lab/repro_graph_events fresh > $O/repro_fresh.log 2>&1; echo "repro fresh rc=$?"; tail -2 $O/repro_fresh.log
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log | cut -c1-400
timeout 900 python -X faulthandler -m pytest tests/test_gpu_step.py tests/test_gpu_composition.py -x -q -p no:cacheprovider > $O/order.log 2>&1; echo "order rc=$?"; tail -3 $O/order.log | cut -c1-300
