#!/bin/bash
# round 6: GPU suite, then (only if green) everything profiles/round6_* is condensed from
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 ) > gpurun_out/r6_tests.txt 2>&1
cat gpurun_out/r6_tests.txt
grep -q " passed" gpurun_out/r6_tests.txt && ! grep -q "failed\|error" gpurun_out/r6_tests.txt || exit 1
bash tools/collect_profiles_round.sh
