#!/bin/bash
# round 6, one gpurun call: collect the profiles, condense them on the box (so that the bench below reads traffic measured on THIS
# source fingerprint), smoke, default bench.  Afterwards, in the build container: python tools/publish_profiles.py round6 (same inputs).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash tools/collect_profiles_round.sh > gpurun_out/r6_collect.log 2>&1
python tools/publish_profiles.py round6 > gpurun_out/r6_publish.log 2>&1
bash tools/lab/r6_smoke_bench.sh
