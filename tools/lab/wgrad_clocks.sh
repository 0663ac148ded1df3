#!/bin/bash
# GPU box: shader clock / power while the grouped weight gradient runs back to back (is the sustained rate a power-capped clock?)
cd $GRAFT_REPO_ROOT; export LAB_SHORT=1 LAB_ITERS=20000
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | head -8
tools/lab/lab_wgrad 2>/dev/null &
PID=$!
sleep 1.0
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -6; sleep 0.7; done
wait $PID
