#!/bin/bash
# GPU box: shader clock / socket power while the LAFS step runs back to back
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-extras --steps 1500 --warmup 20 > /tmp/b.json 2>/dev/null &
PID=$!
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '; echo; sleep 1.5; done
wait $PID
tail -1 /tmp/b.json | grep -o '"ms_per_step": [0-9.]*'
