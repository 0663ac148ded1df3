#!/bin/bash
# the driver's round-end sequence on one box: GPU tests, smoke, default bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log | cut -c1-200
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<PY
import json
for l in open("$O/bench.json"):
    if l.strip().startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print({k:d[k] for k in ("metric","value","unit","ms_per_step","steps","warmup","n_gpus","dtype","scaling","vs_baseline")})
        print("roofline", {k:r[k] for k in ("bound","achieved","peak","unit","frac","traffic","family")}, "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
        print("extras", {k:v.get("ms_per_step") for k,v in d.get("extras",{}).items() if isinstance(v,dict)})
        print("provenance used:", {k:v.get("used") for k,v in d.get("profile_provenance",{}).items()})
PY
