cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/snip.py
import re
s=open('tests/test_gpu_step.py').read()
m=re.search(r"_CHAINS_SNIPPET = r'''(.*?)'''", s, re.S)
print(m.group(1))
PY
for lib in "" oldln; do for mode in 2 2 2 2 2 2 0 0 0; do
  LAFS_LIB_VARIANT=$lib LAFS_ROW_CHAINS=$mode LAFS_MLP_FUSED=${MF:-15} PYTHONPATH=. python gpurun_out/snip.py gpurun_out/o.pt > /dev/null 2>&1
  python - <<PY
import torch
d=torch.load("gpurun_out/o.pt")
print("lib=[$lib] chains=$mode losses", ["%.7f"%x for x in d["losses"]])
PY
done; done
rm -f gpurun_out/o.pt
