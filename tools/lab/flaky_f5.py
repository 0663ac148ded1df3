import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from conftest import load_golden, sub
import test_gpu_step as T
fx = load_golden("f5_lafs_step")
lrs, wds, moms = fx["hyper"].tolist()
bad_runs = 0
for rep in range(30):
    student, teacher, crit, eng = T._build(fx, True)
    tt = crit.teacher_temp_schedule
    for s in range(2):
        crops = [fx[f"s{s}.crop{i}"] for i in range(5)]
        eng.step(crops, lr=lrs[s], wd=wds[s], momentum=moms[s], teacher_temp=float(tt[s]), epoch=s)
        torch.cuda.synchronize()
        post = sub(fx, f"s{s}.grad_post.")
        norms = dict(zip([str(n) for n in fx["norm_names"]], fx[f"s{s}.norms"].tolist()))
        gerr = {}
        for k, g in post.items():
            mine = dict(student.named_parameters())[k].grad
            clip = min(1.0, 3.0 / (norms[k] + 1e-6))
            if float(g.abs().max()) > 1e-6:
                gerr[k] = T.rel_l2(mine * clip, g)
        kk = max(gerr, key=gerr.get)
        ss = eng.sa.seg_sumsq.cpu()
        refn = torch.tensor([norms[n] for n in eng.sa.names if n in norms])
        mine_n = torch.tensor([float(ss[eng.sa.names.index(n)]) ** 0.5 for n in eng.sa.names if n in norms])
        nerr = float(((mine_n - refn).abs() / (refn + 1e-9)).max())
        badseg = [(n, float(ss[eng.sa.names.index(n)])) for n in eng.sa.names if n in norms and abs(float(ss[eng.sa.names.index(n)]) ** 0.5 - norms[n]) > 0.1 * norms[n] + 1e-6]
        gmax = {n: float(eng.sa.view(eng.sa.grad, n).abs().max()) for n, _ in badseg}
        lossv = float(eng.loss.item())
        worst = []
        for k, v in sub(fx, f"s{s}.student.").items():
            e = (student.state_dict()[k].cpu().double() - v.double()).abs().flatten().numpy()
            worst.append((float(np.quantile(e, 0.9)) / lrs[s], k))
        allq = np.quantile(np.concatenate([(student.state_dict()[k].cpu().double() - v.double()).abs().flatten().numpy() for k, v in sub(fx, f"s{s}.student.").items()]), 0.9) / lrs[s]
        if allq > 0.6:
            bad_runs += 1
            worst.sort(reverse=True)
            print("   bad segments:", badseg[:6], "grad absmax:", gmax); print(f"   norm err {nerr:.3e} nan {bool(torch.isnan(ss).any())} sumsq[:4] {ss[:4].tolist()}"); print(f"   loss {lossv:.5f} ref {float(fx[f's{s}.loss']):.5f}; worst grad err {gerr[kk]:.3e} at {kk}; seg_step {eng.sa.seg_step[:4].tolist()} hyper {eng.hyper[:11].tolist()}")
            print(f"rep {rep} step {s}: overall q90 = {allq:.3f} lr; worst tensors:", [(round(a, 2), k) for a, k in worst[:6]])
    # garbage in freed memory for the next repetition
    junk = torch.full((64 << 20,), float("nan"), device="cuda"); del junk
print("bad runs:", bad_runs, "of 40")
