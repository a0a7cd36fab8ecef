import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import correrender_amd as ca
from correrender_amd import Measure, synth
mode = sys.argv[1]
os.environ["CRF_GROUP_EXCHANGE"] = mode
xs, ys, zs, cs = 20, 12, 10, 24
ens = synth.box_ensemble(xs, ys, zs, cs, seed=21)
eng = ca.CorrField(0); eng.set_grid(xs, ys, zs, cs); eng.upload_members(ens)
rng = np.random.default_rng(5)
refs = [(int(rng.integers(0, xs)), int(rng.integers(0, ys)), int(rng.integers(0, zs))) for _ in range(70)]
bad_total = 0
with ca.CorrFieldGroup([0, 0, 0]) as grp:
    print(mode, grp.exchange)
    grp.set_grid(xs, ys, zs, cs); grp.upload_members(ens)
    for rep in range(12):
        for measure in (Measure.PEARSON, Measure.SPEARMAN, Measure.MUTUAL_INFORMATION_BINNED):
            want = [eng.compute(measure, r, k=2).reshape(-1) for r in refs]
            outs = [[torch.empty(xs * ys * grp.slab(s)[1], dtype=torch.float32, device="cuda") for s in range(3)] for _ in refs]
            grp.compute_batch_device(measure, refs, outs, k=2)
            for i, (r, row) in enumerate(zip(refs, outs)):
                got = torch.cat(row).cpu().numpy()
                d = np.nonzero(got.view(np.uint32) != want[i].view(np.uint32))[0]
                if d.size:
                    bad_total += 1
                    # does the wrong data equal another evaluation's result?
                    src = [j for j in range(len(refs)) if np.array_equal(got[d], want[j][d])]
                    print(f"rep {rep} {measure.name} eval {i} ref {r}: {d.size} differ, idx {d[:6]}..{d[-3:]}, equals evaluation(s) {src[:5]} (owner z slab of this ref: {r[2]})")
print("mode", mode, "bad evaluations:", bad_total)
