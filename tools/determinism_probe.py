import sys, torch
sys.path.insert(0, '.')
from pointdsc_amd import PointDSC, workloads
for name, bs in (("n5000_b32", 32), ("n1000_b1", 1), ("lomatch_n10000_b8", 4)):
    w = workloads.WORKLOADS[name]
    m = PointDSC(**w["model"]); m.load_state_dict(workloads.state_dict(name, m.state_dict())); m = m.eval().cuda()
    b = workloads.batch(name, 0, bs)
    d = {k: b[k].cuda() for k in ("corr_pos", "src_keypts", "tgt_keypts")}; d["testing"] = True
    outs = []
    for fmt in ("f32", "u16"):
        m.compat_format = fmt
        rs = []
        for r in range(4):
            with torch.no_grad():
                o = m(d)
            rs.append((o["final_trans"].clone(), o["final_labels"].clone(),
                       m.workspace_view("featA", bs, w["num_corr"])[: bs * w["num_corr"] * 128].clone(),
                       m.workspace_view("seed_trans", bs, w["num_corr"])[: bs * int(w["num_corr"] * 0.1) * 16].clone()))
        same = [all(torch.equal(rs[0][i], r[i]) for r in rs[1:]) for i in range(4)]
        print(name, fmt, "bitwise repeatable: trans", same[0], "labels", same[1], "features", same[2], "seed_trans", same[3],
              "max dT across runs", max(float((rs[0][0] - r[0]).abs().max()) for r in rs[1:]))
