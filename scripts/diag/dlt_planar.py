"""GPU DLT on the planar-pattern fixture: the problems that differ from the reference run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import pnp_solver_test_amd as pnp  # noqa: E402
from conftest import load_golden, rot_angle  # noqa: E402
from oracle import pnp_oracle as po  # noqa: E402

np.set_printoptions(precision=6, suppress=True, linewidth=160)
K = po.K_DEFAULT
dev = torch.device("cuda:0")
g = load_golden("dlt_planar")
out = pnp.solve_pnp_batch(torch.from_numpy(g["uv"].astype(np.float64)).to(dev), torch.from_numpy(g["xyz"]).to(dev), K,
                          method="dlt")
R = out["R"].cpu().numpy()
ang = rot_angle(R, g["R"])
bad = np.flatnonzero(~(ang <= 1e-7))
print("bad", len(bad), "by kind", np.bincount(g["kind_id"][bad], minlength=3))
for i in bad[:8]:
    q = po.dlt_solve(g["uv"][i].astype(np.float64), g["xyz"][i], K)
    print("i", i, "kind", g["kinds"][g["kind_id"][i]], "ang", ang[i], "k2", q["min_abs_k2"], "sim", q["min_abs_sim"], "delta", q["min_abs_delta"])
    print(" gpu\n", R[i], "\n ref\n", g["R"][i], "\n t gpu", out["t"][i].cpu().numpy(), "ref", g["t"][i])
    for rounds in (1, 2):
        o1 = pnp.solve_pnp_batch(g["uv"][i:i + 1].astype(np.float64), g["xyz"][i], K, method="dlt", num_it=rounds)
        q1 = po.dlt_solve(g["uv"][i].astype(np.float64), g["xyz"][i], K, num_it=rounds)
        print("  after %d round(s): max|dR| %.3e  sim %.3e" % (rounds, np.abs(o1["R"][0] - q1["R"]).max(), q1["min_abs_sim"]))
