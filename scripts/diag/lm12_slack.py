"""Where the GPU's 12-state LM solver differs from the reference run on reference-convergent
cases: indices, angles and the oracle's view of the same cases."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import pnp_solver_test_amd as pnp  # noqa: E402
from conftest import load_golden, rel_t, rot_angle  # noqa: E402
from oracle import pnp_oracle as po  # noqa: E402

K = po.K_DEFAULT
dev = torch.device("cuda:0")
report = {}
for tag, keys in (("lm12_n6", po.LM6_KEYS), ("lm12_n15", None)):
    g = load_golden(tag)
    P = po.pattern_array(po.PATTERN_ALEXANDER, keys)
    conv = np.isfinite(g["R"]).all(axis=(1, 2)) & (g["res_norm"] < 1e-2)
    for ft, tol in ((np.float64, 1e-7), (np.float32, 1e-5)):
        tt = torch.float64 if ft == np.float64 else torch.float32
        out = pnp.solve_pnp_batch(torch.from_numpy(g["uv"].astype(ft)).to(dev), torch.from_numpy(P.astype(ft)).to(dev), K,
                                  method="lm12")
        R = out["R"].cpu().numpy()
        t = out["t"].cpu().numpy()
        ang = rot_angle(R, g["R"])
        rt = rel_t(t, g["t"])
        bad = np.flatnonzero(conv & ~((ang <= tol) & (rt <= 10 * tol)))
        row = dict(conv=int(conv.sum()), bad=bad.tolist(), ang=[float(ang[i]) for i in bad], relt=[float(rt[i]) for i in bad],
                   ref_res=[float(g["res_norm"][i]) for i in bad],
                   gpu_res=[float(out["res_norm"][i].item()) for i in bad],
                   max_ang_good=float(np.nanmax(np.where(conv & (ang <= tol), ang, 0.0))))
        if ft == np.float32:
            # the oracle on the float32-rounded pattern: is the difference input rounding?
            Pf = P.astype(np.float32).astype(np.float64)
            row["oracle_f32_inputs_ang_vs_ref"] = [float(rot_angle(po.lm12_solve(g["uv"][i].astype(np.float64), Pf, K)["R"], g["R"][i])[0]) for i in bad]
            row["gpu_vs_oracle_f32_inputs"] = [float(rot_angle(po.lm12_solve(g["uv"][i].astype(np.float64), Pf, K)["R"], R[i])[0]) for i in bad]
        report["%s_%s" % (tag, np.dtype(ft).name)] = row
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "lm12_slack.json"), "w") as f:
    json.dump(report, f, indent=1)
print(json.dumps(report, indent=1))
