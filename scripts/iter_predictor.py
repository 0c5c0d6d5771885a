#!/usr/bin/env python3
"""Can a free input feature predict the QEIF iteration count?  (If problems could be sorted into
wavefronts by expected iteration count, early-exit divergence would shrink.)  CPU only: the C
oracle's iteration counts on the bench workload against features read off the pixels."""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import c_oracle  # noqa: E402
from pnp_solver_test_amd import synth  # noqa: E402

B = 1 << 18
K = synth.K_DEFAULT
uv, gt = synth.stress_batch(B, seed=42, pattern=synth.PATTERN6, K=K, is_quantized=True)
it = c_oracle.qeif_batch(uv.astype(np.float64), synth.PATTERN6.astype(np.float32).astype(np.float64), K)["iters"]
print("iterations: mean %.3f, distribution %s" % (it.mean(), np.bincount(it)[:10].tolist()))
slow = it >= 6
u, v = uv[..., 0].astype(np.float64), uv[..., 1].astype(np.float64)
feat = {
    "bbox_area": (u.max(1) - u.min(1)) * (v.max(1) - v.min(1)),
    "eye_distance": np.hypot(u[:, 0] - u[:, 1], v[:, 0] - v[:, 1]),
    "spread": np.sqrt(u.var(1) + v.var(1)),
    "offaxis": np.hypot(u.mean(1) - 160.0, v.mean(1) - 120.0),
    "true_depth": gt["t"][:, 2],
    "true_abs_yaw": np.abs(gt["yaw"]),
    "true_abs_pitch": np.abs(gt["pitch"]),
    "true_tilt": np.hypot(gt["yaw"], gt["pitch"]),
}


def best_threshold_accuracy(x):
    order = np.argsort(x)
    s = slow[order].astype(np.int64)
    n, tot = len(s), s.sum()
    below_slow = np.concatenate([[0], np.cumsum(s)])            # slow among the k smallest
    k = np.arange(n + 1)
    acc_hi = ((k - below_slow) + (tot - below_slow)) / n          # predict slow above the cut
    acc_lo = (below_slow + ((n - k) - (tot - below_slow))) / n    # predict slow below the cut
    return max(acc_hi.max(), acc_lo.max())


print("share of problems with 6 or more iterations: %.3f (always guessing 'fast' is right %.3f of the time)"
      % (slow.mean(), 1 - slow.mean()))
for name, x in feat.items():
    print("%-16s best single-threshold accuracy %.3f   corr with iterations %+.3f" % (name, best_threshold_accuracy(x), np.corrcoef(x, it)[0, 1]))
# what perfect knowledge would buy: iteration slots per wavefront with problems sorted by count
per_wave = it.reshape(-1, 64).max(1).mean()
sorted_wave = np.sort(it).reshape(-1, 64).max(1).mean()
print("wavefront iteration slots: as generated %.3f, perfectly sorted %.3f, mean iterations %.3f" % (per_wave, sorted_wave, it.mean()))
# sorting by the best feature
for name in ("bbox_area", "true_tilt"):
    o = np.argsort(feat[name])
    print("sorted by %-12s -> %.3f slots" % (name, it[o].reshape(-1, 64).max(1).mean()))
