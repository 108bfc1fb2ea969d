"""debug helper (not a test): per-tensor gradient comparison GPU vs oracle on the G4 batch."""
import sys, os
import numpy as np
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/
sys.path.insert(0, _T)
sys.path.insert(0, os.path.dirname(_T))
import hashfill as hf
import oracle_lib as orc
from __graft_entry__ import load_package
pkg = load_package()
H, A, N, M = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 4, 64, 1
prec = pkg.BF16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else pkg.FP32
params = hf.fill_params(410, H, A)
obs = hf.hf_bytes(401, (N, 4, 84, 84))
actions = (hf.hf_u32(402, N) % np.uint32(A)).astype(np.int64)
old_lp = orc.log_softmax(hf.hf_range(403, (N, A), -1, 1))
adv = hf.hf_range(404, (N,), -1, 1); ret = hf.hf_range(405, (N,), -1, 1)
masks = (hf.hf_unit(406, N) >= np.float32(0.1)).astype(np.uint8)
eng = pkg.Engine(8, 8, A, H, precision=prec)
eng.load_params(params)
eng.set_batch(obs, actions, old_lp, adv, ret, masks)
m = eng.train(2.5e-4, 1, M)
w = orc.train(params, H, A, obs, actions, old_lp, adv, ret, masks, 1, M)
print("loss", m["loss"], w["loss"], "norm", m["grad_norm"], w["grad_norm"])
g = eng.export_grads(); wg = w["last_grads"]
cg = min(1.0, 0.5 / (float(m["grad_norm"][0, 0]) + 1e-6)); cw = min(1.0, 0.5 / (float(w["grad_norm"][0, 0]) + 1e-6))
g = g / cg; wg = wg / cw
offs = orc.param_offsets(H, A)
names = ["c1w", "c1b", "c2w", "c2b", "c3w", "c3b", "fcw", "fcb", "aw", "ab", "vw", "vb"]
for k in range(12):
    a, b = g[offs[k]:offs[k + 1]], wg[offs[k]:offs[k + 1]]
    print(f"{names[k]:4s} |ours|={np.linalg.norm(a):10.4f} |ref|={np.linalg.norm(b):10.4f} maxdiff={np.abs(a - b).max():.3e} "
          f"ratio={np.dot(a, b) / (np.dot(b, b) + 1e-30):.4f}")
