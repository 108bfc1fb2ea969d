"""developer tool: per-tensor difference of the gradients with / without the fused backward kernel"""
import os, sys
import numpy as np
_T = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _T); sys.path.insert(0, os.path.dirname(_T))
import hashfill as hf
from __graft_entry__ import load_package
pkg = load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
H, A, M = 512, 6, int(sys.argv[2]) if len(sys.argv) > 2 else 2
params = hf.fill_params(710, H, A)
obs = hf.hf_bytes(711, (N, 4, 84, 84))
actions = (hf.hf_u32(712, N) % np.uint32(A)).astype(np.int64)
old_lp = np.log(np.full((N, A), 1.0 / A, np.float32))
adv, ret = hf.hf_range(714, (N,), -1, 1), hf.hf_range(715, (N,), -1, 1)
masks = np.ones(N, np.uint8)
res = {}
for fused in (0, 2):
    eng = pkg.Engine(N // 8, 8, A, H, precision=pkg.BF16)
    eng.set_option(pkg.OPT_FUSED_BWD, fused)
    eng.load_params(params)
    eng.set_batch(obs, actions, old_lp, adv, ret, masks)
    m = eng.train(2.5e-4, 1, M)
    res[fused] = (m["loss"], m["grad_norm"], eng.export_grads())
    eng.close()
print("loss", res[0][0], res[2][0]); print("norm", res[0][1], res[2][1])
g0, g1 = res[0][2], res[2][2]
# reference order: conv1.w 8192, b 32, conv2.w 32768, b 64, conv3.w 36864, b 64, fc.w, fc.b, heads
sizes = [("w1", 32 * 4 * 8 * 8), ("b1", 32), ("w2", 64 * 32 * 16), ("b2", 64), ("w3", 64 * 64 * 9), ("b3", 64), ("wfc", 512 * 3136), ("bfc", 512)]
o = 0
for name, sz in sizes:
    a, b = g0[o:o + sz], g1[o:o + sz]
    print(f"{name:4s} max|g0| {np.abs(a).max():.4e} max|diff| {np.abs(a - b).max():.4e} rel-l2 {np.linalg.norm(a - b) / (np.linalg.norm(a) + 1e-30):.3e}")
    if name == "w3" and np.abs(a - b).max() > 1e-3 * np.abs(a).max():
        d = np.abs(a - b).reshape(64, 64, 3, 3)  # [oc][c][kh][kw]
        print("  by oc-atom:", d.reshape(4, 16, -1).max(axis=(1, 2)))
        print("  by c-quarter:", d.transpose(1, 0, 2, 3).reshape(4, 16, -1).max(axis=(1, 2)))
        print("  by tap:", d.max(axis=(0, 1)))
    o += sz
