"""Where the f16-pair backward's worst gradient rows come from (config 5): per-row error of the pair and the exact-fp32
kernels against the fp64 backward on the same record, with the active segment and the cancellation in U v."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from rayen_amd import ops, pack as _pack, workloads
from rayen_amd.constraint_module import ConstraintModule
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
cs = workloads.build_constraints(workloads.make_raw(name, seed=0))
layer = ConstraintModule(cs, create_map=False).cuda()
torch.set_default_dtype(torch.float64); layer64 = ConstraintModule(cs, create_map=False).cuda(); torch.set_default_dtype(torch.float32)
dev = torch.device("cuda", 0)
dp, _ = layer.device_pack(dev); dp64, _ = layer64.device_pack(dev)
exact = _pack.DevicePack(layer.packed_constants(), 0, fp32_mode=1)
print("info", dp.info().bwd_f32, dp.info().bwd32_check_pair, dp.info().bwd32_check_exact)
gen = torch.Generator().manual_seed(17)
B = 20037
v = torch.empty(B, cs.n).uniform_(-3, 3, generator=gen); g = torch.empty(B, cs.k).uniform_(-1, 1, generator=gen)
v, g = v.cuda(), g.cuda()
_, k64, act = ops.project_raw(v.double(), dp64, want_active=True)
kap = k64.float()
truth = ops.backward_raw(v.double(), k64, act, g.double(), dp64).cpu().numpy()
got = ops.backward_raw(v, kap, act, g, dp).cpu().double().numpy()
ref = ops.backward_raw(v, kap, act, g, exact, bucketed=False).cpu().double().numpy()
size = np.maximum(np.abs(truth).max(1), 1e-30)
ep, ee = np.abs(got - truth).max(1) / size, np.abs(ref - truth).max(1) / size
a = act.cpu().numpy()
order = np.argsort(-ep)[:12]
consts = layer.packed_constants()
W = np.asarray(consts.W); segs = consts.segments
for i in order:
    s = int(a[i, 0]); seg = segs[s]
    info = {"row": int(i), "e_pair": float(ep[i]), "e_exact": float(ee[i]), "kappa": float(k64[i]), "seg": s, "type": int(seg[0]) if not hasattr(seg, "type") else int(seg.type)}
    try:
        row0, nrows = (seg.row0, seg.nrows) if hasattr(seg, "row0") else (seg[1], seg[2])
        U = W[row0:row0 + nrows]; x = v[i].cpu().double().numpy()
        uv = U @ x
        info["cancel"] = float((np.abs(U) @ np.abs(x)).max() / max(np.linalg.norm(uv), 1e-300))
    except Exception as e:
        info["err"] = str(e)
    print(json.dumps(info))
print("quantiles pair", np.quantile(ep, [0.5, 0.9, 0.99, 0.999, 1.0]), "exact", np.quantile(ee, [0.5, 0.9, 0.99, 0.999, 1.0]))
is_q = np.array([int(x) for x in a[:, 0]]) > 0
print("rows on quadratics:", int(is_q.sum()), "worst pair there", ep[is_q].max() if is_q.any() else None, "worst on linear rows", ep[~is_q].max(), "exact:", ee[is_q].max() if is_q.any() else None, ee[~is_q].max())
for s in sorted(set(int(x) for x in a[:, 0][is_q]))[:80]:
    m = (a[:, 0] == s)
    print("seg", s, "rows", int(m.sum()), "worst pair %.2e exact %.2e" % (ep[m].max(), ee[m].max()), end=" | ")
print()
i = int(order[0]); s = int(a[i, 0]); seg = segs[s]
U = W[seg.row0:seg.row0 + seg.nrows]; x = v[i].cpu().double().numpy()
print("U abs max per row", np.abs(U).max(1), "W quad max", max(np.abs(W[sg.row0:sg.row0 + sg.nrows]).max() for sg in segs if sg.type == 2))
print("x", x)
print("Ux", U @ x, "got-truth", (got[i] - truth[i]) / size[i], "ref-truth", (ref[i] - truth[i]) / size[i])
print("phi", W[seg.aux_row] if seg.aux_row >= 0 else None)
# the worst row again, at other positions of a batch (arithmetic or position?) and emulated on the host
pos = [0, 1, 31, 32, 63, 64, 100, 1000]
v2 = v[:2048].clone(); g2 = g[:2048].clone(); k2 = k64[:2048].clone(); a2 = act[:2048].clone()
for p_ in pos:
    v2[p_] = v[i]; g2[p_] = g[i]; k2[p_] = k64[i]; a2[p_] = act[i]
t2 = ops.backward_raw(v2.double(), k2, a2, g2.double(), dp64).cpu().numpy()
o2 = ops.backward_raw(v2, k2.float(), a2, g2, dp).cpu().double().numpy()
for p_ in pos:
    print("pos", p_, "err %.3e" % (np.abs(o2[p_] - t2[p_]).max() / np.abs(t2[p_]).max()))
def pair(xx, scale):
    s = xx * scale
    h1 = s.astype(np.float16).astype(np.float64)
    h2 = (s - h1).astype(np.float16).astype(np.float64)
    return h1, h2
big = max(np.abs(W[sg.row0:sg.row0 + sg.nrows]).max() for sg in segs if sg.type == 2)
gU = 2.0 ** (13 - np.floor(np.log2(big)))
U1, U2 = pair(U, gU); x1, x2 = pair(x, 2.0 ** (13 - np.floor(np.log2(np.abs(x).max()))))
acc = U2 @ x1 + U1 @ x2 + U1 @ x1
wv = acc / np.linalg.norm(acc) * 8192.0
w1, w2 = pair(wv, 1.0)
u = (U2.T @ w1 + U1.T @ w2 + U1.T @ w1) / (gU * 8192.0)
u_true = U.T @ (U @ x) / np.linalg.norm(U @ x)
print("emulated pair scheme: rel err of u", np.abs(u - u_true).max() / np.abs(u_true).max())
