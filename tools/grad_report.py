"""dev tool: per-tensor gradient error of the HIP engine vs the oracle (fp32 and lowp) on the tiny config."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tulip_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_model_gpu import build, rel_l2, _load
name = sys.argv[1] if len(sys.argv) > 1 else "g3_tiny_fp32"
z, meta, cfg = _load("tests/golden", name)
sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
m = build(cfg, sd, train=False)
eng = m.engine(); eng.bind(torch.device("cuda", 0))
P = eng.plan(meta["batch"]); P.x_in.copy_(lo.cuda()); P.target.copy_(hi.cuda())
eng.draw_drop_scales(P, False); eng.run_forward(P)
g = torch.zeros(eng.params.total, device="cuda"); eng.run_backward(P, g); torch.cuda.synchronize()
W = eng.params
grads = {n: g[W.offset[n]:W.offset[n] + W.numel[n]].view(W.shape[n]).cpu() for n in W.names}
_, _, _, og = O.tulip_loss_and_grads(sd, cfg, lo, hi)
_, _, _, ol = O.tulip_loss_and_grads(sd, cfg, lo, hi, lowp=True)
rows = []
for k in W.names:
    rows.append((rel_l2(grads[k], og[k]), rel_l2(grads[k], ol[k]), rel_l2(ol[k], og[k]), k, og[k].norm().item()))
rows.sort(reverse=True)
print("hip-vs-fp32  hip-vs-lowp  lowp-vs-fp32   |g|      name")
for r in rows[:25]: print(f"{r[0]:.3e}   {r[1]:.3e}   {r[2]:.3e}   {r[4]:.2e}  {r[3]}")
print("median hip-vs-fp32", np.median([r[0] for r in rows]))
