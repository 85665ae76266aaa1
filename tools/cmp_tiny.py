import torch, numpy as np, sys
sys.path.insert(0, ".")
from tests.test_model_gpu import _load, build, rel_l2
from oracle import tulip_oracle as O
z, meta, cfg = _load("tests/golden", "g12_tiny3_expanding")
sd = O.key_seeded_state_dict(cfg, seed=meta["seed"])
lo, hi = O.synthetic_batch(cfg, meta["batch"], seed=1234 + meta["seed"])
m = build(cfg, sd, train=True); m.eval()
eng = m.engine(); eng.bind(torch.device("cuda", 0))
P = eng.plan(meta["batch"]); P.x_in.copy_(lo.cuda()); P.target.copy_(hi.cuda())
eng.draw_drop_scales(P, False, None); eng.run_forward(P)
g = torch.zeros(eng.params.total, device="cuda"); eng.run_backward(P, g); torch.cuda.synchronize()
out = {"g": g.cpu(), "pred": P.pred.cpu()}
for k, v in P.bufs.items():
    if k.startswith("layers.0.blocks.0.") or k.startswith("enc0"):
        out[k] = v.cpu().clone()
torch.save(out, sys.argv[1])
if len(sys.argv) > 2:
    a = torch.load(sys.argv[2])
    W = eng.params
    for k in out:
        if k == "g":
            for n in W.names:
                x, y = out["g"][W.offset[n]:W.offset[n]+W.numel[n]], a["g"][W.offset[n]:W.offset[n]+W.numel[n]]
                if not torch.equal(x, y): print("grad differs", n, rel_l2(x, y))
        elif not torch.equal(out[k], a[k]):
            print("buffer differs", k, rel_l2(out[k].float(), a[k].float()))
