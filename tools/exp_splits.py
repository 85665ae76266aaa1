import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd import ops
from tulip_amd.trainer import Trainer
from tulip_amd.engine import TulipEngine
dev = torch.device("cuda", 0)
args = argparse.Namespace(model="tulip_base", img=[16,1024], target=[64,1024], batch=8)
def run(target_blocks, kmin):
    def _splits(cls, Mout, Nout, K):
        tiles = ((Mout + 127) // 128) * ((Nout + 95) // 96)
        s = max(1, min(target_blocks // max(tiles, 1), K // kmin, cls.WS_ELEMS // (Mout * Nout)))
        while True:
            e = ops.gemm_effective_splits(K, s)
            if e == s: return s
            s = e
    TulipEngine._splits = classmethod(_splits)
    m = bench.make_model(args).to(dev).train(); tr = Trainer(m, 8, device=dev)
    lo, hi = bench.synthetic(args, 0, dev); tr.load_batch(lo, hi)
    for _ in range(5): tr.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): tr.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 40 * 1e3
for tb, km in [(512, 256), (256, 256), (256, 512), (128, 512), (128, 1024), (64, 1024)]:
    print(f"target_blocks {tb:4d} min-k {km:5d}: {run(tb, km):.3f} ms", flush=True)
