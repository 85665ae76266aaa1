"""dev tool: HIP-graph time of forward only / forward+backward / full step."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tulip_amd import ops
from tulip_amd.trainer import Trainer
from tulip_amd.engine import TulipEngine
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 8))
args = argparse.Namespace(model="tulip_base", img=[16,1024], target=[64,1024], batch=B)
m = bench.make_model(args).to(dev).train(); tr = Trainer(m, B, device=dev, use_graph=False)
lo, hi = bench.synthetic(args, 0, dev); tr.load_batch(lo, hi)
eng, P = tr.eng, tr.P
def graph_time(fn, n=40):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g.capture_begin(); fn(); g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def fwd(): eng.draw_drop_scales(P, True); eng.run_forward(P)
def fwdbwd(): tr.g.zero_(); fwd(); eng.run_backward(P, tr.g)
def full(): fwdbwd(); tr._adamw()
for ov in (True, False):
    TulipEngine.overlap_wgrad = ov
    print(f"overlap_wgrad={ov}: fwd {graph_time(fwd):.3f} ms | fwd+bwd {graph_time(fwdbwd):.3f} ms | full {graph_time(full):.3f} ms", flush=True)
