"""dev tool: what this chip's HBM delivers to plain streaming kernels (torch fill / copy / sum on 1-GiB fp32 tensors, HIP
events, 20 launches each) -- the practical ceilings beside the 8 TB/s the rooflines are priced against."""
import torch
dev = torch.device("cuda", 0)
n = 1 << 28
x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
gb = n * 4 / 1e9
for name, fn, moved in (("fill (write only)", lambda: x.fill_(1.0), gb), ("copy (read + write)", lambda: y.copy_(x), 2 * gb),
                        ("sum (read only)", lambda: x.sum(), gb), ("add in place (read + write same lines)", lambda: x.add_(1.0), 2 * gb)):
    s = t(fn)
    print(f"{name:42s} {moved / s / 1e3:6.2f} TB/s  ({s * 1e6:7.1f} us per GiB tensor)")
