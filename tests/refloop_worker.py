"""Worker of tests/test_round3_gpu.py::test_reference_loop_body_on_the_dropin: the body of the reference's training
loop (tulip/engine_upsampling.py:69-100) restated VERBATIM in its calling convention -- `torch.autocast("cuda")` around
`model(lo, hi, eval=False)`, the NativeScaler sequence scale -> backward -> unscale_ -> get_grad_norm_ -> step -> update
(tulip/util/misc.py:292-305) on a GradScaler with its default initial scale 65 536, `torch.optim.AdamW(betas=(0.9, 0.95))`
over timm-style parameter groups (main_lidar_upsampling.py:282-283: ndim <= 1 -> no decay), the module wrapped in
`DistributedDataParallel(model, device_ids=[gpu])` (:277) over a ONE-rank `nccl` (= RCCL) group -- on the drop-in module,
model and data of fixture g7 (reference model + torch.optim.AdamW, fp32)."""
import math
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def get_grad_norm_(parameters, norm_type: float = 2.0):           # misc.py:317-329
    parameters = [p for p in parameters if p.grad is not None]
    return torch.norm(torch.stack([torch.norm(p.grad.detach(), norm_type) for p in parameters]), norm_type)


def main():
    out_path, seed, batch, data_seed, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    from oracle import tulip_oracle as O
    from tests.test_model_gpu import build
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg = O.tiny_config(drop_path_rate=0.0)
    sd = O.key_seeded_state_dict(cfg, seed=seed)
    lo, hi = O.synthetic_batch(cfg, batch, seed=data_seed)
    lo, hi = lo.cuda(), hi.cuda()
    model = build(cfg, sd, train=True)
    model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=False)   # main:277
    model_without_ddp = model.module
    no_decay = [p for p in model_without_ddp.parameters() if p.ndim <= 1]          # timm param_groups_layer_decay, no
    decay = [p for p in model_without_ddp.parameters() if p.ndim > 1]              # layer decay (main:282)
    optimizer = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.01}],
                                  lr=5e-4, betas=(0.9, 0.95))                     # main:283
    scaler = torch.amp.GradScaler("cuda")                                          # misc.py:290 (init scale 65 536)
    assert scaler.get_scale() == 65536.0
    accum_iter = 1
    losses, norms, scales = [], [], []
    model.train(True)
    optimizer.zero_grad()
    for data_iter_step in range(steps):
        with torch.autocast("cuda"):                                               # engine:77-80
            _, total_loss, pixel_loss = model(lo, hi, eval=False)
        total_loss_value = total_loss.item()
        pixel_loss.item()
        if not math.isfinite(total_loss_value):
            sys.exit(1)
        total_loss /= accum_iter
        scaler.scale(total_loss).backward(create_graph=False)                      # misc.py:295
        scaler.unscale_(optimizer)                                                 # :302
        norm = get_grad_norm_(model.parameters())                                  # :303
        scaler.step(optimizer)                                                     # :304
        scaler.update()                                                            # :305
        optimizer.zero_grad()                                                      # engine:96-97
        torch.cuda.synchronize()                                                   # engine:100
        losses.append(total_loss_value); norms.append(norm.item()); scales.append(scaler.get_scale())
    with torch.no_grad(), torch.autocast("cuda"):
        _, l, _ = model(lo, hi, eval=True)
    losses.append(l.item())
    torch.save({"losses": losses, "norms": norms, "scales": scales, "backend": dist.get_backend(),
                "finite": all(bool(torch.isfinite(p).all()) for p in model.parameters())}, out_path)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
