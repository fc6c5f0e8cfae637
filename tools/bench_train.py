#!/usr/bin/env python3
"""Training-step benchmark (BASELINE config 5): UtNet(64,'PReLU') forward + backward + Adam(amsgrad) on synthetic
crop batches, one process per GPU, gradients averaged with one flat RCCL all-reduce.

    python tools/bench_train.py [--cs 136] [--batch 30] [--steps 5] [--warmup 2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py ...

cs=136 is the valid UtNet size nearest to BASELINE's "128x128" (184 is the reference's own training crop,
configs/train_conf_utnet_std.yaml).  Weak scaling: the per-GPU batch is fixed.  FLOP accounting: 3 x the forward
FLOP of the reference's own convention (forward + data gradient + weight gradient)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from nind_denoise_amd import synth  # noqa: E402
from nind_denoise_amd.networks.UtNet import UtNet  # noqa: E402
from nind_denoise_amd.train import UtNetTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cs", type=int, default=136)
    ap.add_argument("--batch", type=int, default=30)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--funit", type=int, default=64)
    ap.add_argument("--loss", default="L1=0.5,MSE=0.5", help="criterion weights, e.g. MSSSIM=1 (the reference's default; needs "
                                                               "crops >= 161) or SSIM=0.5,L1=0.5")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    net = UtNet(funit=args.funit)
    net.load_state_dict(synth.make_utnet_state_dict(args.funit, seed=123))
    weights = {k: float(v) for k, v in (kv.split("=") for kv in args.loss.split(","))}
    tr = UtNetTrainer(net, lr=1e-4, beta1=0.75, device=dev, weights=weights)
    if world > 1:
        dist.broadcast(tr.flat, src=0)
    g = torch.Generator().manual_seed(100 + rank)
    clean = torch.rand(args.batch, 3, args.cs, args.cs, generator=g).to(dev)
    noisy = (clean + 0.1 * torch.randn(args.batch, 3, args.cs, args.cs, generator=g).to(dev)).clip(0, 1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.learn(noisy, clean)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.learn(noisy, clean)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        fwd = net.flops_per_tile(args.cs)
        samples = args.batch * world * args.steps
        print(json.dumps({"metric": "UtNet training step (fwd+bwd+Adam), crops/s", "value": round(samples / dt, 2),
                          "unit": "crops/s", "n_gpus": world, "ms_per_step": round(1e3 * dt / args.steps, 2),
                          "scaling": "weak", "dtype": "f32", "data": "synthetic",
                          "config": {"workload": f"UtNet({args.funit},PReLU) fwd+bwd+Adam(amsgrad), crop {args.cs}, "
                                                 f"per-GPU batch {args.batch}, loss {args.loss}", "forward_flop_per_crop": fwd},
                          "approx_tflops": round(3 * fwd * samples / dt / 1e12, 2), "loss": float(loss.item())}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
