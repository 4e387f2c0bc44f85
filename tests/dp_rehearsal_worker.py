"""Rank process of tests/test_gpu_dp.py (not a test module).  Two of these run side by side on cuda:0 with the gloo
backend (RCCL refuses two ranks on one device): each takes its half of a seeded batch through `steps` VitTrainer steps -
forward, backward, bucketed gradient all-reduce launched from inside backward, SGD with the 1/world mean folded in - and
writes its final parameters to `out`.  Usage: python dp_rehearsal_worker.py <model> <R_total> <steps> <out.pt> [backend]
With backend "nccl" and WORLD_SIZE=1 (tests/test_gpu_dp.py::test_rccl_single_rank_step_is_bitwise_the_plain_step) the one rank
runs the same steps over RCCL with the collectives forced on (YV_DP_FORCE_COLLECTIVE=1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "yolov8-vit_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def make_batch(name: str, R: int):
    """Seeded state dict + patch-major crops + labels of the UNION batch (the single-rank side calls this too)."""
    from yvhip import engines
    P = engines.vit_cfg(name)[0]
    tok = (224 // P) ** 2
    sd = engines.init_vit_wrapper_state(name, 5, seed=33)
    g = torch.Generator().manual_seed(77)
    patches = (torch.rand(R * tok, 3 * P * P, generator=g) * 2 - 1).to(torch.bfloat16)
    labels = torch.randint(0, 5, (R,), generator=g, dtype=torch.int32)
    return sd, patches, labels, tok


def main():
    name, R, steps, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from yvhip.dist import shard_range
    from yvhip.training import VitTrainer
    sd, patches, labels, tok = make_batch(name, R)
    lo, hi = shard_range(R, rank, world)
    tr = VitTrainer(sd, name, 5, device="cuda:0", bucket_mb=1.0 if "tiny" in name else 32.0)
    pm = patches[lo * tok:hi * tok].to("cuda:0")
    lb = labels[lo:hi].to("cuda:0")
    losses, grads, starts = [], [], []
    for s in range(steps):
        if rank == 0 and s > 0:                   # parameters this step starts from (the test re-derives its gradient from them)
            starts.append({k: v.cpu().clone() for k, v in tr.state_dict().items()})
        tr.forward(pm, hi - lo)
        loss = tr.backward(pm, lb, hi - lo)
        tr.reducer.finish()                       # all-reduced SUM of the two ranks' gradients
        torch.cuda.synchronize()
        grads.append({k: (v / world).cpu() for k, v in tr.grad_dict().items()})
        tr.optimizer_step(0.01)
        losses.append(float(loss[0]))
    torch.cuda.synchronize()
    n_buckets = len(tr.reducer.launched)
    torch.save({"state": {k: v.cpu() for k, v in tr.state_dict().items()}, "losses": losses, "buckets": n_buckets, "grads": grads,
                "starts": starts, "span": (lo, hi)}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
