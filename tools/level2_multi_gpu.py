#!/usr/bin/env python3
"""One contact island over N GPUs (level-2 sharding, library-issued RCCL exchange): parity against the unsplit island + timing.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 tools/level2_multi_gpu.py [--dims 50 40 50]

The rendezvous of the unique id and the max-over-ranks reductions use torch.distributed (gloo: host-side, it never touches the data path);
the data path is the library's own communicator."""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs=3, default=[50, 40, 50])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bits", type=int, default=32)
    ap.add_argument("--substeps", type=int, default=4)
    ap.add_argument("--closed-loop-steps", type=int, default=0, help="> 0: the island's manifolds are the device closed loop's own after that many steps (overflow colour included)")
    ap.add_argument("--check-steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    if os.environ.get("AVN_LEVEL2_SINGLE_DEVICE") == "1":   # validation only: all ranks on GPU 0 (RCCL may refuse a duplicate device)
        local = 0
    torch.cuda.set_device(local)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    import avian_amd
    from avian_amd import level2_bench
    lib = avian_amd.load_library()

    def bcast(b):
        o = [b]
        dist.broadcast_object_list(o, src=0)
        return o[0]

    def armax(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    res = level2_bench.run(lib, rank, world, local, bcast, armax, dist.barrier, dims=tuple(args.dims), steps=args.steps, warmup=args.warmup, bits=args.bits, substeps=args.substeps,
                           closed_loop_steps=args.closed_loop_steps, check_steps=args.check_steps)
    if rank == 0:
        print(json.dumps(res))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
