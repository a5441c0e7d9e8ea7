"""Worker of tests/test_sharded_closed_loop_cpu.py (torch.distributed.run, world_size 2, gloo): each rank runs ONLY its pile's sub-world, the three
per-step exchanges of ShardedClosedLoop go through dist.all_gather_object; rank 0 merges the bodies into <out>.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402,F401
import torch.distributed as dist  # noqa: E402

from avian_amd import _ffi as F, shard  # noqa: E402
from helpers import hip_lib, oracle_lib  # noqa: E402
from test_sharded_closed_loop_cpu import piles, plan_by_pile  # noqa: E402


def main():
    out, steps = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = hip_lib() if os.environ.get("AVN_SHARD_BACKEND") == "hip" else oracle_lib()
    bodies, colliders = piles(world, 24)
    p = plan_by_pile(bodies, world, 24)
    b, loc, g2l = shard.split_bodies(p, rank, bodies)
    c = shard.split_colliders(g2l, colliders)
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**b); w.colliders_upload(**c); w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    native = len(sys.argv) > 3 and sys.argv[3] == "native"

    def all_gather(obj):
        box = [None] * world
        dist.all_gather_object(box, obj)
        return box

    if native:   # the bookkeeping in the library (avn_shard_*), the three exchanges as flat tensors
        loop = shard.ShardedClosedLoopNative(lib, w, p, rank, colliders)
        gather = shard.tensor_gather(dist, torch)
        for _ in range(steps):
            loop.step(gather)
    else:
        loop = shard.ShardedClosedLoop(lib, w, p, rank, colliders, np.asarray(bodies["rb_type"]))
        for _ in range(steps):
            loop.step(all_gather)
    mine = p.rank_of_body[loc] == rank
    got = all_gather((loc[mine], {k: v[mine] for k, v in w.bodies_download().items()}))
    if rank == 0:
        ref = F.World(lib, F.default_config(32, substeps=4))   # only for the static bodies' (unchanged) records and the array shapes
        ref.bodies_upload(**bodies)
        merged = ref.bodies_download()
        for idx, d in got:
            for k in merged:
                merged[k][idx] = d[k]
        off, handles = loop.global_lists
        np.savez(out, offsets=off, handles=handles, **merged)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
