"""Worker of tests/test_dshard_cpu.py / tests/test_gpu_dshard.py (torch.distributed.run, world_size 2, gloo): each process is one rank of the device-sharded closed
loop (avn_dshard_*); the bodies' records go through ONE tensor all-gather per step (avian_amd/shard.py: dshard_step_distributed).  Rank 0 writes <out>.npz."""
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
from test_dshard_cpu import owner_by_pile  # noqa: E402
from test_sharded_closed_loop_cpu import piles  # noqa: E402


def main():
    out, steps = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = hip_lib() if os.environ.get("AVN_SHARD_BACKEND") == "hip" else oracle_lib()
    bodies, colliders = piles(world, 24)
    owner = owner_by_pile(bodies, world, 24)
    w = F.World(lib, F.default_config(32, substeps=4))
    w.bodies_upload(**bodies); w.colliders_upload(**colliders); w.existing_pairs_upload(np.zeros(0, np.uint64)); w.collider_materials_upload(friction=0.5)
    w.pipeline_enable(); w.dshard_enable(world, rank, owner)
    gather = shard.tensor_gather(dist, torch)
    for _ in range(steps):
        shard.dshard_step_distributed(w, gather)
    if rank == 0:
        off, handles = w.pipeline_handles()
        np.savez(out, offsets=off, handles=handles, **w.bodies_download())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
