"""Multi-GPU sharding of the hot path (BASELINE.json north_star): reads shard by k-mer (minimizer) bucket across the GPUs of one
node with a single all-to-all over xGMI, after which every GPU chains and delta-encodes its shard independently.  One process per
GPU.

The bucket function, the grouping by destination and the collective itself (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd,
8W + 4 bytes per read: packed read + u32 global id) are inside libharc_amd.so (harc_amd_shard_exchange, csrc/shard.hip + comm.cpp).
This module only bootstraps the library's communicator from a torch.distributed process group: rank 0 asks the library for an
ncclUniqueId and the bytes are broadcast with the group the caller already has.
"""
import torch


def init_comm(ctx, dist, device=None):
    """ctx: harc_amd.HarcAmd; dist: an initialised torch.distributed.  Every rank calls this once; afterwards
    ctx.shard_exchange() runs the library's all-to-all."""
    from . import api
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    buf = torch.zeros(api.COMM_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        buf = torch.frombuffer(bytearray(api.comm_get_id()), dtype=torch.uint8).clone()
    buf = buf.to(dev)
    dist.broadcast(buf, src=0)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    ctx.comm_init(bytes(buf.cpu().numpy().tobytes()), world, rank)
    return world, rank


def allreduce_signature(dist, sig, device):
    """(count, sum mod 2^64, xor) of a read multiset, combined over the ranks: the order-independent signature of the whole job"""
    import numpy as np
    t = torch.from_numpy(np.array([sig[0], sig[1], sig[2]], dtype=np.uint64).view(np.int64)).to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    cnt, sm, xr = 0, 0, 0
    for o in out:
        v = o.cpu().numpy().view(np.uint64)
        cnt += int(v[0]); sm = (sm + int(v[1])) % (1 << 64); xr ^= int(v[2])
    return (cnt, sm, xr)
