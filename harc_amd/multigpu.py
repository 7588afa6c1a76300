"""Multi-GPU sharding of the hot path (BASELINE.json north_star): reads shard by k-mer (minimizer) bucket across the GPUs of one
node with a single all-to-all over xGMI (torch.distributed backend "nccl" == RCCL), after which every GPU chains and
delta-encodes its shard independently.  One process per GPU.  torch is plumbing here (device buffers, the collective);
the bucket function and everything after the exchange are HIP kernels behind the C-ABI.

The exchange is deterministic: reads are stably grouped by destination, so rank r receives, for s = 0..world-1 in
order, the reads of rank s whose bucket is r in their original order.
"""
import torch


class BucketSharder:
    def __init__(self, ctx, dist, device, readlen, bucket_fn=None):
        """ctx: harc_amd.HarcAmd (None on CPU tests when bucket_fn is given); dist: torch.distributed (initialised);
        bucket_fn(packed[int64 n x W], world) -> int64 bucket per read, defaults to the HIP kernel k_bucket."""
        self.ctx, self.dist, self.device, self.L = ctx, dist, device, readlen
        self.W = (2 * readlen + 63) // 64
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.bucket_fn = bucket_fn
        self.last_counts = None

    def pack(self, ascii_reads):
        """[n, L] uint8 ASCII (device) -> [n, W] int64 2-bit packed (reorder.cpp:184-209 layout), HIP kernel k_pack2"""
        n = ascii_reads.shape[0]
        out = torch.empty((n, self.W), dtype=torch.int64, device=self.device)
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()             # libharc_amd runs on its own stream: its inputs must be complete
        if n:
            self.ctx.pack_reads_device(ascii_reads.data_ptr(), n, ascii_reads.stride(0), out.data_ptr())
        return out

    def buckets(self, packed):
        if self.bucket_fn is not None:
            return self.bucket_fn(packed, self.world)
        n = packed.shape[0]
        b = torch.empty((n,), dtype=torch.int32, device=self.device)
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        if n:
            self.ctx.bucket_reads_device(packed.data_ptr(), n, self.world, b.data_ptr())
        return b.long()

    def partition(self, packed):
        """-> (send [n, W] grouped by destination, original order inside a group; counts [world] int64)"""
        if self.bucket_fn is None:                                    # HIP: k_bucket + one radix pass + gather (harc_amd_partition_reads_device)
            n = packed.shape[0]
            send = torch.empty_like(packed)
            counts = torch.zeros((self.world,), dtype=torch.int64, device=self.device)
            torch.cuda.current_stream().synchronize()                 # libharc_amd runs on its own stream: its inputs must be complete
            self.ctx.partition_reads_device(packed.data_ptr(), n, self.world, send.data_ptr(), counts.data_ptr())
            return send, counts
        b = self.buckets(packed)                                      # CPU tests: the same in torch
        order = torch.sort(b, stable=True).indices
        return packed[order].contiguous(), torch.bincount(b, minlength=self.world).to(torch.int64)

    def exchange(self, packed):
        """one all-to-all(v): -> [m, W] int64, the reads of every rank whose bucket is this rank"""
        dist, world = self.dist, self.world
        send, counts = self.partition(packed)
        recv_counts = torch.empty_like(counts)
        dist.all_to_all_single(recv_counts, counts)                 # 8 B per peer: how many reads each rank sends me
        sc, rc = counts.tolist(), recv_counts.tolist()
        recv = torch.empty((sum(rc), self.W), dtype=torch.int64, device=packed.device)
        dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc)   # the payload: 8W B per read, one chunk per xGMI link
        self.last_counts = (sc, rc)
        return recv

    def exchange_and_set(self, packed):
        recv = self.exchange(packed)
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()             # the all-to-all (RCCL stream, ordered behind torch's) must have landed
        self.ctx.set_reads_packed_device(recv.data_ptr(), recv.shape[0])
        return recv.shape[0]
