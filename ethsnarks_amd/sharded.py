"""MSM base-range sharding over the GPUs of one node (BASELINE config 5).

Each rank holds 1/G of every query's bases (static, uploaded once) and proves its shard; the four partial
sums (At, Bt, Ht, Lt in XYZZ coordinates, 640 bytes in zk_partials layout) stay in a device buffer of the
context.  They are exchanged with ONE all-gather on device buffers (RCCL over xGMI when the backend is
"nccl"; the payload is latency- not bandwidth-bound) and folded in fixed rank order by
`zk_prove_combine_device`, so every rank obtains the same, deterministic proof.  Elliptic-curve addition is
not an ncclRedOp_t, hence all-gather + local fold instead of a literal all-reduce.  The witness -> H
pipeline is replicated on every rank.

With a CPU `device` (gloo: the multi-process tests drive the CPU emulation build, whose "device" memory is host
memory) the same entry points run, the buffers being viewed through numpy instead of __cuda_array_interface__.
"""
import ctypes as C
import numpy as np
import torch


class _DevView:
    """a device buffer owned by the library, seen by torch through __cuda_array_interface__ (no copy)"""
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class ShardedProver:
    def __init__(self, ctx, dist, device):
        self.ctx, self.dist, self.device = ctx, dist, torch.device(device)
        self.world = dist.get_world_size()
        self.buf = torch.empty((self.world, 640), dtype=torch.uint8, device=self.device)
        ptr = ctx.partials_device_ptr()
        if self.device.type == "cuda":
            self.mine = torch.as_tensor(_DevView(ptr, 640), device=self.device)
        else:
            self.mine = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * 640).from_address(ptr)))

    def submit(self, witness=None, canonical=False, device_ptr=None):
        """enqueue this rank's share of a proof (host witness, or a witness resident in device memory)"""
        if device_ptr is not None:
            self.ctx.submit_resident(device_ptr, canonical)
        else:
            self.ctx.submit(witness, canonical)

    def finish(self):
        """wait for the share, all-gather the 640-byte partials, fold them in rank order: (ZkProof, timings)"""
        tm = self.ctx.collect_device()
        self.dist.all_gather_into_tensor(self.buf.view(-1), self.mine)
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()
        return self.ctx.prove_combine_device(self.buf.data_ptr(), self.world), tm

    def prove_struct(self, witness, canonical=False, timings=False):
        self.submit(witness, canonical)
        proof, tm = self.finish()
        return (proof, tm) if timings else proof
