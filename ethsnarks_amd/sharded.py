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

    def _view(self, ptr, nbytes):
        if self.device.type == "cuda":
            return torch.as_tensor(_DevView(ptr, nbytes), device=self.device)
        return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)))

    def prove_struct_split_witness_map(self, witness, canonical=False):
        """SURVEY 8(e) option 2: instead of every rank recomputing the whole witness map, ranks 0, 1, 2 run the A, B and C
        transform chains (row evaluations, iFFT, cosetFFT), ranks 1 and 2 send their m coset evaluations to rank 0 (two
        32 m-byte peer transfers), rank 0 forms H ((a b - c) / Z, icosetFFT: 3 transforms on the critical path instead of 7)
        and sends every rank the coefficients its H-query shard multiplies.  Needs at least three ranks.

        Pipelined: every rank first queues what needs only the witness (zk_prove_submit_defer_h: upload, witness sort, A-, B-,
        L-query accumulations), so its GPU works on three of the four multi-exponentiations while the chains travel; the H-query
        is queued when the coefficients arrive (zk_prove_submit_h).  An unsatisfying witness is found by rank 0 alone (degree of
        H): it broadcasts a status word before the scatter, and EVERY rank drops its deferred proof and raises ZK_ERR_DEGREE
        together instead of waiting for coefficients that never come."""
        from .prover import ZkError
        ctx, dist, rank, world = self.ctx, self.dist, self.dist.get_rank(), self.world
        if world < 3:
            return self.prove_struct(witness, canonical)
        m = ctx.r1cs.domain_size
        shard = lambda r: ((m - 1) * r // world, (m - 1) * (r + 1) // world)      # zk_ctx_create's base-range rule for H
        sync = (lambda: torch.cuda.current_stream(self.device).synchronize()) if self.device.type == "cuda" else (lambda: None)
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        ctx.submit_defer_h(witness, canonical)
        try:
            if rank < 3:
                ctx.chain_submit(None, rank)                            # behind the witness sorts on the main stream, beside the accumulations
            if rank == 0:
                got = [torch.empty(32 * m, dtype=torch.uint8, device=self.device) for _ in range(2)]
                reqs = [dist.irecv(got[i], src=i + 1) for i in range(2)]
                ctx.chain_wait()
                for q in reqs:
                    q.wait()
                sync()
                ctx.h_from_chains_submit(ctx.chain_device_ptr(0), got[0].data_ptr(), got[1].data_ptr())
                try:
                    ctx.chain_wait(check_degree=True)
                except ZkError as e:
                    status[0] = e.code
            elif rank < 3:
                ctx.chain_wait()
                dist.send(self._view(ctx.chain_device_ptr(rank), 32 * m), dst=0)
            dist.broadcast(status, src=0)                               # every rank learns whether H exists before it waits for its share
            code = int(status.item())
            if code:
                raise ZkError(code, "h[m-1] != 0 on rank 0: the witness does not satisfy the constraint system")
            if rank == 0:
                h = self._view(ctx.h_device_ptr(), 32 * m)
                reqs = [dist.isend(h[32 * shard(r)[0]:32 * shard(r)[1]], dst=r) for r in range(1, world)]
                ctx.submit_h(ctx.h_device_ptr() + 32 * shard(0)[0])
                for q in reqs:
                    q.wait()
            else:
                lo, hi = shard(rank)
                mine_h = torch.empty(32 * (hi - lo), dtype=torch.uint8, device=self.device)
                dist.recv(mine_h, src=0)
                sync()
                self._keep_h = mine_h                                   # stays alive until the proof is collected
                ctx.submit_h(mine_h.data_ptr())
        except Exception:
            ctx.abort()                                                 # the deferred proof will not be completed: drain, free the context
            raise
        proof, _ = self.finish()
        return proof

    def prove_struct(self, witness, canonical=False, timings=False):
        self.submit(witness, canonical)
        proof, tm = self.finish()
        return (proof, tm) if timings else proof
