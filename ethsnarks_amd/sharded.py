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
With a CUDA `device` under the gloo backend (the one-GPU rehearsal of bench.py, ZK_BENCH_REHEARSE=1: gloo moves host
memory only) every exchanged buffer is staged through a host copy; the calls, their order and their sizes are the
ones the RCCL path makes.
"""
import ctypes as C
import numpy as np
import torch


class _DevView:
    """a device buffer owned by the library, seen by torch through __cuda_array_interface__ (no copy)"""
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class _Done:
    def wait(self):
        return None


class _StagedRecv:
    """irecv into a host buffer; wait() copies it to the device tensor it was meant for"""
    def __init__(self, req, host, dst):
        self.req, self.host, self.dst = req, host, dst

    def wait(self):
        self.req.wait()
        self.dst.copy_(self.host)


class ShardedProver:
    def __init__(self, ctx, dist, device):
        self.ctx, self.dist, self.device = ctx, dist, torch.device(device)
        self.world = dist.get_world_size()
        self.stage_host = self.device.type == "cuda" and dist.get_backend() == "gloo"
        self.buf = torch.empty((self.world, 640), dtype=torch.uint8, device=self.device)
        self.mine = self._view(ctx.partials_device_ptr(), 640)
        self._keep = []

    # ---- the collectives and transfers of this module, on device buffers (RCCL; gloo on the CPU emulation) or staged through the host
    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def _all_gather(self, out, mine):
        if self.stage_host:
            h = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(h.view(-1), mine.cpu())
            out.copy_(h)
        else:
            self.dist.all_gather_into_tensor(out.view(-1), mine)

    def _irecv(self, t, src):
        if self.stage_host:
            h = torch.empty(t.shape, dtype=t.dtype)
            return _StagedRecv(self.dist.irecv(h, src=src), h, t)
        return self.dist.irecv(t, src=src)

    def _isend(self, t, dst):
        if self.stage_host:
            h = t.cpu()
            self._keep.append(h)
            return self.dist.isend(h, dst=dst)
        return self.dist.isend(t, dst=dst)

    def _agree(self, code):
        """every rank contributes its local error code (0 = fine); all of them learn the largest: a failure on ANY rank ends the proof on
        EVERY rank, before a rank could block in a transfer or a collective its failed peer will never join"""
        st = torch.tensor([int(code)], dtype=torch.int32, device="cpu" if self.stage_host else self.device)
        self.dist.all_reduce(st, op=self.dist.ReduceOp.MAX)
        return int(st.item())

    def submit(self, witness=None, canonical=False, device_ptr=None):
        """enqueue this rank's share of a proof (host witness, or a witness resident in device memory)"""
        if device_ptr is not None:
            self.ctx.submit_resident(device_ptr, canonical)
        else:
            self.ctx.submit(witness, canonical)

    def finish(self):
        """wait for the share, all-gather the 640-byte partials, fold them in rank order: (ZkProof, timings)"""
        tm = self.ctx.collect_device()
        self._all_gather(self.buf, self.mine)
        self._sync()
        return self.ctx.prove_combine_device(self.buf.data_ptr(), self.world), tm

    def _view(self, ptr, nbytes):
        if self.device.type == "cuda":
            return torch.as_tensor(_DevView(ptr, nbytes), device=self.device)
        return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)))

    def prove_struct_split_witness_map(self, witness, canonical=False):
        """SURVEY 8(e) option 2: instead of every rank recomputing the whole witness map, ranks 0, 1, 2 run the A, B and C
        transform chains (row evaluations, iFFT, and -- A and B -- cosetFFT), ranks 1 and 2 send their m results to rank 0 (two
        32 m-byte peer transfers), rank 0 forms H (2 transforms on the critical path instead of 6) and sends every rank the
        coefficients its H-query shard multiplies.  Needs at least three ranks.

        Pipelined: every rank first queues what needs only the witness (zk_prove_submit_defer_h: upload, witness sort, A-, B-,
        L-query accumulations), so its GPU works on three of the four multi-exponentiations while the chains travel; the H-query
        is queued when the coefficients arrive (zk_prove_submit_h).

        Failures are agreed on before every phase in which a rank would wait for a peer (`_agree`: one 4-byte all-reduce, MAX of the
        local error codes): a rank whose local step failed -- an unsatisfying witness shows on rank 0 alone (degree of H), a HIP or
        argument error can show anywhere -- makes EVERY rank drop its deferred proof (zk_prove_abort) and raise together, instead of
        leaving its peers blocked in a receive, a broadcast or the final all-gather."""
        from .prover import ZkError
        ctx, dist, rank, world = self.ctx, self.dist, self.dist.get_rank(), self.world
        if world < 3:
            return self.prove_struct(witness, canonical)
        m = ctx.r1cs.domain_size
        shard = lambda r: ((m - 1) * r // world, (m - 1) * (r + 1) // world)      # zk_ctx_create's base-range rule for H
        self._keep = []

        def phase(fn):
            """run this rank's local step, then agree: returns normally on every rank, or raises on every rank"""
            code, err = 0, None
            try:
                fn()
            except ZkError as e:
                code, err = (e.code or 4), e
            except Exception as e:                                     # not an error code of the library: still a failure every rank must hear of
                code, err = 4, e
            worst = self._agree(code)
            if worst:
                try:
                    ctx.abort()                                         # the deferred proof will not be completed: drain, free the context
                except Exception:
                    pass
                if err is not None:
                    raise err
                raise ZkError(worst, "h[m-1] != 0 on rank 0: the witness does not satisfy the constraint system" if worst == 7
                              else "a peer rank failed (code %d): sharded proof dropped on every rank" % worst)

        def local_chains():
            ctx.submit_defer_h(witness, canonical)
            if rank < 3:
                ctx.chain_submit(None, rank)                            # behind the witness sorts on the main stream, beside the accumulations
                ctx.chain_wait()
        phase(local_chains)

        got = []

        def gather_chains_and_form_h():
            if rank == 0:
                got.extend(torch.empty(32 * m, dtype=torch.uint8, device=self.device) for _ in range(2))
                for q in [self._irecv(got[i], src=i + 1) for i in range(2)]:
                    q.wait()
                self._sync()
                ctx.h_from_chains_submit(ctx.chain_device_ptr(0), got[0].data_ptr(), got[1].data_ptr())
                ctx.chain_wait(check_degree=True)                       # ZK_ERR_DEGREE here is the unsatisfying witness
            elif rank < 3:
                self._isend(self._view(ctx.chain_device_ptr(rank), 32 * m), dst=0).wait()
        phase(gather_chains_and_form_h)                                 # every rank learns whether H exists before it waits for its share

        def scatter_h():
            if rank == 0:
                h = self._view(ctx.h_device_ptr(), 32 * m)
                reqs = [self._isend(h[32 * shard(r)[0]:32 * shard(r)[1]], dst=r) for r in range(1, world)]
                ctx.submit_h(ctx.h_device_ptr() + 32 * shard(0)[0])
                for q in reqs:
                    q.wait()
            else:
                lo, hi = shard(rank)
                mine_h = torch.empty(32 * (hi - lo), dtype=torch.uint8, device=self.device)
                self._irecv(mine_h, src=0).wait()
                self._sync()
                self._keep.append(mine_h)                               # stays alive until the proof is collected
                ctx.submit_h(mine_h.data_ptr())
        phase(scatter_h)                                                # (a rank whose H-query could not be queued must not leave the others in the all-gather)
        proof, _ = self.finish()
        self._keep = []
        return proof

    def prove_struct(self, witness, canonical=False, timings=False):
        self.submit(witness, canonical)
        proof, tm = self.finish()
        return (proof, tm) if timings else proof
