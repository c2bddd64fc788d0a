"""MSM base-range sharding over the GPUs of one node (BASELINE config 5).

Each rank holds 1/G of every query's bases (static, uploaded once) and proves its shard:
`zk_prove_partial` -> 640 bytes of partial sums (At, Bt, Ht, Lt in XYZZ coordinates).  The partials
are exchanged with ONE all-gather (RCCL over xGMI when the backend is "nccl"; the payload is latency-
not bandwidth-bound) and folded in fixed rank order by `zk_prove_combine`, so every rank obtains the
same, deterministic proof.  Elliptic-curve addition is not an ncclRedOp_t, hence all-gather + local
fold instead of a literal all-reduce.  The witness -> H pipeline is replicated on every rank.
"""
import numpy as np
import torch


class ShardedProver:
    def __init__(self, ctx, dist, device):
        self.ctx, self.dist, self.device = ctx, dist, device
        self.world = dist.get_world_size()
        self.buf = torch.empty((self.world, 640), dtype=torch.uint8, device=device)

    def prove_struct(self, witness, canonical=False, timings=False):
        res = self.ctx.prove_partial(witness, canonical, timings=timings)
        part, tm = (res if timings else (res, None))
        mine = torch.from_numpy(part.view(np.uint8).copy()).to(self.device)
        self.dist.all_gather_into_tensor(self.buf.view(-1), mine)
        allp = self.buf.cpu().numpy().reshape(-1).view(np.uint64)
        proof = self.ctx.prove_combine(allp)
        return (proof, tm) if timings else proof
