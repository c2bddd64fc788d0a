"""N > 1 path on CPU: gloo ranks (world sizes 2, 3, 4 and 8), each with a sharded prover context (CPU emulation build of
the HIP sources, shipped launch shapes), exchange their 640-byte partials with all_gather -- through the same
device-buffer entry points the GPU path uses -- and must reproduce the oracle's proof.  World 2 runs the replicated
witness map only (option 2 of SURVEY 8(e) needs three ranks and falls back); 3 is option 2's minimum (shard boundaries
that are not powers of two), 8 the target of BASELINE config 5."""
import os
import socket
import sys
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, nC, port, emul_so, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import oracle_lib as O
    from ethsnarks_amd import prover as P, r1cs as R, fields as F
    from ethsnarks_amd.sharded import ShardedProver
    P.load_library(emul_so)
    r, w = R.synthetic_chain(nC, 1)
    wm = F.fr_to_mont(w)
    pk_o, _ = O.keygen(r, seed=17)
    expect, _ = O.prove(pk_o, r, wm)
    pk = P.ProvingKey.from_parts(**pk_o.parts())
    ctx = P.ProverContext(pk, r, shard_rank=rank, shard_count=world)
    shared = ctx.info()["share_B"]                      # the rank's A / B / L shards ride one shared witness sort
    sp = ShardedProver(ctx, dist, torch.device("cpu"))
    got = P.proof_to_json(sp.prove_struct(wm), wm[1:2])
    got2 = P.proof_to_json(sp.prove_struct(wm, timings=True)[0], wm[1:2])
    got3 = P.proof_to_json(sp.prove_struct_split_witness_map(wm), wm[1:2])      # SURVEY 8(e) option 2 (three ranks or more)
    # an unsatisfying witness through option 2: rank 0 alone forms H, EVERY rank must raise ZK_ERR_DEGREE (no rank is left waiting
    # for coefficients), and the contexts prove again afterwards
    bad = wm.copy(); bad[7] = wm[8]
    all_raise = True
    if world >= 3:
        try:
            sp.prove_struct_split_witness_map(bad)
            all_raise = False
        except P.ZkError as e:
            all_raise = e.code == 7
        all_raise = all_raise and P.proof_to_json(sp.prove_struct_split_witness_map(wm), wm[1:2]) == expect
        # a LOCAL failure on a rank other than 0 (here: rank 1's chain cannot be queued): the agreement step makes every rank drop the
        # proof and raise -- nobody is left in a receive, a broadcast or the all-gather -- and the contexts prove again afterwards
        real = ctx.chain_submit
        if rank == 1:
            def failing(*a, **k):
                ctx.chain_submit = real
                raise P.ZkError(4, "injected failure on rank 1")
            ctx.chain_submit = failing
        try:
            sp.prove_struct_split_witness_map(wm)
            all_raise = False
        except P.ZkError as e:
            all_raise = all_raise and e.code == 4
        all_raise = all_raise and P.proof_to_json(sp.prove_struct_split_witness_map(wm), wm[1:2]) == expect
    q.put((rank, got == expect and got2 == expect and got3 == expect and shared and all_raise))
    dist.barrier()
    dist.destroy_process_group()


# (world, constraints): every shard keeps >= 64 witness indices, so that its A / B / L shards ride ONE shared witness sort
@pytest.mark.parametrize("world,nC", [(2, 254), (3, 254), (4, 254), (8, 1022)])
def test_sharded_prove_gloo(emul, world, nC):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, nC, port, emul, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(k, True) for k in range(world)]
