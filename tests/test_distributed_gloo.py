"""N > 1 path on CPU: world_size-2 and -4 gloo ranks, each with a sharded prover context (CPU emulation build of
the HIP sources, shipped launch shapes), exchange their 640-byte partials with all_gather -- through the same
device-buffer entry points the GPU path uses -- and must reproduce the oracle's proof."""
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


def _worker(rank, world, port, emul_so, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import oracle_lib as O
    from ethsnarks_amd import prover as P, r1cs as R, fields as F
    from ethsnarks_amd.sharded import ShardedProver
    P.load_library(emul_so)
    r, w = R.synthetic_chain(254, 1)
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
    q.put((rank, got == expect and got2 == expect and got3 == expect and shared and all_raise))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_prove_gloo(emul, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, emul, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(k, True) for k in range(world)]
