"""child process of test_gpu_parity.py::test_sharded_device_exchange_over_rccl_one_rank (needs a GPU)"""
import os
import socket
import sys
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")                                  # torch owns the HIP runtime of this process first, as in bench.py
import oracle_lib as O
from ethsnarks_amd import prover as hip, r1cs as R, fields as F
from ethsnarks_amd.sharded import ShardedProver

hip.load_library()
r, w = R.synthetic_chain((1 << 12) - 2, 1)
wm = F.fr_to_mont(w)
pk_o, _ = O.keygen(r, seed=6)
expect, _ = O.prove(pk_o, r, wm)
pk = hip.ProvingKey.from_parts(**pk_o.parts())
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    c0 = hip.ProverContext(pk, r, shard_rank=0, shard_count=2)
    c1 = hip.ProverContext(pk, r, shard_rank=1, shard_count=2)
    sp = ShardedProver(c0, dist, torch.device("cuda", 0))
    via_device = sp.prove_struct(wm)                            # collect_device + all_gather (1 rank) + combine_device
    host_part = c0.prove_partial(wm)
    assert bytes(via_device) == bytes(c0.prove_combine(host_part)), "device path differs from host path"
    both = np.stack([host_part, c1.prove_partial(wm)])
    assert hip.proof_to_json(c0.prove_combine(both), wm[1:2]) == expect
    d_parts = torch.from_numpy(both.view(np.uint8).reshape(2, 640).copy()).cuda()      # the same two records gathered in device memory
    assert hip.proof_to_json(c0.prove_combine_device(d_parts.data_ptr(), 2), wm[1:2]) == expect
    c0.close(); c1.close()
    print("RCCL_ONE_RANK_OK")
finally:
    dist.destroy_process_group()
