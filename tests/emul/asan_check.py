"""AddressSanitizer pass over the kernel logic of the CPU emulation build (GPU sanitizers are not available on the pool: the CPU build is where
indexing mistakes of the kernels can be caught).  Not part of the pytest suites (several minutes):

    make -C tests/emul asan
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 python tests/emul/asan_check.py

Covers the round-4 kernels: multi-exponentiations with 1 / 2 / 4 bucket planes x row / column segment widths x G1 / G2 x window sizes, a proof through the
synchronous and the queued entry points (merged H + L tail), a batch, a context with frugal tables, sharded partial sums.  Prints ALL True when every
result equals the oracle's; any ASan report aborts the run."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import oracle_lib as O
from ethsnarks_amd import prover as P, r1cs as R, fields as F
from helpers import rand_scalars
P.load_library(os.path.join(ROOT, "tests", "emul", "libzkhip_emul_asan.so"))
print(P._lib.zk_version(), flush=True)
ok = True
for plog in ("0", "1", "2"):
    os.environ["ZK_TEST_PLANES_LOG"] = plog
    for seg in ("4", "256"):
        os.environ["ZK_ROWCOL_SEG"] = seg
        for g2 in (False, True):
            for n, c in [(7, 0), (220, 3), (300, 8), (500, 9)]:
                sc = rand_scalars(n, n + 3, ones_every=5, zeros_every=7)
                bases = O.batch_mul(F.fr_to_mont(rand_scalars(n, 98)), g2=g2)
                s = F.fr_to_mont(sc)
                ok &= bool(np.array_equal(P.msm(bases, s, g2=g2, c=c), O.msm(bases, s, g2=g2)))
    print("planes", plog, ok, flush=True)
del os.environ["ZK_TEST_PLANES_LOG"]; del os.environ["ZK_ROWCOL_SEG"]
r, w = R.synthetic_chain(254, 1)
wm = F.fr_to_mont(w)
pk_o, _ = O.keygen(r, seed=31)
expect = O.prove(pk_o, r, wm)[0]
pk = P.ProvingKey.from_parts(**pk_o.parts())
ctx = P.ProverContext(pk, r, max_batch=3)
ok &= P.prove(ctx, wm) == expect
ctx.submit(wm); part, _ = ctx.collect(); ok &= P.proof_to_json(ctx.prove_combine(part), wm[1:2]) == expect
ok &= P.prove_batch(ctx, np.stack([wm, wm, wm])) == [expect] * 3
ctx.close()
os.environ["ZK_TABLE_BUDGET"] = "100000"
pk2 = P.ProvingKey.from_parts(**pk_o.parts())
c2 = P.ProverContext(pk2, r)
print(c2.info()["planes"], flush=True)
ok &= P.prove(c2, wm) == expect
parts = [P.ProverContext(pk2, r, shard_rank=k, shard_count=3).prove_partial(wm) for k in range(3)]
ok &= P.proof_to_json(c2.prove_combine(np.stack(parts)), wm[1:2]) == expect
print("ALL", ok, flush=True)
