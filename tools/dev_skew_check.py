import sys, time, os
sys.path[:0] = [os.getcwd(), "tests", "oracle"]
import numpy as np
import oracle_lib as O
from ethsnarks_amd import prover as P, fields as F, r1cs as R
from helpers import tiled_bases
P.load_library()
n = 1 << 20
bases = tiled_bases(O, n, distinct=8192)
rng = R.SplitMix64(3)
for name, sc in (("all ones", [1]*n), ("all equal", [rng.fr()]*n), ("half zero half small", [0,3]*(n//2)), ("8-bit values", [ (i*2654435761) % 256 for i in range(n)])):
    s = F.fr_to_mont(sc)
    t = time.time(); got = P.msm(bases, s); dt = time.time() - t
    exp = O.msm(bases, s)
    print("%-22s gpu call %.2fs (incl. table precompute) parity %s" % (name, dt, np.array_equal(got, exp)), flush=True)
