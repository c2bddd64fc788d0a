"""dev tool: host-side cost of zk_prove_submit / zk_prove_collect at 2^20 with three contexts in flight (is the host or the GPU the limit?)"""
import os, sys, time
sys.path[:0] = [os.getcwd()]
import torch; torch.zeros(1).cuda()
import numpy as np
from ethsnarks_amd import prover as P, r1cs as R, fields as F
r, w = R.synthetic_chain((1 << 20) - 2, 1); wm = F.fr_to_mont(w)
pk, vk = P.keygen(r, seed=R.SEED_DEFAULT)
ctxs = [P.ProverContext(pk, r) for _ in range(3)]
for c in ctxs: c.submit(wm); c.collect()
ts=[]; tc=[]
pend=[]
t0=time.perf_counter()
for i in range(30):
    if len(pend)==3:
        t=time.perf_counter(); pend.pop(0).collect(); tc.append(time.perf_counter()-t)
    c=ctxs[i%3]; t=time.perf_counter(); c.submit(wm); ts.append(time.perf_counter()-t); pend.append(c)
while pend: pend.pop(0).collect()
print("host submit ms avg %.3f min %.3f; collect avg %.3f; per proof %.3f" % (1e3*np.mean(ts),1e3*np.min(ts),1e3*np.mean(tc),1e3*(time.perf_counter()-t0)/30))
a=np.empty_like(wm); t=time.perf_counter(); np.copyto(a, wm); print("numpy copy of the witness: %.3f ms"%(1e3*(time.perf_counter()-t)))
