"""dev tool: why is the SECOND set of contexts of a process slower than the first at 2^20?  (bench.py's second leg lost 6 %
whichever witness mode it ran.)  Runs the same pipelined leg several times under different lifetimes:
  A  fresh contexts (first set of the process)
  B  fresh contexts after A's were closed (tables rebuilt, streams re-created)
  C  fresh contexts while a keeper context holds the device tables alive (streams re-created only)
  D  the SAME contexts as C again (nothing re-created)
usage: dev_second_leg.py [logm] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import torch
from ethsnarks_amd import prover as P, r1cs as R, fields as F
logm = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
torch.cuda.set_device(0)
P.load_library(os.environ.get("ZK_LIB") or None)
r, w = R.synthetic_chain((1 << logm) - 2, 1)
wm = F.fr_to_mont(w)
pk, _ = P.keygen(r, seed=R.SEED_DEFAULT)
d_w = torch.from_numpy(np.ascontiguousarray(wm).view(np.int64).copy()).cuda()

def leg(ctxs, n, mode="resident"):
    pending, staged = [], [False] * len(ctxs)
    def fin(s):
        part, tm = ctxs[s].collect(); ctxs[s].prove_combine(part)
    for i in range(n):
        if len(pending) == len(ctxs): fin(pending.pop(0))
        s = i % len(ctxs)
        if mode == "resident": ctxs[s].submit_resident(d_w.data_ptr())
        elif mode == "plain": ctxs[s].submit(wm)
        else:
            if staged[s]: ctxs[s].submit_staged()
            else: ctxs[s].submit(wm)
            ctxs[s].stage(wm); staged[s] = True
        pending.append(s)
    while pending: fin(pending.pop(0))
    if mode == "staged":                      # leave no staged witness behind
        for s in range(len(ctxs)):
            if staged[s]: ctxs[s].submit_staged(); fin(s)

def timed(ctxs, mode):
    leg(ctxs, 6, mode); torch.cuda.synchronize()
    t0 = time.perf_counter(); leg(ctxs, steps, mode); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / (steps + (len(ctxs) if mode == "staged" else 0))

mk = lambda: [P.ProverContext(pk, r) for _ in range(3)]
c = mk(); print("A first set            resident %.3f  staged %.3f  plain %.3f  resident %.3f ms/proof" % (timed(c, "resident"), timed(c, "staged"), timed(c, "plain"), timed(c, "resident")), flush=True)
for x in c: x.close()
c = mk(); print("B second set (rebuilt) resident %.3f  staged %.3f  plain %.3f ms/proof" % (timed(c, "resident"), timed(c, "staged"), timed(c, "plain")), flush=True)
keeper = c[0]
for x in c[1:]: x.close()
c = mk(); print("C third set (tables kept alive by a keeper context) resident %.3f  staged %.3f ms/proof" % (timed(c, "resident"), timed(c, "staged")), flush=True)
print("D same contexts again  resident %.3f  staged %.3f ms/proof" % (timed(c, "resident"), timed(c, "staged")), flush=True)
keeper.close()
print("E keeper closed        resident %.3f ms/proof" % timed(c, "resident"), flush=True)
