#!/usr/bin/env python3
"""bench.py -- Groth16 proofs/s of the HIP backend on the reference's headline workload.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full proof (witness -> QAP coefficients -> four multi-exponentiations -> proof) of the
synthetic chain R1CS of SURVEY 8(d) with nC = 2^20 - 2 constraints (domain m = 2^20 exactly), nIn = 1,
on a real proving key produced by this library's GPU key generator (seeded toxic waste), key and
constraint system resident in HBM, witness handed over as a host buffer (its 33.5 MB upload is inside the
timed region).  N > 1 (launched by torch.distributed.run, one rank per GPU): proofs are independent units, so
the headline leg runs one prover per GPU with no data-path collective ("weak" scaling, value = all proofs of
all ranks / max-over-ranks time).  The MSM-sharded leg of north_star / BASELINE config 5 is timed too and
reported under "msm_sharded": the base ranges of the key are split over the ranks, every rank proves its
shard, the 640-byte partial results are exchanged with one RCCL all-gather and folded in rank order
("strong": the GPUs share each proof; the witness -> H pipeline is replicated).  --mode shard makes that
leg the headline instead.  --inflight K (default 3) keeps K
prover contexts per GPU busy through the asynchronous zk_prove_submit / zk_prove_collect pair, so the next
proof's witness upload and sort overlap the bucket-reduction tails of the current one; every step still
completes a full proof, and --inflight 1 gives the one-proof-at-a-time latency figure.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (k_msm_accumulate over G2,
the B-query); `cpu_baseline` is the CPU oracle (oracle/, a restatement of the reference prover --
libsnark itself cannot be built offline) timed on this host's cores on one proof of the same key.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MAD_LANE_OPS_PER_CLK_CU = 43.4  # measured v_mad_u64_u32 rate, tools/microbench (profiles/r01_microbench.txt)
MADS_PER_G2_MADD = 6 * 400 + 2 * 272 + 656   # = 3600, see the valu block below


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--logm", type=int, default=20, help="domain size 2^logm (nC = 2^logm - 2)")
    ap.add_argument("--mode", choices=["shard", "replicas"], default="replicas")
    ap.add_argument("--workload", choices=["chain", "merkle29", "mimc11"], default="chain",
                    help="chain: the headline synthetic R1CS (configs 2/3/5 by --logm); merkle29 / mimc11: BASELINE configs 4 / 1 "
                         "(real MiMC circuits from ethsnarks_amd.gadgets; latency-sized, not the headline)")
    ap.add_argument("--multi-exp-c", type=int, default=0)
    ap.add_argument("--inflight", type=int, default=3,
                    help="prover contexts kept in flight per GPU (1 = one synchronous proof at a time; 2 overlaps the "
                         "next proof's upload/sort with the current proof's bucket-reduction tails)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-logm", type=int, default=0, help="size of the CPU-baseline sample (default: same as --logm)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from ethsnarks_amd import prover as P, r1cs as R, fields as F

    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # rehearsal aid for a 1-GPU box: ZK_BENCH_REHEARSE=1 maps every rank to cuda:0 and exchanges over gloo
    rehearse = os.environ.get("ZK_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # "nccl" is RCCL on ROCm
        assert dist.get_world_size() == world, "--gpus must equal WORLD_SIZE"
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cpu") if (rehearse and world > 1) else torch.device("cuda", local_rank)

    P.load_library()                                   # raises if libzkhip.so is missing: no CPU fallback
    logm = args.logm
    nC = (1 << logm) - 2
    t0 = time.time()
    if args.workload == "chain":
        r1cs, w_ints = R.synthetic_chain(nC, 1)
        workload = "synthetic chain R1CS (SURVEY 8d), nC=2^%d-2=%d, nIn=1, V=%d, domain m=2^%d; real seeded Groth16 key" % (logm, nC, r1cs.V, logm)
    else:
        from ethsnarks_amd import gadgets as G
        r1cs, w_ints, _ = G.merkle_membership_circuit(29) if args.workload == "merkle29" else G.mimc_preimage_circuit(11)
        nC = r1cs.nC
        logm = r1cs.domain_size.bit_length() - 1
        workload = "%s (BASELINE config %s): nC=%d, nIn=%d, V=%d, domain m=2^%d; real seeded Groth16 key" % (
            "merkle_path_authenticator<MiMC_e7_hash_gadget> depth 29" if args.workload == "merkle29" else "MiMC-e7 hash preimage, 11 words",
            "4" if args.workload == "merkle29" else "1", nC, r1cs.nIn, r1cs.V, logm)
    wm = F.fr_to_mont(w_ints)
    t_circuit = time.time() - t0
    t0 = time.time()
    pk, vk = P.keygen(r1cs, seed=R.SEED_DEFAULT, device=local_rank)   # same seeded key on every rank
    t_keygen = time.time() - t0
    m = r1cs.domain_size

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_leg(shard):
        """W warm-up + K timed proofs in one parallelisation; returns (proofs/s, elapsed, kernel ms list, timings, json)"""
        ctxs = [P.ProverContext(pk, r1cs, multi_exp_c=args.multi_exp_c, device=local_rank,
                                shard_rank=rank if shard else 0, shard_count=world if shard else 1)
                for _ in range(max(1, args.inflight))]
        gather_buf = torch.empty((world, 640), dtype=torch.uint8, device=dev) if shard else None
        acc_b, pending, state = [], [], {"t": {}}

        def finish(slot):
            """collect slot's proof; sharded: one RCCL all-gather of the 640-byte partials, folded in rank order"""
            part, tm = ctxs[slot].collect()
            if shard:
                mine = torch.from_numpy(part.view(np.uint8).copy()).to(dev)
                dist.all_gather_into_tensor(gather_buf.view(-1), mine)
                part = gather_buf.cpu().numpy().reshape(-1).view(np.uint64)
            proof = ctxs[slot].prove_combine(part)
            state["t"] = tm
            acc_b.append(tm["acc_b"])
            return P.proof_to_json(proof, wm[1:2])

        def run(nsteps):
            js = None
            for i in range(nsteps):
                if len(pending) == len(ctxs):
                    js = finish(pending.pop(0))
                slot = i % len(ctxs)
                ctxs[slot].submit(wm)
                pending.append(slot)
            while pending:
                js = finish(pending.pop(0))
            return js

        if args.warmup:
            run(args.warmup)
        acc_b.clear()
        sync()
        t0 = time.perf_counter()
        js = run(args.steps)
        sync()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        proofs = args.steps * (world if (world > 1 and not shard) else 1)
        for c in ctxs:
            c.close()
        return proofs / elapsed, elapsed, list(acc_b), state["t"], js

    # N > 1: the headline leg runs independent provers per GPU ("weak": proofs are independent units, no
    # data-path collective); the MSM-sharded leg (north_star / config 5: base ranges over the GPUs + one RCCL
    # all-gather of 640-byte partials per proof, "strong") is timed as well and reported under "msm_sharded".
    mode = args.mode if world > 1 else "replicas"
    shard = world > 1 and mode == "shard"
    value, elapsed, acc_b, last_t, js = run_leg(shard)
    nB_local = pk.nB // world if shard else pk.nB
    sharded_extra = None
    if world > 1 and mode == "replicas":
        try:
            v2, e2, _, t2, js2 = run_leg(True)
            sharded_extra = {"value": round(v2, 4), "unit": "proofs/s", "scaling": "strong", "ms_per_step": round(1e3 * e2 / args.steps, 3),
                             "parallelism": "msm-shard%d + RCCL all-gather of 640 B partials" % world,
                             "matches_replica_proof": js2 == js}
        except Exception as e:                                  # the headline line must survive a failure of the extra leg
            sharded_extra = {"error": repr(e)[:300]}

    out = None
    if rank == 0:
        kern_ms = float(np.mean(acc_b)) if acc_b else float("nan")
        W = 254 // (args.multi_exp_c or P_pick_c(nB_local)) + 1
        alg_bytes = 160.0 * nB_local                                   # 128 B G2 base + 32 B scalar per pair
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        # multiplies (v_mad_u64_u32 lane-ops) of one G2 mixed addition as the kernel computes it: 6 Fq2 products x (4 x 64 + 2 x 72)
        # + 2 Fq2 squarings x (2 x 136) + one two-term Fq2 dot product x 2 x (4 x 64 + 72)   (bn254.hpp: Fq2::lmul / lsqr / lmul2)
        mads = float(MADS_PER_G2_MADD) * nB_local * W
        bytes_per_proof = proof_bytes(r1cs, pk, m)
        out = {
            "metric": "groth16_proofs_per_sec", "value": round(value, 4), "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "strong" if shard else "weak", "vs_baseline": None,
            "dtype": "u32 (8-limb 254-bit Montgomery integers)", "data": "synthetic",
            "constraints_per_sec": round(value * nC, 1),
            "config": {"workload": workload,
                       "parallelism": ("msm-shard%d+allgather640B" % world) if shard else ("replicas%d" % world if world > 1 else "1gpu"),
                       "multi_exp_c": args.multi_exp_c or P_pick_c(nB_local), "witness": "host buffer, H2D inside the timed region", "inflight": max(1, args.inflight)},
            "roofline": {"kernel": "k_msm_accumulate<G2, 1> (B-query bucket accumulation)", "bound": "hbm",
                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": pmc_traffic("k_msm_accumulate<G2, 1>", world if shard else 1),
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(kern_ms, 4),
                         "note": "VALU-integer bound kernel (no dense contraction, no MFMA): see valu. traffic > algorithmic bytes by design: "
                                 "each base is read through its W window multiples (msm.hpp), trading HBM bytes for 16x fewer bucket reductions"},
            "valu": {"kernel": "k_msm_accumulate<G2, 1>", "unit": "T mad/s (v_mad_u64_u32 lane-ops)",
                     "achieved": round(mads / (kern_ms * 1e-3) / 1e12, 3) if kern_ms > 0 else 0.0,
                     "peak": round(MAD_LANE_OPS_PER_CLK_CU * 256 * 2.4e9 / 1e12, 2),
                     "frac": round(mads / (kern_ms * 1e-3) / (MAD_LANE_OPS_PER_CLK_CU * 256 * 2.4e9), 4) if kern_ms > 0 else 0.0,
                     "peak_basis": "measured v_mad_u64_u32 issue rate (43.4 lane-ops/clk/CU, profiles/r01_microbench.txt) x 256 CU x 2.4 GHz; "
                                   "the multiplies are 2/3 of the kernel's VALU issue slots, the rest are carry/fold/select instructions",
                     "mads_per_mixed_addition": MADS_PER_G2_MADD},
            "proof_hbm": {"algorithmic_bytes_per_proof": bytes_per_proof,
                          "achieved_GBps": round(bytes_per_proof * value / 1e9, 3),
                          "frac_of_peak": round(bytes_per_proof * value / 1e9 / (world * HBM_PEAK_GBPS), 6)},
            "phases_ms_last_step": {k: round(v, 3) for k, v in last_t.items()},
            "setup_s": {"circuit": round(t_circuit, 2), "gpu_keygen": round(t_keygen, 2)},
        }
        if sharded_extra is not None:
            out["msm_sharded"] = sharded_extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity_vs_oracle"] = cpu_baseline(args, pk, r1cs, wm, js, logm)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def pmc_traffic(kernel, shards):
    """HBM bytes per launch of `kernel` from the committed PMC passes (2 x FETCH_SIZE + WRITE_SIZE, collected with
    rocprofv3 --pmc in separate runs, tools/pmc_summary.py); valid for the unsharded 2^20 workload only."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        return d["kernels"][kernel]["hbm_bytes_per_launch"] if shards == 1 else None
    except Exception:
        return None


def P_pick_c(n):
    """mirror of MsmShape::pick_c (ethsnarks_amd/csrc/msm.hpp)"""
    for c in range(17, 2, -1):
        if n * (254 // c + 1) >= (32 << (c - 1)):
            return c
    return 2


def proof_bytes(r1cs, pk, m):
    """each operand read once, each result written once."""
    V, nIn = r1cs.V, r1cs.nIn
    return (64 * (pk.nA + (m - 1) + (V - nIn)) + 128 * pk.nB + 32 * (pk.nA + pk.nB + (m - 1) + (V - nIn))
            + 36 * r1cs.nnz + 32 * (V + 1) + 3 * 32 * m + 7 * 2 * 32 * m + 4 * 32 * m)


def cpu_baseline(args, pk, r1cs, wm, gpu_json, logm):
    """The CPU oracle (restatement of r1cs_gg_ppzksnark_zok.tcc:451-550; libsnark unavailable offline)
    on this host's cores, one proof with the same key; also the parity check of the timed GPU proof."""
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    import ctypes as C
    import numpy as np
    import oracle_lib as O
    threads = min(O.cpu_share(), 64)
    O.lib().orc_set_threads(threads)
    parts = pk.parts()
    h = C.c_void_p()
    u64 = lambda a: O._p64(np.ascontiguousarray(a, dtype=np.uint64))
    u32 = lambda a: O._p32(np.ascontiguousarray(a, dtype=np.uint32))
    keep = {k: np.ascontiguousarray(v) for k, v in parts.items() if hasattr(v, "dtype")}
    rc = O.lib().orc_pk_from_parts(u64(keep["alpha_g1"]), u64(keep["beta_g1"]), u64(keep["beta_g2"]), u64(keep["delta_g1"]), u64(keep["delta_g2"]),
                                   C.c_uint32(parts["a_domain"]), C.c_uint32(len(keep["a_idx"])), u32(keep["a_idx"]), u64(keep["a_val"]),
                                   C.c_uint32(parts["b_domain"]), C.c_uint32(len(keep["b_idx"])), u32(keep["b_idx"]), u64(keep["b_val"]),
                                   C.c_uint32(len(keep["H"])), u64(keep["H"]), C.c_uint32(len(keep["L"])), u64(keep["L"]), C.byref(h))
    assert rc == 0
    opk = O.PK(h)
    t0 = time.perf_counter()
    js, phases = O.prove(opk, r1cs, wm)
    dt = time.perf_counter() - t0
    base = {"value": round(1.0 / dt, 5), "unit": "proofs/s", "cores": threads, "kind": "port",
            "sample": "1 proof of the same 2^%d circuit and key (%.1f s); CPU restatement of r1cs_gg_ppzksnark_zok.tcc:451-550, OpenMP" % (logm, dt),
            "phases_s": [round(p, 3) for p in phases]}
    return base, (js == gpu_json)


if __name__ == "__main__":
    main()
