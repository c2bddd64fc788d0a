#!/usr/bin/env python3
"""bench.py -- Groth16 proofs/s of the HIP backend on the reference's headline workload.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full proof (witness -> QAP coefficients -> four multi-exponentiations -> proof) of the
synthetic chain R1CS of SURVEY 8(d) with nC = 2^20 - 2 constraints (domain m = 2^20 exactly), nIn = 1,
on a real proving key produced by this library's GPU key generator (seeded toxic waste).  Key and constraint
system are resident in HBM; the WITNESS lies in pinned host memory when the timed region starts and every proof's
33.5 MB H2D copy is inside it (SURVEY 8(d)'s metric: `value`; the prover pipeline copies a context's next witness on
its copy stream while the current proof runs: zk_prove_stage_pinned / zk_prove_submit_staged).  The other ways to
hand a witness over are timed on legs of the same length and reported under "witness_modes": resident in HBM (the
PCIe-free rate, never `value`), pinned with the copy on the proof's own stream (zk_prove_submit_pinned), and a
pageable host buffer with and without the double buffering (zk_prove_submit / zk_prove_stage).

N > 1 (launched by torch.distributed.run, one rank per GPU): proofs are independent units, so the
headline leg runs one prover per GPU with no data-path collective ("weak" scaling, value = all proofs of
all ranks / max-over-ranks time).  The MSM-sharded prover of north_star / BASELINE config 5 is timed too
and reported under "msm_sharded" (same circuit: the GPUs share each proof) and "msm_sharded_2p22"
(config 5's size, 2^22 constraints): the base ranges of the key are split over the ranks, every rank
proves its shard, the 640-byte partial results are exchanged with ONE RCCL all-gather on device buffers
and folded in rank order ("strong"; the witness -> H pipeline is replicated).  --mode shard makes the
sharded leg the headline instead.  --inflight K (default 3) keeps K prover contexts per GPU busy through
the asynchronous zk_prove_submit / zk_prove_collect pair, so the next proof's sorts overlap the
bucket-reduction tails of the current one; every step still completes a full proof, and --inflight 1
gives the one-proof-at-a-time latency figure.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (k_msm_accumulate over G2, the
B-query) from HIP events recorded around it on its own stream in this run; `cpu_baseline` is the CPU
oracle (oracle/, a restatement of the reference prover -- libsnark itself cannot be built offline) timed
on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WITNESS_MODES = ["host", "resident", "host-direct", "host-pageable", "host-pageable-staged"]
WITNESS_NOTE = {
    "host": "pinned host buffer (SURVEY 8(d)), every witness's H2D copy inside the timed region, double-buffered: a context's NEXT witness is copied "
            "on the copy stream while its current proof runs (zk_prove_stage_pinned / zk_prove_submit_staged)",
    "resident": "resident in HBM when the timed region starts (no PCIe traffic)",
    "host-direct": "pinned host buffer, one zk_prove_submit_pinned per proof: the H2D copy heads the proof's own stream (on its critical path)",
    "host-pageable": "pageable host buffer: plain zk_prove_submit copies it into pinned staging memory, then H2D, both inside the timed region",
    "host-pageable-staged": "pageable host buffer, double-buffered: zk_prove_stage copies a context's NEXT witness (memcpy into pinned staging + H2D) while its current proof runs",
}
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MAD_LANE_OPS_PER_CLK_CU = 43.4  # measured v_mad_u64_u32 rate, tools/microbench (profiles/r01_microbench.txt)
MADS_PER_G2_MADD = 6 * 400 + 2 * 272 + 656   # = 3600: 6 Fq2 products (2 x (2 x 64 + 72)), 2 Fq2 squarings (2 x 136), one 2-term Fq2 dot product (2 x (4 x 64 + 72))
PMC_PROFILE = os.path.join("profiles", "r04_pmc_traffic.json")
KERNEL_SOURCES = ["msm.hpp", "msm_impl.hpp", "bn254.hpp", "fips_asm.hpp"]     # what k_msm_accumulate is compiled from (ethsnarks_amd/csrc)


def kernel_sources_sha16():
    """identity of the accumulation kernels' sources: the committed PMC passes name the sources they were taken with, and
    roofline.traffic is only quoted while they are still the ones this library was built from (tools/pmc_summary.py writes the same digest)"""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "ethsnarks_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


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
    ap.add_argument("--one-stream", action="store_true", help="zk_config.schedule = ZK_SCHED_ONE_STREAM: every launch of a context on one stream (small circuits, many contexts)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="prover contexts kept in flight per GPU (1 = one synchronous proof at a time)")
    ap.add_argument("--batch", type=int, default=1,
                    help="proofs per launch sequence (zk_prove_batch): a step then proves this many witnesses of the circuit; "
                         "what the latency-sized workloads need (one small proof alone is ~50 launches of latency-bound kernels)")
    ap.add_argument("--witness", choices=WITNESS_MODES, default="host",
                    help="where the witness lives when the timed region starts.  host (the default, SURVEY 8(d)): pinned host memory, "
                         "H2D inside the timed region, double-buffered (zk_prove_stage_pinned / zk_prove_submit_staged); resident: already in HBM; "
                         "host-direct: pinned, one zk_prove_submit_pinned per proof; host-pageable: pageable buffer through plain zk_prove_submit; "
                         "host-pageable-staged: zk_prove_stage")
    ap.add_argument("--batch-identical", action="store_true", help="--batch k: k copies of ONE witness instead of k different ones (shows what identical digits / gather addresses are worth)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary legs (host-witness rate, sharded legs, kernel sum)")
    ap.add_argument("--cpu-1t-logm", type=int, default=18, help="size of the one-thread CPU sample (a full 2^20 proof takes more than a minute on one core; 2^18: ~17 s)")
    ap.add_argument("--shard-logm", type=int, default=22, help="size of the config-5 sharded leg at N > 1")
    ap.add_argument("--extras-timeout", type=int, default=420, help="N > 1: seconds the secondary legs may take before the headline line is printed without them")
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
    gloo = rehearse and world > 1

    P.load_library(os.environ.get("ZK_LIB") or None)  # raises if libzkhip.so is missing: no CPU fallback (ZK_LIB: A/B of two builds of the library)
    devinfo = P.device_info(local_rank)

    def make_workload(workload, logm):
        t0 = time.time()
        if workload == "chain":
            nC = (1 << logm) - 2
            r1cs, w_ints = R.synthetic_chain(nC, 1)
            name = "synthetic chain R1CS (SURVEY 8d), nC=2^%d-2=%d, nIn=1, V=%d, domain m=2^%d; real seeded Groth16 key" % (logm, nC, r1cs.V, logm)
        else:
            from ethsnarks_amd import gadgets as G
            r1cs, w_ints, _ = G.merkle_membership_circuit(29) if workload == "merkle29" else G.mimc_preimage_circuit(11)
            name = "%s (BASELINE config %s): nC=%d, nIn=%d, V=%d, domain m=2^%d; real seeded Groth16 key" % (
                "merkle_path_authenticator<MiMC_e7_hash_gadget> depth 29" if workload == "merkle29" else "MiMC-e7 hash preimage, 11 words",
                "4" if workload == "merkle29" else "1", r1cs.nC, r1cs.nIn, r1cs.V, r1cs.domain_size.bit_length() - 1)
        wm = F.fr_to_mont(w_ints)
        t_circuit = time.time() - t0
        t0 = time.time()
        pk, vk = P.keygen(r1cs, seed=R.SEED_DEFAULT, device=local_rank)   # same seeded key on every rank
        return r1cs, wm, pk, name, t_circuit, time.time() - t0

    r1cs, wm, pk, workload, t_circuit, t_keygen = make_workload(args.workload, args.logm)
    if rank == 0:
        print("[bench] circuit built in %.1f s, key generated in %.1f s" % (t_circuit, t_keygen), file=sys.stderr, flush=True)
    nC, m = r1cs.nC, r1cs.domain_size
    logm = m.bit_length() - 1

    # which GPU every rank really sits on: gathered once, asserted distinct, reported in the line (SCALE can check it)
    rccl_info = None
    if world > 1:
        mine = "%s|%s" % (os.uname().nodename, P.device_pci_bus_id(local_rank))
        ids = [None] * world
        dist.all_gather_object(ids, mine)
        rccl_info = {"world": world, "backend": dist.get_backend(), "device_ids": ids}
        if not rehearse:
            assert len(set(ids)) == world, "ranks share a GPU: %r" % (ids,)

    def batch_witnesses(r1cs, wm, kb):
        """kb x (V + 1) x 4: the witnesses of one launch sequence.  kb DISTINCT satisfying witnesses of the circuit (other seeds /
        leaves / preimages: identical witnesses would share digits, bucket occupancy and gather addresses); --batch-identical tiles one"""
        first = np.ascontiguousarray(wm).reshape(1, -1, 4)
        if kb == 1:
            return first
        if args.batch_identical:
            return np.ascontiguousarray(np.tile(first, (kb, 1, 1)))
        out = [first[0]]
        for p in range(1, kb):
            if args.workload == "chain":
                _, w = R.synthetic_chain(r1cs.nC, 1, seed=R.SEED_DEFAULT + p)
            else:
                from ethsnarks_amd import gadgets as G
                if args.workload == "merkle29":
                    _, w, _ = G.merkle_membership_circuit(29, leaf=1000 + p, address=(0x2545F491 * p) & ((1 << 29) - 1), path=[G.merkle_unique(d, p) for d in range(29)])
                else:
                    _, w, _ = G.mimc_preimage_circuit(11, seed=7 + p)
            out.append(F.fr_to_mont(w))
        return np.ascontiguousarray(np.stack(out))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from ethsnarks_amd.sharded import ShardedProver

    all_acc_b = []          # every B-query accumulation this process launched (warm-ups and secondary legs too): what a profiler averages

    def run_leg(pk, r1cs, wm, shard, steps, warmup, witness):
        """`warmup` untimed + `steps` timed proofs in one parallelisation"""
        kb = max(1, args.batch) if not shard else 1
        ctxs = [P.ProverContext(pk, r1cs, multi_exp_c=args.multi_exp_c, device=local_rank,
                                shard_rank=rank if shard else 0, shard_count=world if shard else 1, max_batch=kb, one_stream=args.one_stream)
                for _ in range(max(1, args.inflight))]
        wmk = batch_witnesses(r1cs, wm, kb)                        # the batch: kb DISTINCT satisfying witnesses of the circuit, contiguous
        d_w = torch.from_numpy(wmk.view(np.int64).copy()).cuda() if witness == "resident" else None
        pinned = None
        if witness in ("host", "host-direct"):                      # the witness lies in pinned host memory
            pinned = P.PinnedBuffer(wmk.nbytes)
            pinned.array[:] = wmk.reshape(-1)
        # (under the one-GPU rehearsal, ZK_BENCH_REHEARSE=1, ShardedProver stages every exchanged buffer through the host: gloo moves host memory only)
        sharded = [ShardedProver(c, dist, torch.device("cuda", local_rank)) for c in ctxs] if shard else None
        acc_b, pending, state = [], [], {"t": {}}
        use_stage = witness in ("host", "host-pageable-staged") and not shard
        staged = [False] * len(ctxs)

        def finish(slot):
            """collect slot's proof; sharded: one all-gather of the 640-byte partials (device buffers), folded in rank order"""
            if shard:
                proof, tm = sharded[slot].finish()
            elif kb > 1:
                parts, tm = ctxs[slot].collect_batch(kb)
                proof = ctxs[slot].prove_combine(parts[kb - 1])
            else:
                part, tm = ctxs[slot].collect()
                proof = ctxs[slot].prove_combine(part)
            state["t"] = tm
            acc_b.append(tm["acc_b"])
            all_acc_b.append(tm["acc_b"])
            return P.proof_to_json(proof, wmk[kb - 1].reshape(-1, 4)[1:1 + r1cs.nIn])

        def run(nsteps):
            js = None
            for i in range(nsteps):
                if len(pending) == len(ctxs):
                    js = finish(pending.pop(0))
                slot = i % len(ctxs)
                if use_stage:
                    # host witness, double-buffered upload: this context's NEXT witness was staged while its previous proof ran
                    if staged[slot]:
                        ctxs[slot].submit_staged()
                    elif pinned is not None:
                        ctxs[slot].submit_pinned(pinned, k=kb)
                    elif kb > 1:
                        ctxs[slot].submit_batch(wmk, k=kb)
                    else:
                        ctxs[slot].submit(wmk[0])
                    if pinned is not None:
                        ctxs[slot].stage_pinned(pinned, k=kb)
                    else:
                        ctxs[slot].stage(wmk.reshape(kb, -1, 4) if kb > 1 else wmk[0])
                    staged[slot] = True
                elif pinned is not None:
                    ctxs[slot].submit_pinned(pinned, k=kb)
                elif kb > 1:
                    ctxs[slot].submit_batch(wmk, device_ptr=d_w.data_ptr() if d_w is not None else None, k=kb)
                elif d_w is not None:
                    ctxs[slot].submit_resident(d_w.data_ptr())
                else:
                    ctxs[slot].submit(wmk[0])
                pending.append(slot)
            while pending:
                js = finish(pending.pop(0))
            return js

        if warmup:
            run(warmup)
        acc_b.clear()
        sync()
        n0 = P.launch_count()
        t0 = time.perf_counter()
        js = run(steps)
        sync()
        elapsed = time.perf_counter() - t0
        launches = (P.launch_count() - n0) / float(steps * kb)
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if gloo else torch.device("cuda", local_rank))
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        proofs = steps * kb * (world if (world > 1 and not shard) else 1)
        info = ctxs[0].info()
        ksum = None
        if rank == 0 and not args.no_extras and not shard and kb == 1:
            # one more (untimed) proof with every launch bracketed by HIP events: sum of kernel durations per proof
            P.profile_begin()
            ctxs[0].submit_resident(d_w.data_ptr()) if d_w is not None else ctxs[0].submit(wmk[0])
            _, t_alone = ctxs[0].collect()
            all_acc_b.append(t_alone["acc_b"])
            s_ms, n_l, per = P.profile_end()
            top = sorted(per.items(), key=lambda kv: -kv[1][1])[:6]
            ksum = {"kernel_ms_sum_one_proof_alone": round(s_ms, 3), "launches": n_l, "acc_b_ms_alone": round(t_alone["acc_b"], 4),
                    "top": {k.strip("()"): {"calls": c, "ms": round(v, 3)} for k, (c, v) in top}}
        for c in ctxs:
            c.close()
        if pinned is not None:
            pinned.free()
        return {"value": proofs / elapsed, "elapsed": elapsed, "acc_b": list(acc_b), "timings": state["t"], "json": js,
                "launches_per_proof": launches, "info": info, "kernel_sum": ksum}

    # N > 1: the headline leg runs independent provers per GPU ("weak": proofs are independent units, no data-path
    # collective); the MSM-sharded legs (north_star / config 5) are timed as well and reported under named keys.
    mode = args.mode if world > 1 else "replicas"
    shard = world > 1 and mode == "shard"
    head = run_leg(pk, r1cs, wm, shard, args.steps, args.warmup, args.witness)
    value, elapsed, acc_b, last_t, js = head["value"], head["elapsed"], head["acc_b"], head["timings"], head["json"]
    extras = {}

    def run_extras():
        if args.no_extras:
            return
        if True:
            short = max(3, min(args.steps, 10))
            modes = {}
            for other in WITNESS_MODES:                             # the other ways to hand the witness over, legs as long as the headline leg
                if other == args.witness or shard or world > 1:     # (a one-GPU property: the scaling runs spend their time on the sharded legs)
                    continue
                try:
                    h2 = run_leg(pk, r1cs, wm, shard, args.steps, args.warmup, other)
                    modes[other] = {"value": round(h2["value"], 4), "unit": "proofs/s", "steps": args.steps,
                                    "ms_per_step": round(1e3 * h2["elapsed"] / args.steps, 3), "witness": WITNESS_NOTE[other],
                                    "same_proof": h2["json"] == js}
                except Exception as e:
                    modes[other] = {"error": repr(e)[:300]}
            if modes:
                extras["witness_modes"] = modes
            if world > 1 and mode == "replicas":
                try:
                    s2 = run_leg(pk, r1cs, wm, True, short, 2, args.witness)
                    extras["msm_sharded"] = {"value": round(s2["value"], 4), "unit": "proofs/s", "scaling": "strong", "steps": short,
                                             "ms_per_step": round(1e3 * s2["elapsed"] / short, 3),
                                             "parallelism": "msm-shard%d + RCCL all-gather of 640 B partials (device buffers)" % world,
                                             "matches_replica_proof": s2["json"] == js}
                except Exception as e:                              # the headline line must survive a failure of an extra leg
                    extras["msm_sharded"] = {"error": repr(e)[:300]}
                if world >= 3:                                      # SURVEY 8(e) option 2: A / B / C transform chains on three ranks, one proof at a time
                    try:
                        ctx2 = P.ProverContext(pk, r1cs, multi_exp_c=args.multi_exp_c, device=local_rank, shard_rank=rank, shard_count=world)
                        sp2 = ShardedProver(ctx2, dist, torch.device("cuda", local_rank))
                        lat = {}
                        for name, fn in (("split_witness_map", sp2.prove_struct_split_witness_map), ("replicated_witness_map", sp2.prove_struct)):
                            fn(wm); sync()
                            t0 = time.perf_counter()
                            for _ in range(short):
                                pr = fn(wm)
                            sync()
                            lat[name] = {"ms_per_proof": round(1e3 * (time.perf_counter() - t0) / short, 3), "matches_replica_proof": P.proof_to_json(pr, wm[1:1 + r1cs.nIn]) == js}
                        extras["msm_sharded_latency"] = dict(lat, note="one sharded proof at a time (no pipelining): witness map replicated on every rank vs its A / B / C chains on ranks 0-2 (SURVEY 8(e) option 2)")
                        ctx2.close()
                    except Exception as e:
                        extras["msm_sharded_latency"] = {"error": repr(e)[:300]}
                if args.workload == "chain" and args.shard_logm and args.shard_logm != logm:
                    try:
                        r5, w5, pk5, name5, _, _ = make_workload("chain", args.shard_logm)
                        s5 = run_leg(pk5, r5, w5, True, max(3, short // 2), 1, args.witness)
                        extras["msm_sharded_2p%d" % args.shard_logm] = {
                            "value": round(s5["value"], 4), "unit": "proofs/s", "scaling": "strong", "steps": max(3, short // 2),
                            "ms_per_step": round(1e3 * s5["elapsed"] / max(3, short // 2), 3), "constraints_per_sec": round(s5["value"] * r5.nC, 1),
                            "workload": name5, "parallelism": "msm-shard%d + RCCL all-gather of 640 B partials (device buffers)" % world}
                        pk5.close()
                    except Exception as e:
                        extras["msm_sharded_2p%d" % args.shard_logm] = {"error": repr(e)[:300]}


    out = None
    if rank == 0:
        info = head["info"]
        nB_local = pk.nB // world if shard else pk.nB
        kern_ms = float(np.mean(acc_b)) if acc_b else float("nan")
        W = info["B"]["W"]                                              # the windows the B-query context really uses (zk_ctx_info)
        kb = max(1, args.batch) if not shard else 1
        alg_bytes = (128.0 + 32.0 * kb) * nB_local                     # one launch: 128 B per G2 base + 32 B per scalar of each of its kb proofs
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        mads = float(MADS_PER_G2_MADD) * nB_local * W * kb
        mad_peak = MAD_LANE_OPS_PER_CLK_CU * devinfo["compute_units"] * devinfo["clock_mhz"] * 1e6
        bytes_per_proof = proof_bytes(r1cs, pk, m)
        traffic, traffic_src = pmc_traffic("k_msm_accumulate<G2, 1>", args.workload, logm, world if shard else 1)
        out = {
            "metric": "groth16_proofs_per_sec", "value": round(value, 4), "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,      # a step = config.proofs_per_step proofs
            "scaling": "strong" if shard else "weak", "vs_baseline": None,
            "dtype": "u32 (8-limb 254-bit Montgomery integers)", "data": "synthetic",
            "constraints_per_sec": round(value * nC, 1),
            "config": {"workload": workload,
                       "parallelism": ("msm-shard%d+allgather640B" % world) if shard else ("replicas%d" % world if world > 1 else "1gpu"),
                       "multi_exp_c": info["B"]["c"], "windows": W,
                       "tables": {"bytes": info["table_bytes"], "planes": info["planes"],
                                  "note": "window-multiple tables of the key; planes > 1 = memory-frugal layout (every planes-th window tabulated: the full tables did not fit the device)"},
                       "witness": WITNESS_NOTE[args.witness], "witness_bytes_h2d_per_proof": 0 if args.witness == "resident" else 32 * (r1cs.V + 1),
                       "inflight": max(1, args.inflight), "schedule": "one stream per context" if args.one_stream else "overlap (five streams per context)", "proofs_per_step": max(1, args.batch) if not shard else 1,
                       "batch_witnesses": ("identical" if args.batch_identical else "distinct") if args.batch > 1 else None, "device": devinfo},
            "roofline": {"kernel": "k_msm_accumulate<G2, 1> (B-query bucket accumulation)", "bound": "hbm",
                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(kern_ms, 4),
                         "note": "VALU-integer bound kernel (no dense contraction, no MFMA): see valu. traffic > algorithmic bytes by design: "
                                 "each base is read through its W window multiples (msm.hpp), trading HBM bytes for 16x fewer bucket reductions"},
            "valu": {"kernel": "k_msm_accumulate<G2, 1>", "unit": "T mad/s (v_mad_u64_u32 lane-ops)",
                     "achieved": round(mads / (kern_ms * 1e-3) / 1e12, 3) if kern_ms > 0 else 0.0,
                     "peak": round(mad_peak / 1e12, 2),
                     "frac": round(mads / (kern_ms * 1e-3) / mad_peak, 4) if kern_ms > 0 else 0.0,
                     "peak_basis": "measured v_mad_u64_u32 issue rate (43.4 lane-ops/clk/CU, profiles/r01_microbench.txt) x %d CU x %.1f GHz (device properties); "
                                   "measured on MI355X the kernel's time is the SUM of its instructions' issue times (multiplies ~5.9 cycles, carry adds ~2.4: "
                                   "tools/mulbench.cpp, DESIGN section 4), so the multiplies alone cap it at ~0.65 of this peak" % (devinfo["compute_units"], devinfo["clock_mhz"] / 1e3),
                     "mads_per_mixed_addition": MADS_PER_G2_MADD},
            "valu_issue": valu_issue(args.workload, logm, world if shard else 1, 1e3 * elapsed / args.steps / max(1, kb), devinfo),
            "proof_hbm": {"algorithmic_bytes_per_proof": bytes_per_proof,
                          "achieved_GBps": round(bytes_per_proof * value / 1e9, 3),
                          "frac_of_peak": round(bytes_per_proof * value / 1e9 / (world * HBM_PEAK_GBPS), 6)},
            "phases_ms_last_step": {k: round(v, 3) for k, v in last_t.items()},
            "launches_per_proof": round(head["launches_per_proof"], 1),
            "kernel_sum": head["kernel_sum"],
            "setup_s": {"circuit": round(t_circuit, 2), "gpu_keygen": round(t_keygen, 2)},
        }
        if rccl_info is not None:
            out["rccl"] = rccl_info
        if out["kernel_sum"] is not None:
            out["kernel_sum"]["note"] = ("HIP-event durations of every launch of one proof run alone, overlapping tails counted each; the timed steps keep %d proofs in flight, "
                                         "so ms_per_step (%.2f) is below this sum" % (max(1, args.inflight), out["ms_per_step"]))
        if out["kernel_sum"] is not None and out["kernel_sum"].get("acc_b_ms_alone", 0) > 0:
            alone = out["kernel_sum"]["acc_b_ms_alone"]
            out["roofline"]["alone"] = {"avg_launch_ms": alone, "achieved": round(alg_bytes / (alone * 1e-3) / 1e9, 3),
                                        "frac": round(alg_bytes / (alone * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
                                        "valu_frac": round(mads / (alone * 1e-3) / mad_peak, 4),
                                        "note": "the same kernel when one proof runs alone; in the timed steps the accumulations of %d proofs share the machine "
                                                "(low-priority stream) with each other and with the sorts / transforms, which stretches every launch" % max(1, args.inflight)}
    # secondary legs.  At N > 1 they contain collectives this builder could only rehearse (gloo / one RCCL rank): a watchdog
    # makes sure the headline line is printed even if one of them stalls.
    def bail():
        if rank == 0:
            out.update(extras)
            out["extras_aborted"] = "secondary legs did not finish within %d s; headline unaffected" % args.extras_timeout
            print(json.dumps(out), flush=True)
        os._exit(3)         # a process that has touched the GPU and bails out of a stalled collective must not look like a clean run
    watchdog = None
    if world > 1 and not args.no_extras:
        import threading
        watchdog = threading.Timer(args.extras_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
    run_extras()
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        out.update(extras)
        if all_acc_b and not shard:
            out["roofline"]["all_launches"] = {"launches": len(all_acc_b), "avg_launch_ms": round(float(np.mean(all_acc_b)), 4),
                                               "note": "every launch of the kernel in this process (warm-ups, secondary legs and the single-proof pass included): the average a `rocprofv3 --kernel-trace --stats` of this command reports"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity_vs_oracle"] = cpu_baseline(args, P, R, F, pk, r1cs, wm, js, logm, local_rank)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def pmc_traffic(kernel, workload, logm, shards):
    """HBM bytes per launch of `kernel` from the committed PMC passes of THIS round's kernels (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate runs, tools/pmc_summary.py).  Only returned when the profiled configuration is the one
    being run (workload, domain size, unsharded); otherwise null."""
    try:
        d = json.load(open(os.path.join(ROOT, PMC_PROFILE)))
        cfg = d.get("config", {})
        if d.get("kernel_sources_sha16") != kernel_sources_sha16():
            return None, "%s (commit %s) was taken with other kernel sources than this build's: traffic not quoted" % (PMC_PROFILE, d.get("commit", "?"))
        if cfg.get("workload") == workload and cfg.get("logm") == logm and cfg.get("shards", 1) == shards:
            return d["kernels"][kernel]["hbm_bytes_per_launch"], "%s (PMC passes of commit %s, kernel sources %s)" % (PMC_PROFILE, d.get("commit", "?"), d.get("kernel_sources_sha16"))
        return None, "no PMC profile for this configuration (%s was taken at workload=%s, logm=%s, shards=%s)" % (
            PMC_PROFILE, cfg.get("workload"), cfg.get("logm"), cfg.get("shards", 1))
    except Exception:
        return None, "no PMC profile committed"


def valu_issue(workload, logm, shards, ms_per_proof, devinfo):
    """The whole step against the VALU-issue bound of the instructions it executes: wave-level VALU instructions per proof from the committed SQ-counter
    pass of THESE kernels (profiles/r04_instruction_counts.json <- r04_sq_counters.txt; quoted only while the kernel sources match), the issue time of
    their mix (v_mad_u64_u32 5.9 cycles, everything else 2.4: measured), and the SIMD-cycles the timed steps really took per proof."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r04_instruction_counts.json")))
        cfg = d.get("config", {})
        if d.get("kernel_sources_sha16") != kernel_sources_sha16() or cfg.get("workload") != workload or cfg.get("logm") != logm or cfg.get("shards", 1) != shards:
            return None
        insts, bound_cpi = d["wave_valu_instructions_per_proof"], d["issue_bound_cycles_per_instruction"]
        simd_cycles = ms_per_proof * 1e-3 * devinfo["clock_mhz"] * 1e6 * devinfo["compute_units"] * 4
        return {"wave_valu_instructions_per_proof": insts, "cycles_per_instruction": round(simd_cycles / insts, 3), "issue_bound_cycles_per_instruction": bound_cpi,
                "frac": round(bound_cpi * insts / simd_cycles, 4), "source": "profiles/r04_instruction_counts.json (commit %s)" % d.get("commit", "?"),
                "note": "the step is its instruction count: SIMD-cycles per wave-level VALU instruction of the timed steps against the issue time of the instruction mix "
                        "(no schedule, priority or occupancy change moves the step any more: profiles/r04_schedule_sweep.txt)"}
    except Exception:
        return None


def proof_bytes(r1cs, pk, m):
    """each operand read once, each result written once."""
    V, nIn = r1cs.V, r1cs.nIn
    return (64 * (pk.nA + (m - 1) + (V - nIn)) + 128 * pk.nB + 32 * (pk.nA + pk.nB + (m - 1) + (V - nIn))
            + 36 * r1cs.nnz + 32 * (V + 1) + 3 * 32 * m + 7 * 2 * 32 * m + 4 * 32 * m)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(args, P, R, F, pk, r1cs, wm, gpu_json, logm, device):
    """The CPU oracle (restatement of r1cs_gg_ppzksnark_zok.tcc:451-550; libsnark unavailable offline) on this host's
    cores: one proof of the bench's own circuit and key on all cores (also the parity check of the timed GPU proof), and a
    one-thread figure beside it on a smaller sample (a full 2^20 proof takes about a minute on one core)."""
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    import oracle_lib as O
    threads = min(O.cpu_share(), 64)
    O.lib().orc_set_threads(threads)
    opk = O.pk_from_parts(pk.parts())
    t0 = time.perf_counter()
    js, phases = O.prove(opk, r1cs, wm)
    dt = time.perf_counter() - t0
    base = {"value": round(1.0 / dt, 5), "unit": "proofs/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "constraints_per_sec": round(r1cs.nC / dt, 1),
            "sample": "1 proof of the same 2^%d circuit and key (%.1f s); CPU restatement of r1cs_gg_ppzksnark_zok.tcc:451-550, OpenMP, %d threads" % (logm, dt, threads),
            "phases_s": [round(p, 3) for p in phases]}
    try:                                                    # one thread, bounded sample
        l1 = min(args.cpu_1t_logm, logm)
        r1, w1 = R.synthetic_chain((1 << l1) - 2, 1)
        w1m = F.fr_to_mont(w1)
        pk1, _ = P.keygen(r1, seed=R.SEED_DEFAULT, device=device)
        opk1 = O.pk_from_parts(pk1.parts())
        O.lib().orc_set_threads(1)
        t0 = time.perf_counter()
        O.prove(opk1, r1, w1m)
        d1 = time.perf_counter() - t0
        O.lib().orc_set_threads(threads)
        t0 = time.perf_counter()
        O.prove(opk1, r1, w1m)
        dn = time.perf_counter() - t0
        base["one_thread"] = {"value": round(1.0 / d1, 5), "unit": "proofs/s at 2^%d" % l1, "cores": 1, "constraints_per_sec": round(r1.nC / d1, 1),
                              "sample": "1 proof of the 2^%d chain circuit on ONE thread (%.1f s); the same proof on %d threads: %.2f s" % (l1, d1, threads, dn)}
    except Exception as e:
        base["one_thread"] = {"error": repr(e)[:200]}
    finally:
        O.lib().orc_set_threads(threads)
    return base, (js == gpu_json)


if __name__ == "__main__":
    main()
