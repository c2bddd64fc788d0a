"""Known answers of the reference for the MiMC / Merkle front end (ethsnarks_amd/gadgets.py) and the two
circuit-shaped measurement configs.  Sources of the constants (read as text from the reference):
  H(123), mimc(1,1), mimc_hash([1,1]), cipher + hash KATs   ethsnarks/mimc/permutation.py:32,175-185
                                                             (= src/test/test_mimc.cpp:52-54, src/test/test_mimc_hash.cpp:16-27)
  Merkle IVs                                                 src/gadgets/merkle_tree.cpp:78-108
  depth-1 root, depth-29 root / levels / placeholders        test/test_merkle.py:34-40,82-114 (= src/test/test_merkle_tree.cpp:62-64)
"""
import json
import numpy as np
import pytest
from ethsnarks_amd import gadgets as G, fields as F

A = 3703141493535563179657531719960160174296085208671919316200479060314459804651
B = 134551314051432487569247388144051420116740427803855572138106146683954151557


def test_keccak_and_mimc_kats():
    assert G.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert G.H(123) == 38632140595220392354280998614525578145353818029287874088356304829962854601866
    assert G.mimc(1, 1) == 2447343676970420247355835473667983267115132689045447905848734383579598297563
    assert G.mimc_hash([1, 1]) == 4087330248547221366577133490880315793780387749595119806283278576811074525767
    assert G.mimc(A, B) == 11437467823393790387399137249441941313717686441929791910070352316474327319704
    assert G.mimc_hash([A, B], 918403109389145570117360101535982733651217667914747213867238065296420114726) == \
        15683951496311901749339509118960676303290224812129752890706581988986633412003


def test_merkle_kats():
    ivs = G.merkle_ivs(29)
    assert ivs[0] == 149674538925118052205057075966660054952481571156186698930522557832224430770
    assert ivs[1] == 9670701465464311903249220692483401938888498641874948577387207195814981706974
    assert ivs[28] == 6037428193077828806710267464232314380014232668931818917272972397574634037180
    assert G.merkle_root(B, [1], [A], ivs) == 3075442268020138823380831368198734873612490112867968717790651410945045657947
    assert G.merkle_unique(20, 20) == 6738165491478210350639451800403024427867073896603076888955948358229240057870
    assert G.merkle_unique(2, 2) == 21534879888322772601810176771999178940739467644392123609236489175629034941722
    assert G.merkle_unique(0, 0) == 2544023609834722662089612003212769975105508295482723304413974529614913939747
    assert G.merkle_unique(1, 1) == 17296471688945713021042054900108821045192859417413320566181654591511652308323
    assert G.merkle_unique(13, 1) == 14116139569958633576637617144876714429777518811711593939929091541932333542283


def test_mimc_hash_gadget_matches_reference_test():
    """src/test/test_mimc_hash.cpp: two public messages, private IV, expected digest, 2*(91*4+1) constraints"""
    pb = G.Protoboard()
    m0, m1 = pb.allocate(A), pb.allocate(B)
    pb.set_input_sizes(2)
    iv = pb.allocate(918403109389145570117360101535982733651217667914747213867238065296420114726)
    g = G.MiMCe7HashGadget(pb, iv, [m0, m1])
    g.generate_r1cs_witness(); g.generate_r1cs_constraints()
    assert pb.val(g.result()) == 15683951496311901749339509118960676303290224812129752890706581988986633412003
    assert pb.num_constraints() == 2 * (91 * 4 + 1) and pb.is_satisfied()


def test_merkle_membership_depth29_circuit():
    r, w, root = G.merkle_membership_circuit(29)
    assert root == 14972246236048249827985830600768475898195156734731557762844426864943654467818      # test/test_merkle.py:92
    assert r.nC == 21345 and r.nIn == 1 and r.domain_size == 1 << 15 and w[1] == root
    # many 0/1 witness values (selector bits and their products)
    assert sum(1 for v in w if v in (0, 1)) >= 88     # ONE, 29 address bits, 2 x 29 zero products


def test_config1_mimc_preimage_on_cpu_path(oracle):
    """BASELINE config 1: 4 015-constraint MiMC preimage circuit through the CPU oracle path (plumbing):
    keygen -> prove -> pairing verification, and .raw / JSON round trip"""
    import pyref
    r, w, digest = G.mimc_preimage_circuit(11)
    assert r.nC == 11 * (91 * 4 + 1) + 1 and r.domain_size == 1 << 12
    wm = F.fr_to_mont(w)
    pk, vk = oracle.keygen(r, seed=11)
    js, _ = oracle.prove(pk, r, wm)
    proof = json.loads(js)
    assert int(proof["input"][0], 16) == digest
    assert pyref.verify(json.loads(vk.to_json()), proof)


def test_merkle_circuit_emulated_kernels(emul, oracle):
    """depth-1 membership circuit through the HIP sources (CPU emulation): 0/1-heavy witness, sparse queries"""
    from ethsnarks_amd import prover
    prover._lib = None; prover._lib_path_loaded = None
    prover.load_library(emul)
    try:
        r, w, root = G.merkle_membership_circuit(1, leaf=5, address=1, path=[7])
        wm = F.fr_to_mont(w)
        pk_o, _ = oracle.keygen(r, seed=4)
        expect, _ = oracle.prove(pk_o, r, wm)
        ctx = prover.ProverContext(prover.ProvingKey.from_parts(**pk_o.parts()), r)
        assert prover.prove(ctx, wm) == expect
    finally:
        prover._lib = None; prover._lib_path_loaded = None
