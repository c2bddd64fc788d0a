// dev tool: device vs host results of the loose-domain field / curve operations (bisecting an arithmetic change).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ethsnarks_amd/csrc tools/dev_curve_check.cpp -o tools/dev_curve_check
#include "bn254.hpp"
#include <stdio.h>
#include <vector>
namespace zk { thread_local char g_last_error[256] = ""; }
using namespace zk;

struct Out { fe lsqr, lmul, lmul2, lmul4, ldbl3; G2::XYZZ madd2, add2; G1::XYZZ dbl, add_same, madd, add_pq, mul13; };

__global__ void k_ops(const fe *in, Out *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
    Out o;
    o.lsqr = Fq::canon(Fq::lsqr(a));
    o.lmul = Fq::canon(Fq::lmul(a, b));
    o.lmul2 = Fq::canon(Fq::lmul2(a, b, c, d));
    o.lmul4 = Fq::canon(Fq::lmul4(a, b, c, d, b, c, d, a));
    {   G2::XYZZ p2; p2.X.c0 = a; p2.X.c1 = b; p2.Y.c0 = c; p2.Y.c1 = d; p2.ZZ.c0 = b; p2.ZZ.c1 = d; p2.ZZZ.c0 = c; p2.ZZZ.c1 = a;
        G2::Affine q2; q2.x.c0 = d; q2.x.c1 = c; q2.y.c0 = b; q2.y.c1 = a;
        o.madd2 = G2::canon(G2::madd(p2, q2));
        G2::XYZZ r2; r2.X = p2.Y; r2.Y = p2.ZZ; r2.ZZ = p2.ZZZ; r2.ZZZ = p2.X;
        o.add2 = G2::canon(G2::add(p2, r2)); }
    fe xx = Fq::lsqr(a); o.ldbl3 = Fq::canon(Fq::ladd(Fq::ldbl(xx), xx));
    G1::XYZZ p; p.X = a; p.Y = b; p.ZZ = c; p.ZZZ = d;       // not a curve point: the formulas are polynomial identities anyway
    o.dbl = G1::canon(G1::dbl(p));
    o.add_same = G1::canon(G1::add(p, p));
    G1::Affine q; q.x = c; q.y = d;
    o.madd = G1::canon(G1::madd(p, q));
    G1::XYZZ p2; p2.X = b; p2.Y = c; p2.ZZ = d; p2.ZZZ = a;
    o.add_pq = G1::canon(G1::add(p, p2));
    o.mul13 = G1::canon(G1::mul_small(p, 13));
    out[i] = o;
}

static bool eq(const fe &a, const fe &b) { for (int i = 0; i < 8; i++) if (a.l[i] != b.l[i]) return false; return true; }
static bool eq2(const fe2 &a, const fe2 &b) { return eq(a.c0, b.c0) && eq(a.c1, b.c1); }
static bool eqp2(const G2::XYZZ &a, const G2::XYZZ &b) { return eq2(a.X, b.X) && eq2(a.Y, b.Y) && eq2(a.ZZ, b.ZZ) && eq2(a.ZZZ, b.ZZZ); }
static bool eqp(const G1::XYZZ &a, const G1::XYZZ &b) { return eq(a.X, b.X) && eq(a.Y, b.Y) && eq(a.ZZ, b.ZZ) && eq(a.ZZZ, b.ZZZ); }

int main() {
    const int n = 4096;
    std::vector<fe> in(4 * n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto &v : in) { fe t; for (int k = 0; k < 8; k += 2) { uint64_t r = rnd(); t.l[k] = (uint32_t)r; t.l[k + 1] = (uint32_t)(r >> 32); } t.l[7] &= 0x0fffffffu; v = Fq::to_mont(t); }
    // edge operands: 0, 1, p - 1 in a few slots
    in[0] = Fq::zero(); in[5] = Fq::one(); in[10] = Fq::neg(Fq::one()); in[15] = Fq::neg(Fq::one());
    fe *d_in; Out *d_out;
    hipMalloc(&d_in, sizeof(fe) * in.size()); hipMalloc(&d_out, sizeof(Out) * n);
    hipMemcpy(d_in, in.data(), sizeof(fe) * in.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_ops, dim3(n / 64), dim3(64), 0, 0, d_in, d_out, n);
    std::vector<Out> out(n);
    hipMemcpy(out.data(), d_out, sizeof(Out) * n, hipMemcpyDeviceToHost);
    int bad[12] = {0};
    for (int i = 0; i < n; i++) {
        const fe a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
        G1::XYZZ p; p.X = a; p.Y = b; p.ZZ = c; p.ZZZ = d; G1::Affine q; q.x = c; q.y = d;
        fe xx = Fq::sqr(a);
        bad[0] += !eq(out[i].lsqr, Fq::sqr(a));
        bad[1] += !eq(out[i].lmul, Fq::mul(a, b));
        bad[2] += !eq(out[i].lmul2, Fq::add(Fq::mul(a, b), Fq::mul(c, d)));
        bad[3] += !eq(out[i].ldbl3, Fq::add(Fq::add(xx, xx), xx));
        bad[4] += !eqp(out[i].dbl, G1::dbl(p));
        bad[5] += !eqp(out[i].add_same, G1::add(p, p));
        bad[6] += !eqp(out[i].madd, G1::madd(p, q));
        G1::XYZZ p2; p2.X = b; p2.Y = c; p2.ZZ = d; p2.ZZZ = a;
        bad[7] += !eqp(out[i].add_pq, G1::add(p, p2));
        bad[8] += !eqp(out[i].mul13, G1::mul_small(p, 13));
        bad[9] += !eq(out[i].lmul4, Fq::add(Fq::add(Fq::mul(a, b), Fq::mul(c, d)), Fq::add(Fq::mul(b, c), Fq::mul(d, a))));
        {   G2::XYZZ q2p; q2p.X.c0 = a; q2p.X.c1 = b; q2p.Y.c0 = c; q2p.Y.c1 = d; q2p.ZZ.c0 = b; q2p.ZZ.c1 = d; q2p.ZZZ.c0 = c; q2p.ZZZ.c1 = a;
            G2::Affine q2; q2.x.c0 = d; q2.x.c1 = c; q2.y.c0 = b; q2.y.c1 = a;
            bad[10] += !eqp2(out[i].madd2, G2::canon(G2::madd(q2p, q2)));
            G2::XYZZ r2; r2.X = q2p.Y; r2.Y = q2p.ZZ; r2.ZZ = q2p.ZZZ; r2.ZZZ = q2p.X;
            bad[11] += !eqp2(out[i].add2, G2::canon(G2::add(q2p, r2))); }
    }
    printf("mismatches of %d: lsqr %d lmul %d lmul2 %d 3xx %d dbl %d add(p,p) %d madd %d add(p,q) %d mul_small %d lmul4 %d G2 madd %d G2 add %d\n", n, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6], bad[7], bad[8], bad[9], bad[10], bad[11]);
    int any = 0; for (int b : bad) any |= b; return any ? 1 : 0;
}
