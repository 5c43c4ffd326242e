#include "mimo_hip.h"
#include <vector>
#include <random>
#include <cstdio>
#include <cmath>
int main() {
  std::mt19937_64 g(1);
  std::normal_distribution<double> nd;
  for (int K : {1, 3, 4, 5, 64, 130}) for (int D : {1, 2, 5, 16, 32}) {
    std::vector<double> a(K * D), b(K), c((size_t)K * D * D), d(K), mus(K * D), psis((size_t)K * D * D), nus(K), hld(K),
        cc(K), bb(K * D), W((size_t)K * D * D), E2(K), E4(K);
    for (int k = 0; k < K; ++k) {
      b[k] = 1.0 + k; d[k] = 3.0 + k;
      std::vector<double> A(D * D);
      for (auto& v : A) v = nd(g);
      for (int i = 0; i < D; ++i) a[k * D + i] = nd(g);
      for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) {
        double s = (i == j) ? D + 1.0 : 0.0;
        for (int l = 0; l < D; ++l) s += A[i * D + l] * A[j * D + l];
        c[((size_t)k * D + i) * D + j] = s + a[k * D + i] * a[k * D + j] / b[k];
      }
    }
    int rc = mimo_host_nw_vi(K, D, a.data(), b.data(), c.data(), d.data(), mus.data(), psis.data(), nus.data(), hld.data(),
                             cc.data(), bb.data(), W.data(), E2.data(), E4.data());
    if (rc != 0 || !std::isfinite(cc[K - 1])) { printf("nw K=%d D=%d rc=%d\n", K, D, rc); return 1; }
    {   // the bound's term from the outputs above (prior := the same block; its log-partition value is arbitrary here)
      std::vector<double> out(K);
      rc = mimo_host_nw_vlb(K, D, a.data(), b.data(), c.data(), d.data(), a.data(), b.data(), c.data(), d.data(), E4.data(),
                            nus.data(), hld.data(), bb.data(), E2.data(), W.data(), E4.data(), out.data());
      if (rc != 0 || !std::isfinite(out[K - 1])) { printf("nw vlb K=%d D=%d rc=%d\n", K, D, rc); return 1; }
    }
    for (int tied : {0, 1}) {   // the whole sweep in two calls (prior := the block above, statistics := the same block again)
      const size_t DD = (size_t)D * D;
      std::vector<double> al0(K, 1.5), cnt(K, 3.0), al(K), qa(K * D), qb(K), qc(K * DD), qd(K), m2(K * D), p2(K * DD), n2(K), h2(K),
          natc(K * DD), c2(K), b2(K * D), W2(K * DD), e2(K), e4(K), elp(K), ct(K), plz(K, 0.25), vlb(2);
      rc = mimo_host_gmm_vi_sweep(K, D, tied, al0.data(), cnt.data(), a.data(), b.data(), c.data(), d.data(), a.data(), b.data(),
                                  c.data(), al.data(), qa.data(), qb.data(), qc.data(), qd.data(), m2.data(), p2.data(), n2.data(),
                                  h2.data(), natc.data(), c2.data(), b2.data(), W2.data(), e2.data(), e4.data(), elp.data(), ct.data());
      if (rc == 0)
        rc = mimo_host_gmm_vi_bound(K, D, tied, al0.data(), al.data(), elp.data(), a.data(), b.data(), c.data(), d.data(), plz.data(),
                                    qa.data(), qb.data(), qc.data(), qd.data(), m2.data(), n2.data(), h2.data(), natc.data(),
                                    b2.data(), e2.data(), W2.data(), e4.data(), vlb.data());
      if (rc != 0 || !std::isfinite(vlb[0] + vlb[1] + ct[K - 1])) { printf("sweep K=%d D=%d tied=%d rc=%d\n", K, D, tied, rc); return 1; }
    }
    {   // hierarchical update: five rounds from the statistics above (xk := a, nk := b, sum of second moments := the first block of c, scaled)
      const size_t DD = (size_t)D * D;
      std::vector<double> kp(K, 0.3), m0(D, 0.1), p0(DD, 0.0), sx2(DD), muq(D, 0.0), pm(K * D), pk(K), psq(DD);
      double kq = 0, nq = 0;
      for (int i = 0; i < D; ++i) p0[(size_t)i * D + i] = 1.0;
      for (size_t i = 0; i < DD; ++i) sx2[i] = c[i] * K;
      rc = mimo_host_hier_vi(K, D, 5, kp.data(), m0.data(), 0.5, p0.data(), D + 2.0, a.data(), b.data(), sx2.data(), muq.data(), pm.data(),
                             pk.data(), &kq, psq.data(), &nq);
      if (rc != 0 && rc != -1) { printf("hier K=%d D=%d rc=%d\n", K, D, rc); return 1; }      // (not positive definite is a legal answer here)
    }
    {   // the tied flavour on the same natural parameters (its pooled block is an average of SPD blocks)
      std::vector<double> natc((size_t)K * D * D), tp((size_t)K * D * D), tm(K * D), tn(K), th(K);
      rc = mimo_host_nw_vi_tied(K, D, a.data(), b.data(), c.data(), d.data(), tm.data(), tp.data(), tn.data(), th.data(),
                                natc.data(), cc.data(), bb.data(), W.data(), E2.data(), E4.data());
      if (rc != 0 || !std::isfinite(cc[K - 1])) { printf("tied nw K=%d D=%d rc=%d\n", K, D, rc); return 1; }
    }
    {   // Gibbs draw from the posterior just computed (psis is SPD here)
      const int nt = D * (D - 1) / 2;
      std::vector<double> z((size_t)K * nt + 1), gg(K * D), ee(K * D), omu(K * D), olam((size_t)K * D * D), oc(K), ob(K * D);
      for (auto& v : z) v = nd(g);
      for (auto& v : ee) v = nd(g);
      for (auto& v : gg) v = 0.5 + std::fabs(nd(g));
      rc = mimo_host_nw_gibbs(K, D, mus.data(), b.data(), psis.data(), z.data(), gg.data(), ee.data(), omu.data(), olam.data(),
                              oc.data(), ob.data());
      if (rc != 0 || !std::isfinite(oc[K - 1])) { printf("gibbs K=%d D=%d rc=%d\n", K, D, rc); return 1; }
    }
    // matrix-normal-Wishart: dy x dc blocks
    int dy = D > 8 ? 8 : D, dc = (D > 9 ? 9 : D) + 1;
    for (int affine : {0, 1}) {
      std::vector<double> ma((size_t)K * dy * dc), mb((size_t)K * dc * dc), mc((size_t)K * dy * dy), md(K), Ms((size_t)K * dy * dc),
          mp((size_t)K * dy * dy), mn(K), mh(K), Kinv((size_t)K * dc * dc), e1((size_t)K * dy * dc), e2((size_t)K * dc * dc), e4(K);
      int Dz = dc - affine + dy;
      std::vector<double> mcc(K), mbb((size_t)K * Dz), mW((size_t)K * Dz * Dz);
      for (int k = 0; k < K; ++k) {
        md[k] = 2.0 + k;
        for (int i = 0; i < dc; ++i) for (int j = 0; j < dc; ++j) mb[((size_t)k * dc + i) * dc + j] = (i == j) ? 2.0 + i : 0.1;
        for (int i = 0; i < dy * dc; ++i) ma[(size_t)k * dy * dc + i] = 0.01 * nd(g);
        for (int i = 0; i < dy; ++i) for (int j = 0; j < dy; ++j) mc[((size_t)k * dy + i) * dy + j] = (i == j) ? 3.0 + i : 0.2;
      }
      rc = mimo_host_mnw_vi(K, dy, dc, affine, ma.data(), mb.data(), mc.data(), md.data(), Ms.data(), mp.data(), mn.data(),
                            mh.data(), Kinv.data(), mcc.data(), mbb.data(), mW.data(), e1.data(), e2.data(), e4.data());
      if (rc != 0 && dc >= 2) { printf("mnw K=%d dy=%d dc=%d affine=%d rc=%d\n", K, dy, dc, affine, rc); return 1; }
    }
  }
  {   // numpy's legacy stream: every branch of the gamma sampler, the refill of the Mersenne Twister, the cached gaussian
    std::vector<uint32_t> key(624);
    for (int i = 0; i < 624; ++i) key[i] = 1812433253u * (i + 7) + 12345u * i;
    int pos = 624, has = 0; double gs = 0.0;
    const int K = 50, nb = 13, ng = 6, na = 5;
    std::vector<double> sh((size_t)K * ng), ob((size_t)K * nb), og((size_t)K * ng), oa((size_t)K * na);
    for (int k = 0; k < K; ++k) for (int i = 0; i < ng; ++i) sh[(size_t)k * ng + i] = i == 0 ? 1.0 : i == 1 ? 0.0 : i == 2 ? 0.07 * (k + 1) : 0.4 * k + i;
    int rc = mimo_host_legacy_draws(key.data(), &pos, &has, &gs, K, nb, ng, na, sh.data(), ob.data(), og.data(), oa.data());
    if (rc != 0 || !std::isfinite(og[(size_t)K * ng - 1]) || pos < 0 || pos > 624) { printf("legacy draws rc=%d\n", rc); return 1; }
    {   // ... and the in-place form: the pair rewind with and without a refill since the pair began
      for (int start : {0, 3, 617, 621, 624}) {
        std::vector<uint32_t> k2(key), fin(625);
        int p2 = start, redraw = -1;
        rc = mimo_host_legacy_draws_inplace(k2.data(), &p2, start & 1, 0.25, K, nb, ng, na, sh.data(), ob.data(), og.data(), oa.data(), &redraw, fin.data());
        if (rc != 0 || redraw < 0 || redraw > 1 || p2 < 0 || p2 > 624) { printf("legacy in place start=%d rc=%d\n", start, rc); return 1; }
      }
    }
    {   // random.sample's two branches
      std::vector<int64_t> idx(300);
      std::vector<uint32_t> k3(key);
      int p3 = 624;
      if (mimo_host_py_sample(k3.data(), &p3, 300, 300, 1, idx.data()) != 0 || mimo_host_py_sample(k3.data(), &p3, 100000, 300, 0, idx.data()) != 0 ||
          mimo_host_py_sample(k3.data(), &p3, 10, 11, 0, idx.data()) >= 0) { printf("py sample\n"); return 1; }
    }
    sh[3] = -1.0;
    if (mimo_host_legacy_draws(key.data(), &pos, &has, &gs, K, nb, ng, na, sh.data(), ob.data(), og.data(), oa.data()) >= 0) { printf("negative shape accepted\n"); return 1; }
  }
  // a block that is not positive definite must come back as an error, not as a crash
  double a1[2] = {0, 0}, b1[1] = {1}, c1[4] = {1, 2, 2, 1}, d1[1] = {3}, o[64];
  int rc = mimo_host_nw_vi(1, 2, a1, b1, c1, d1, o, o + 2, o + 6, o + 7, o + 8, o + 9, o + 11, o + 15, o + 16);
  double m1[2] = {0, 0}, k1[1] = {1}, z1[1] = {0.3}, g1[2] = {1, 1}, e1[2] = {0.1, -0.2};
  int rc2 = mimo_host_nw_gibbs(1, 2, m1, k1, c1, z1, g1, e1, o, o + 2, o + 6, o + 7);
  printf("non-SPD rc=%d, %d (expected negative)\nsanitizer run ok\n", rc, rc2);
  return rc < 0 && rc2 < 0 ? 0 : 1;
}
