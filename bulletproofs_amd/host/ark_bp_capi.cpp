// C entry points over ark_bp.hpp for the Python tests and bench.py (ctypes).  A Rust build of
// the reference would not use this file: it calls include/bpgpu.h directly.
#include <chrono>
#include <cstring>
#include <string>

#include "ark_bp.hpp"

using namespace ark_bp;

namespace {
thread_local std::string g_msg;

IpaInstance make_instance(const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen, const uint64_t* result,
                          size_t n) {
  IpaInstance x;
  x.gens.vec_size = n;
  x.gens.a_gens.resize(n);
  x.gens.b_gens.resize(n);
  if (n) {
    std::memcpy(x.gens.a_gens.data(), a_gens, n * 96);
    std::memcpy(x.gens.b_gens.data(), b_gens, n * 96);
  }
  std::memcpy(x.gens.ip_gen.v, ip_gen, 96);
  if (result) std::memcpy(x.result.v, result, 96);
  return x;
}
IpaWitness make_witness(const uint64_t* a, const uint64_t* b, size_t n) {
  IpaWitness w;
  w.a.resize(n);
  w.b.resize(n);
  if (n) {
    std::memcpy(w.a.data(), a, n * 32);
    std::memcpy(w.b.data(), b, n * 32);
  }
  return w;
}

template <class F>
int guarded(F f) {
  try {
    return f();
  } catch (const Panic& p) {
    g_msg = p.what();
    return -100;
  } catch (const std::exception& e) {
    g_msg = e.what();
    return -101;
  }
}

template <int CURVE>
int bp_prove(const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen, const uint64_t* a, const uint64_t* b,
             size_t n, const uint8_t* fs_seed, uint64_t* ls, uint64_t* rs, uint64_t* challenges, uint64_t* final_a,
             uint64_t* final_b, double* prove_ms) {
  IpaInstance x = make_instance(a_gens, b_gens, ip_gen, nullptr, n);
  IpaWitness w = make_witness(a, b, n);
  FiatShamirRng fs = FiatShamirRng::from_seed(fs_seed, 32);
  std::vector<Fr> ch;
  auto t0 = std::chrono::steady_clock::now();
  BpProof pf = Bp<CURVE>::prove(x, w, fs, &ch);  // the window test_ipa times (src/lib.rs:119-121)
  auto t1 = std::chrono::steady_clock::now();
  if (prove_ms) *prove_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  for (size_t i = 0; i < pf.ls.size(); i++) {
    std::memcpy(ls + 12 * i, pf.ls[i].v, 96);
    std::memcpy(rs + 12 * i, pf.rs[i].v, 96);
    if (challenges) std::memcpy(challenges + 4 * i, ch[i].v, 32);
  }
  std::memcpy(final_a, pf.final_a.v, 32);
  std::memcpy(final_b, pf.final_b.v, 32);
  return (int)pf.ls.size();
}

// Bp::prove with the generators already in HBM (handle from arkbp_gens_upload)
template <int CURVE>
int bp_prove_cached(const DeviceGens* dg, const uint64_t* a, const uint64_t* b, size_t n, const uint8_t* fs_seed,
                    uint64_t* ls, uint64_t* rs, uint64_t* challenges, uint64_t* final_a, uint64_t* final_b,
                    double* prove_ms) {
  IpaInstance x;  // the host copy of the generators is not needed: sizes only
  x.gens.vec_size = n;
  x.gens.a_gens.resize(n);
  x.gens.b_gens.resize(n);
  IpaWitness w = make_witness(a, b, n);
  FiatShamirRng fs = FiatShamirRng::from_seed(fs_seed, 32);
  std::vector<Fr> ch;
  auto t0 = std::chrono::steady_clock::now();
  BpProof pf = Bp<CURVE>::prove(x, w, fs, &ch, dg);
  auto t1 = std::chrono::steady_clock::now();
  if (prove_ms) *prove_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  for (size_t i = 0; i < pf.ls.size(); i++) {
    std::memcpy(ls + 12 * i, pf.ls[i].v, 96);
    std::memcpy(rs + 12 * i, pf.rs[i].v, 96);
    if (challenges) std::memcpy(challenges + 4 * i, ch[i].v, 32);
  }
  std::memcpy(final_a, pf.final_a.v, 32);
  std::memcpy(final_b, pf.final_b.v, 32);
  return (int)pf.ls.size();
}

template <int CURVE>
int bp_verify(const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen, const uint64_t* result, size_t n,
              const uint64_t* ls, const uint64_t* rs, size_t k, const uint64_t* final_a, const uint64_t* final_b,
              const uint8_t* fs_seed, const DeviceGens* cached, double* verify_ms) {
  IpaInstance x;
  if (cached) {  // the host copy of the generators is not needed: sizes, ip_gen and result only
    x.gens.vec_size = n;
    x.gens.a_gens.resize(n);
    x.gens.b_gens.resize(n);
    std::memcpy(x.result.v, result, 96);
  } else {
    x = make_instance(a_gens, b_gens, ip_gen, result, n);
  }
  BpProof pf;
  pf.ls.resize(k);
  pf.rs.resize(k);
  if (k) {
    std::memcpy(pf.ls.data(), ls, k * 96);
    std::memcpy(pf.rs.data(), rs, k * 96);
  }
  std::memcpy(pf.final_a.v, final_a, 32);
  std::memcpy(pf.final_b.v, final_b, 32);
  FiatShamirRng fs = FiatShamirRng::from_seed(fs_seed, 32);
  auto t0 = std::chrono::steady_clock::now();
  Bp<CURVE>::verify(x, pf, fs, cached);  // the window test_ipa times (src/lib.rs:127-131)
  auto t1 = std::chrono::steady_clock::now();
  if (verify_ms) *verify_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  return 0;
}

template <int CURVE>
int step_prove(const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen, const uint64_t* result,
               const uint64_t* a, const uint64_t* b, size_t n, const uint8_t* fs_seed, uint64_t* l, uint64_t* r,
               uint64_t* a_gens_next, uint64_t* b_gens_next, uint64_t* a_next, uint64_t* b_next, uint64_t* result_next) {
  IpaInstance x = make_instance(a_gens, b_gens, ip_gen, result, n);
  IpaWitness w = make_witness(a, b, n);
  FiatShamirRng fs = FiatShamirRng::from_seed(fs_seed, 32);
  auto o = Bp2aryStep<CURVE>::prove(x, w, fs);
  size_t m = o.witness.a.size();
  std::memcpy(l, o.proof.l.v, 96);
  std::memcpy(r, o.proof.r.v, 96);
  std::memcpy(a_gens_next, o.instance.gens.a_gens.data(), m * 96);
  std::memcpy(b_gens_next, o.instance.gens.b_gens.data(), m * 96);
  std::memcpy(a_next, o.witness.a.data(), m * 32);
  std::memcpy(b_next, o.witness.b.data(), m * 32);
  std::memcpy(result_next, o.instance.result.v, 96);
  return (int)m;
}

template <int CURVE>
int step_verify(const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen, const uint64_t* result, size_t n,
                const uint64_t* l, const uint64_t* r, const uint8_t* fs_seed, uint64_t* a_gens_next, uint64_t* b_gens_next,
                uint64_t* result_next) {
  IpaInstance x = make_instance(a_gens, b_gens, ip_gen, result, n);
  typename Bp2aryStep<CURVE>::Proof pf;
  std::memcpy(pf.l.v, l, 96);
  std::memcpy(pf.r.v, r, 96);
  FiatShamirRng fs = FiatShamirRng::from_seed(fs_seed, 32);
  IpaInstance nx = Bp2aryStep<CURVE>::verify(x, pf, fs);
  size_t m = nx.gens.a_gens.size();
  std::memcpy(a_gens_next, nx.gens.a_gens.data(), m * 96);
  std::memcpy(b_gens_next, nx.gens.b_gens.data(), m * 96);
  std::memcpy(result_next, nx.result.v, 96);
  return (int)m;
}

// KaryBp::prove_step; cross terms: (k - 1) points each; next vectors: ck = ceil(n / k) elements each
template <int CURVE>
int kary_prove_step(size_t k, const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen, const uint64_t* result,
                    const uint64_t* a, const uint64_t* b, size_t n, const uint8_t* fs_seed, uint64_t* pos_cross,
                    uint64_t* neg_cross, uint64_t* a_gens_next, uint64_t* b_gens_next, uint64_t* a_next, uint64_t* b_next,
                    uint64_t* result_next) {
  IpaInstance x = make_instance(a_gens, b_gens, ip_gen, result, n);
  IpaWitness w = make_witness(a, b, n);
  FiatShamirRng fs = FiatShamirRng::from_seed(fs_seed, 32);
  auto o = KaryBp<CURVE>::prove_step(k, x, w, fs);
  const size_t m = o.witness.a.size();
  if (k > 1) {
    std::memcpy(pos_cross, o.pos_cross.data(), (k - 1) * 96);
    std::memcpy(neg_cross, o.neg_cross.data(), (k - 1) * 96);
  }
  std::memcpy(a_gens_next, o.instance.gens.a_gens.data(), m * 96);
  std::memcpy(b_gens_next, o.instance.gens.b_gens.data(), m * 96);
  std::memcpy(a_next, o.witness.a.data(), m * 32);
  std::memcpy(b_next, o.witness.b.data(), m * 32);
  std::memcpy(result_next, o.instance.result.v, 96);
  return (int)m;
}
template <int CURVE>
int kary_verify_step(size_t k, const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen, const uint64_t* result,
                     size_t n, const uint64_t* pos_cross, const uint64_t* neg_cross, const uint8_t* fs_seed,
                     uint64_t* a_gens_next, uint64_t* b_gens_next, uint64_t* result_next) {
  IpaInstance x = make_instance(a_gens, b_gens, ip_gen, result, n);
  std::vector<G> pos(k - 1), neg(k - 1);
  if (k > 1) {
    std::memcpy(pos.data(), pos_cross, (k - 1) * 96);
    std::memcpy(neg.data(), neg_cross, (k - 1) * 96);
  }
  FiatShamirRng fs = FiatShamirRng::from_seed(fs_seed, 32);
  IpaInstance nx = KaryBp<CURVE>::verify_step(k, x, pos, neg, fs);
  const size_t m = nx.gens.a_gens.size();
  std::memcpy(a_gens_next, nx.gens.a_gens.data(), m * 96);
  std::memcpy(b_gens_next, nx.gens.b_gens.data(), m * 96);
  std::memcpy(result_next, nx.result.v, 96);
  return (int)m;
}
}  // namespace

#define BY_CURVE(fn, ...) (curve == 0 ? fn<0>(__VA_ARGS__) : fn<1>(__VA_ARGS__))

extern "C" {
const char* arkbp_last_error(void) { return g_msg.c_str(); }

// util::msm with the reference's zip semantics (the shorter sequence decides); out: 12 x u64
int arkbp_msm(int curve, const uint64_t* bases, size_t nb, const uint64_t* scalars, size_t ns, uint64_t* out) {
  return guarded([&] {
    G g = curve == 0 ? util::msm<0>((const G*)bases, nb, (const Fr*)scalars, ns)
                     : util::msm<1>((const G*)bases, nb, (const Fr*)scalars, ns);
    std::memcpy(out, g.v, 96);
    return 0;
  });
}
// util::msm::bos_coster_msm (host-only heap algorithm, src/util/msm.rs:34-81); out: 12 x u64
int arkbp_bos_coster_msm(int curve, const uint64_t* bases, size_t nb, const uint64_t* scalars, size_t ns, uint64_t* out) {
  return guarded([&] {
    std::vector<G> b((const G*)bases, (const G*)bases + nb);
    std::vector<Fr> s((const Fr*)scalars, (const Fr*)scalars + ns);
    G g = curve == 0 ? util::bos_coster_msm<0>(b, s) : util::bos_coster_msm<1>(b, s);
    std::memcpy(out, g.v, 96);
    return 0;
  });
}
int arkbp_ip(int curve, const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out) {
  return guarded([&] {
    Fr r = curve == 0 ? util::ip<0>((const Fr*)a, na, (const Fr*)b, nb) : util::ip<1>((const Fr*)a, na, (const Fr*)b, nb);
    std::memcpy(out, r.v, 32);
    return 0;
  });
}
int arkbp_commit(int curve, const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen, const uint64_t* a,
                 const uint64_t* b, size_t n, uint64_t* result) {
  return guarded([&] {
    IpaInstance x = make_instance(a_gens, b_gens, ip_gen, nullptr, n);
    IpaWitness w = make_witness(a, b, n);
    G g = curve == 0 ? IpaRelation<0>::commit_for_witness(x.gens, w) : IpaRelation<1>::commit_for_witness(x.gens, w);
    std::memcpy(result, g.v, 96);
    return 0;
  });
}
// returns the number of rounds (log2 of the padded length) or a negative code
int arkbp_bp_prove(int curve, const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen, const uint64_t* a,
                   const uint64_t* b, size_t n, const uint8_t* fs_seed, uint64_t* ls, uint64_t* rs, uint64_t* challenges,
                   uint64_t* final_a, uint64_t* final_b, double* prove_ms) {
  return guarded([&] {
    return BY_CURVE(bp_prove, a_gens, b_gens, ip_gen, a, b, n, fs_seed, ls, rs, challenges, final_a, final_b, prove_ms);
  });
}
// generators to HBM once, for any number of arkbp_bp_prove_cached calls
int arkbp_gens_upload(int curve, const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen, size_t n,
                      void** handle) {
  return guarded([&] {
    if (curve != 0 && curve != 1) throw Panic("unknown curve");
    IpaInstance x = make_instance(a_gens, b_gens, ip_gen, nullptr, n);
    *handle = new DeviceGens(curve, x.gens);
    return 0;
  });
}
void arkbp_gens_free(void* handle) { delete static_cast<DeviceGens*>(handle); }
int arkbp_bp_prove_cached(int curve, const void* handle, const uint64_t* a, const uint64_t* b, size_t n,
                          const uint8_t* fs_seed, uint64_t* ls, uint64_t* rs, uint64_t* challenges, uint64_t* final_a,
                          uint64_t* final_b, double* prove_ms) {
  return guarded([&] {
    const DeviceGens* dg = static_cast<const DeviceGens*>(handle);
    return BY_CURVE(bp_prove_cached, dg, a, b, n, fs_seed, ls, rs, challenges, final_a, final_b, prove_ms);
  });
}
// 0 = accepted; -100 = the reference's verify would have panicked
int arkbp_bp_verify(int curve, const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen,
                    const uint64_t* result, size_t n, const uint64_t* ls, const uint64_t* rs, size_t k,
                    const uint64_t* final_a, const uint64_t* final_b, const uint8_t* fs_seed, double* verify_ms) {
  return guarded([&] {
    return BY_CURVE(bp_verify, a_gens, b_gens, ip_gen, result, n, ls, rs, k, final_a, final_b, fs_seed, nullptr, verify_ms);
  });
}
// Bp::verify with the generators already in HBM (handle from arkbp_gens_upload)
int arkbp_bp_verify_cached(int curve, const void* handle, const uint64_t* result, size_t n, const uint64_t* ls,
                           const uint64_t* rs, size_t k, const uint64_t* final_a, const uint64_t* final_b,
                           const uint8_t* fs_seed, double* verify_ms) {
  return guarded([&] {
    const DeviceGens* dg = static_cast<const DeviceGens*>(handle);
    return BY_CURVE(bp_verify, nullptr, nullptr, nullptr, result, n, ls, rs, k, final_a, final_b, fs_seed, dg, verify_ms);
  });
}
int arkbp_2ary_prove(int curve, const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen,
                     const uint64_t* result, const uint64_t* a, const uint64_t* b, size_t n, const uint8_t* fs_seed,
                     uint64_t* l, uint64_t* r, uint64_t* a_gens_next, uint64_t* b_gens_next, uint64_t* a_next,
                     uint64_t* b_next, uint64_t* result_next) {
  return guarded([&] {
    return BY_CURVE(step_prove, a_gens, b_gens, ip_gen, result, a, b, n, fs_seed, l, r, a_gens_next, b_gens_next, a_next,
                    b_next, result_next);
  });
}
int arkbp_2ary_verify(int curve, const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen,
                      const uint64_t* result, size_t n, const uint64_t* l, const uint64_t* r, const uint8_t* fs_seed,
                      uint64_t* a_gens_next, uint64_t* b_gens_next, uint64_t* result_next) {
  return guarded([&] {
    return BY_CURVE(step_verify, a_gens, b_gens, ip_gen, result, n, l, r, fs_seed, a_gens_next, b_gens_next, result_next);
  });
}
// Fiat-Shamir primitive checks for the CPU tests (no GPU needed)
int arkbp_fs_challenge(int curve, const uint8_t* fs_seed, const uint64_t* points_xyz, size_t npoints, uint64_t* out) {
  return guarded([&] {
    FiatShamirRng fs = FiatShamirRng::from_seed(fs_seed, 32);
    for (size_t i = 0; i < npoints; i++) {
      G g;
      std::memcpy(g.v, points_xyz + 12 * i, 96);
      if (curve == 0) fs.absorb<0>(g); else fs.absorb<1>(g);
    }
    Fr x = curve == 0 ? fs.rand_scalar<0>() : fs.rand_scalar<1>();
    std::memcpy(out, x.v, 32);
    return 0;
  });
}
int arkbp_scalar_inverse(int curve, const uint64_t* x, uint64_t* out) {
  return guarded([&] {
    Fr a;
    std::memcpy(a.v, x, 32);
    Fr r = curve == 0 ? ScalarOps<0>::inverse(a) : ScalarOps<1>::inverse(a);
    std::memcpy(out, r.v, 32);
    return 0;
  });
}
// IpaRelation::check (ipa.rs:134-146): 0 = holds, -100 = the reference would have panicked
int arkbp_ipa_check(int curve, const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen, const uint64_t* result,
                    const uint64_t* a, const uint64_t* b, size_t n) {
  return guarded([&] {
    IpaInstance x = make_instance(a_gens, b_gens, ip_gen, result, n);
    IpaWitness w = make_witness(a, b, n);
    if (curve == 0) IpaRelation<0>::check(x, w); else IpaRelation<1>::check(x, w);
    return 0;
  });
}
// one round of the k-ary protocol (bp_rec_kary.rs:27-190); both return the next vector length
int arkbp_kary_prove_step(int curve, size_t k, const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen,
                          const uint64_t* result, const uint64_t* a, const uint64_t* b, size_t n, const uint8_t* fs_seed,
                          uint64_t* pos_cross, uint64_t* neg_cross, uint64_t* a_gens_next, uint64_t* b_gens_next,
                          uint64_t* a_next, uint64_t* b_next, uint64_t* result_next) {
  return guarded([&] {
    return BY_CURVE(kary_prove_step, k, a_gens, b_gens, ip_gen, result, a, b, n, fs_seed, pos_cross, neg_cross, a_gens_next,
                    b_gens_next, a_next, b_next, result_next);
  });
}
int arkbp_kary_verify_step(int curve, size_t k, const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen,
                           const uint64_t* result, size_t n, const uint64_t* pos_cross, const uint64_t* neg_cross,
                           const uint8_t* fs_seed, uint64_t* a_gens_next, uint64_t* b_gens_next, uint64_t* result_next) {
  return guarded([&] {
    return BY_CURVE(kary_verify_step, k, a_gens, b_gens, ip_gen, result, n, pos_cross, neg_cross, fs_seed, a_gens_next,
                    b_gens_next, result_next);
  });
}
// padding helpers on 32-byte elements (util/mod.rs:103-134): mode 0 = zero_pad_to_multiple,
// 1 = zero_intersperse_to_multiple; returns the output length (out holds at least n + k elements)
long arkbp_pad(int mode, const uint64_t* in, size_t n, size_t k, uint64_t* out) {
  return guarded([&] {
    std::vector<Fr> v(n);
    if (n) std::memcpy(v.data(), in, n * 32);
    Fr zero{};
    std::vector<Fr> o = mode == 0 ? util::zero_pad_to_multiple(v, k, zero) : util::zero_intersperse_to_multiple(v, k, zero);
    std::memcpy(out, o.data(), o.size() * 32);
    return (int)o.size();
  });
}
// the reference's input recipes over test_rng(), points sampled in one batch on the device:
// mode 0 = examples/msm_bench.rs:40-44 (n points, n scalars; a, b_gens, ip_gen, fs_seed unused),
// mode 1 = src/lib.rs:111-118 (a_gens, b_gens, ip_gen, a, b, then the 32-byte Fiat-Shamir seed)
int arkbp_sample(int curve, int mode, size_t n, uint64_t* a_gens, uint64_t* b_gens, uint64_t* ip_gen, uint64_t* a, uint64_t* b,
                 uint8_t* fs_seed) {
  return guarded([&] {
    StdRng rng = test_rng();
    auto scalars = [&](uint64_t* out) {
      for (size_t i = 0; i < n; i++) {
        Fr s = curve == 0 ? rand_scalar<0>(rng) : rand_scalar<1>(rng);
        std::memcpy(out + 4 * i, s.v, 32);
      }
    };
    if (mode == 0) {
      std::vector<G> gs = curve == 0 ? rand_points<0>(rng, n) : rand_points<1>(rng, n);
      if (n) std::memcpy(a_gens, gs.data(), n * 96);
      scalars(a);
    } else {
      std::vector<G> gs = curve == 0 ? rand_points<0>(rng, 2 * n + 1) : rand_points<1>(rng, 2 * n + 1);
      if (n) std::memcpy(a_gens, gs.data(), n * 96);
      if (n) std::memcpy(b_gens, gs.data() + n, n * 96);
      std::memcpy(ip_gen, gs[2 * n].v, 96);
      scalars(a);
      scalars(b);
      std::array<uint8_t, 32> seed = gen_seed32(rng);
      std::memcpy(fs_seed, seed.data(), 32);
    }
    return 0;
  });
}
// util::mat_vec_mult / vec_mat_mult over a Matrix rebuilt from compressed rows (the tests' transport format)
int arkbp_mat_mult(int curve, int transposed, const uint64_t* row_ptr, const uint64_t* col_idx, const uint64_t* vals,
                   size_t n_rows, const uint64_t* vec, size_t vec_len, size_t output_size, uint64_t* out) {
  return guarded([&] {
    util::Matrix mat(n_rows);
    for (size_t r = 0; r < n_rows; r++)
      for (uint64_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
        Fr v;
        std::memcpy(v.v, vals + 4 * e, 32);
        mat[r].emplace_back(v, (size_t)col_idx[e]);
      }
    std::vector<Fr> x(vec_len);
    if (vec_len) std::memcpy(x.data(), vec, vec_len * 32);
    std::vector<Fr> o = transposed ? (curve == 0 ? util::vec_mat_mult<0>(x, mat, output_size) : util::vec_mat_mult<1>(x, mat, output_size))
                                   : (curve == 0 ? util::mat_vec_mult<0>(mat, x) : util::mat_vec_mult<1>(mat, x));
    if (!o.empty()) std::memcpy(out, o.data(), o.size() * 32);
    return (int)o.size();
  });
}
}
