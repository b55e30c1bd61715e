// Host side above the C ABI, in C++ because the reference is compiled Rust and no Rust
// toolchain exists in the build image: a mirror of the reference's interface for the hot
// path -- same names, argument meaning and error behaviour (it throws where the reference
// panics) -- whose bodies dispatch to libbpgpu exactly where the Rust shim would
// (INTEGRATION.md).  Nothing here computes group arithmetic on the CPU.
//
//   util::msm / util::ip / util::zero_pad_to_two_power   src/util/mod.rs:15,17-23,137-144
//   FiatShamirRng (ark-marlin 0.3, Blake2s + ChaCha20)     src/lib.rs:5
//   IpaGens / IpaInstance / IpaWitness / IpaRelation       src/relations/ipa.rs:10-160
//   Bp::{prove, verify}, BpProof                           src/proofs/bp_iter.rs:24-197
//   Bp2aryStep::{prove, verify}                            src/reductions/bp_2ary_step.rs:19-142
//   util::{powers, scale_vec, sum_vecs, hadamard, hadamard_gp, zero_pad_to_multiple,
//          zero_intersperse_to_multiple}                    src/util/mod.rs:25-134
//   KaryBp::{prove_step, verify_step}                      src/proofs/bp_rec_kary.rs:27-190
//   test_rng, Fp256 / point sampling, sample_from_length   examples/msm_bench.rs:40-44, src/lib.rs:111-118,
//                                                          src/relations/ipa.rs:28-36,90-97,110-115
//   util::naive_msm (msm_bench's default algorithm)        src/util/msm.rs:13-24
//   util::bos_coster_msm (serial heap algorithm, host only) src/util/msm.rs:34-81
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bpgpu.h"
#include "../csrc/host_field.h"

namespace ark_bp {

enum Curve : int { Pallas = BPGPU_PALLAS, Vesta = BPGPU_VESTA };

struct Panic : std::runtime_error {  // the reference's only error convention on this path
  using std::runtime_error::runtime_error;
};
inline void check(int rc) {
  if (rc != 0) throw Panic(std::string("bpgpu: ") + bpgpu_last_error());
}

// ark-ff Fp256 in memory: 4 x u64, Montgomery form
struct Fr {
  uint64_t v[4];
  bool operator==(const Fr& o) const { return std::memcmp(v, o.v, 32) == 0; }
};
// ark-ec GroupProjective in memory: X | Y | Z, each an Fp256; Z = 0 is the identity
struct G {
  uint64_t v[12];
  bool operator==(const G& o) const { return std::memcmp(v, o.v, 96) == 0; }  // both sides normalised
};

// ---- scalar-field helpers for the few host-side field operations of the protocols ----------
template <int CURVE>
struct ScalarOps {
  using HF = bpgpu::host::HF<1 - CURVE>;
  using H4 = bpgpu::host::H4;
  static H4 h(const Fr& a) { H4 r; std::memcpy(r.v, a.v, 32); return r; }
  static Fr f(const H4& a) { Fr r; std::memcpy(r.v, a.v, 32); return r; }
  static Fr one() { return f(HF::one()); }
  static Fr mul(const Fr& a, const Fr& b) { return f(HF::mul(h(a), h(b))); }
  static Fr square(const Fr& a) { return mul(a, a); }
  static Fr neg(const Fr& a) { return f(HF::neg(h(a))); }
  static Fr inverse(const Fr& a) {
    if (HF::is_zero(h(a))) throw Panic("called `Option::unwrap()` on a `None` value (inverse of zero)");
    return f(HF::inv(h(a)));
  }
};

template <int CURVE>
struct PointOps {
  using HF = bpgpu::host::HF<CURVE>;
  using H4 = bpgpu::host::H4;
  // xy_inf from the C ABI -> Projective { x, y, z: 1 } or zero() = (0, 1, 0)
  static G from_xy_inf(const uint64_t out[9]) {
    G g;
    H4 one = HF::one();
    if (out[8]) {
      std::memset(g.v, 0, 96);
      std::memcpy(g.v + 4, one.v, 32);
    } else {
      std::memcpy(g.v, out, 64);
      std::memcpy(g.v + 8, one.v, 32);
    }
    return g;
  }
  // ToBytes for GroupProjective: canonical X | Y | Z (96 bytes)
  static void to_bytes(const G& g, uint8_t out[96]) {
    for (int k = 0; k < 3; k++) {
      H4 t;
      std::memcpy(t.v, g.v + 4 * k, 32);
      H4 c = HF::to_canon(t);
      std::memcpy(out + 32 * k, c.v, 32);
    }
  }
};

// ---- BLAKE2s-256 (RFC 7693), unkeyed ------------------------------------------------------
class Blake2s {
 public:
  static std::array<uint8_t, 32> digest(const uint8_t* data, size_t len) {
    static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                   0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
    uint32_t h[8];
    for (int i = 0; i < 8; i++) h[i] = IV[i];
    h[0] ^= 0x01010000u ^ 32u;
    uint64_t t = 0;
    size_t off = 0;
    uint8_t block[64];
    while (len - off > 64) {
      t += 64;
      compress(h, data + off, t, false, IV);
      off += 64;
    }
    std::memset(block, 0, 64);
    if (len > off) std::memcpy(block, data + off, len - off);
    t += len - off;
    compress(h, block, t, true, IV);
    std::array<uint8_t, 32> out;
    std::memcpy(out.data(), h, 32);
    return out;
  }

 private:
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  static void compress(uint32_t h[8], const uint8_t* block, uint64_t t, bool last, const uint32_t IV[8]) {
    static const uint8_t S[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    uint32_t m[16], v[16];
    std::memcpy(m, block, 64);
    for (int i = 0; i < 8; i++) {
      v[i] = h[i];
      v[8 + i] = IV[i];
    }
    v[12] ^= (uint32_t)t;
    v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
    auto mix = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
      v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 16);
      v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 12);
      v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 8);
      v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 7);
    };
    for (int r = 0; r < 10; r++) {
      const uint8_t* s = S[r];
      mix(0, 4, 8, 12, m[s[0]], m[s[1]]);
      mix(1, 5, 9, 13, m[s[2]], m[s[3]]);
      mix(2, 6, 10, 14, m[s[4]], m[s[5]]);
      mix(3, 7, 11, 15, m[s[6]], m[s[7]]);
      mix(0, 5, 10, 15, m[s[8]], m[s[9]]);
      mix(1, 6, 11, 12, m[s[10]], m[s[11]]);
      mix(2, 7, 8, 13, m[s[12]], m[s[13]]);
      mix(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
  }
};

// ---- rand_chacha ChaCha{12,20}Rng behind rand_core's BlockRng (64-word buffer) ----------------
template <int DOUBLE_ROUNDS>
class ChaChaRng {
 public:
  explicit ChaChaRng(const std::array<uint8_t, 32>& seed) {
    std::memcpy(key_, seed.data(), 32);
    counter_ = 0;
    index_ = 64;
  }
  uint32_t next_u32() {
    if (index_ >= 64) {
      refill();
      index_ = 0;
    }
    return buf_[index_++];
  }
  uint64_t next_u64() {
    uint64_t lo, hi;
    if (index_ < 63) {
      lo = buf_[index_];
      hi = buf_[index_ + 1];
      index_ += 2;
    } else if (index_ >= 64) {
      refill();
      lo = buf_[0];
      hi = buf_[1];
      index_ = 2;
    } else {
      lo = buf_[63];
      refill();
      hi = buf_[0];
      index_ = 1;
    }
    return lo | (hi << 32);
  }

 private:
  static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  void block(uint64_t ctr, uint32_t* out) const {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; i++) s[4 + i] = key_[i];
    s[12] = (uint32_t)ctr;
    s[13] = (uint32_t)(ctr >> 32);
    s[14] = 0;
    s[15] = 0;
    uint32_t x[16];
    std::memcpy(x, s, 64);
    auto qr = [&](int a, int b, int c, int d) {
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
      x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
      x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < DOUBLE_ROUNDS; r++) {
      qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
      qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
  }
  void refill() {
    for (int i = 0; i < 4; i++) block(counter_ + i, buf_ + 16 * i);
    counter_ += 4;
  }
  uint32_t key_[8];
  uint64_t counter_;
  uint32_t buf_[64];
  int index_;
};
using ChaCha20Rng = ChaChaRng<10>;  // rand_chacha::ChaChaRng, the Fiat-Shamir generator
using StdRng = ChaChaRng<6>;        // rand 0.8 StdRng = ChaCha12

// ark_std::test_rng(): StdRng from a fixed seed (examples/msm_bench.rs:40, src/lib.rs:111)
inline StdRng test_rng() {
  return StdRng(std::array<uint8_t, 32>{1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0});
}

// ---- UniformRand for Fp256 and for short-Weierstrass points (ark-ff / ark-ec 0.3) ---------------
// Fp256::rand: four next_u64 limbs, top bit cleared, accepted iff below the modulus; the accepted limbs
// ARE the Montgomery representation.  FID: 0 = Pallas base field (= Vesta scalar field), 1 = the other.
template <int FID, class Rng>
bpgpu::host::H4 rand_field(Rng& rng) {
  using C = bpgpu::FieldConsts<FID>;
  for (;;) {
    bpgpu::host::H4 x;
    for (int i = 0; i < 4; i++) x.v[i] = rng.next_u64();
    x.v[3] &= 0x7fffffffffffffffull;
    for (int i = 3; i >= 0; i--) {
      if (x.v[i] < C::P64[i]) return x;
      if (x.v[i] > C::P64[i]) break;
    }
  }
}
template <int FID>
struct FieldSqrt {
  using HF = bpgpu::host::HF<FID>;
  using H4 = bpgpu::host::H4;
  static H4 pow(const H4& a, const uint64_t e[4]) {
    H4 acc = HF::one();
    for (int i = 3; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        acc = HF::sqr(acc);
        if ((e[i] >> b) & 1) acc = HF::mul(acc, a);
      }
    return acc;
  }
  // Tonelli-Shanks for p - 1 = 2^32 t (both Pasta fields); false for a non-residue
  static bool sqrt(const H4& a, H4& root) {
    using C = bpgpu::FieldConsts<FID>;
    if (HF::is_zero(a)) { root = a; return true; }
    uint64_t t[4], half[4];  // t = (p - 1) / 2^32, half = (t - 1) / 2
    for (int i = 0; i < 4; i++) t[i] = (C::P64[i] >> 32) | (i + 1 < 4 ? C::P64[i + 1] << 32 : 0);
    for (int i = 0; i < 4; i++) half[i] = (t[i] >> 1) | (i + 1 < 4 ? t[i + 1] << 63 : 0);
    static const H4 z0 = [&] {  // a 2^32-th root of unity of full order: g^t for the first non-residue g
      uint64_t e[4];            // (p - 1) / 2
      for (int i = 0; i < 4; i++) e[i] = (C::P64[i] >> 1) | (i + 1 < 4 ? C::P64[i + 1] << 63 : 0);
      for (uint64_t g = 2;; g++) {
        H4 gm = HF::from_u64(g);
        if (!HF::eq(pow(gm, e), HF::one())) return pow(gm, t);
      }
    }();
    H4 z = z0;
    H4 w = pow(a, half), x = HF::mul(a, w), b = HF::mul(x, w);
    int v = 32;
    while (!HF::eq(b, HF::one())) {
      int k = 0;
      H4 b2k = b;
      while (!HF::eq(b2k, HF::one())) {
        b2k = HF::sqr(b2k);
        if (++k == v) return false;
      }
      H4 ww = z;
      for (int j = 0; j < v - k - 1; j++) ww = HF::sqr(ww);
      z = HF::sqr(ww);
      b = HF::mul(b, z);
      x = HF::mul(x, ww);
      v = k;
    }
    root = x;
    return true;
  }
  static bool canon_less(const H4& a, const H4& b) {
    H4 ca = HF::to_canon(a), cb = HF::to_canon(b);
    for (int i = 3; i >= 0; i--) {
      if (ca.v[i] != cb.v[i]) return ca.v[i] < cb.v[i];
    }
    return false;
  }
};

// ---- ark-marlin 0.3 FiatShamirRng<Blake2s> (src/lib.rs:5) ----------------------------------
class FiatShamirRng {
 public:
  static FiatShamirRng from_seed(const uint8_t* bytes, size_t len) { return FiatShamirRng(Blake2s::digest(bytes, len)); }
  void absorb(const uint8_t* bytes, size_t len) {
    std::vector<uint8_t> buf(bytes, bytes + len);
    buf.insert(buf.end(), seed_.begin(), seed_.end());  // new material first, then the old seed
    seed_ = Blake2s::digest(buf.data(), buf.size());
    r_ = ChaCha20Rng(seed_);
  }
  template <int CURVE>
  void absorb(const G& g) {
    uint8_t b[96];
    PointOps<CURVE>::to_bytes(g, b);
    absorb(b, 96);
  }
  uint32_t next_u32() { return r_.next_u32(); }
  uint64_t next_u64() { return r_.next_u64(); }
  // G::ScalarField::rand(fs): 4 x next_u64, clear the top bit, accept iff < r; limbs are the Montgomery repr
  template <int CURVE>
  Fr rand_scalar() {
    using C = bpgpu::FieldConsts<1 - CURVE>;
    for (;;) {
      Fr x;
      for (int i = 0; i < 4; i++) x.v[i] = next_u64();
      x.v[3] &= 0x7fffffffffffffffull;
      for (int i = 3; i >= 0; i--) {
        if (x.v[i] < C::P64[i]) return x;
        if (x.v[i] > C::P64[i]) break;
      }
    }
  }

 private:
  explicit FiatShamirRng(const std::array<uint8_t, 32>& seed) : seed_(seed), r_(seed) {}
  std::array<uint8_t, 32> seed_;
  ChaCha20Rng r_;
};

// ---- util (src/util/mod.rs) -----------------------------------------------------------------
namespace util {
// pippenger_msm: zips the two sequences (msm.rs:94-97), so the shorter one decides the length
template <int CURVE>
G msm(const G* bases, size_t nb, const Fr* scalars, size_t ns) {
  size_t n = nb < ns ? nb : ns;
  uint64_t out[9];
  check(bpgpu_msm(CURVE, n ? bases->v : nullptr, n ? scalars->v : nullptr, n, out));
  return PointOps<CURVE>::from_xy_inf(out);
}
template <int CURVE>
G msm(const std::vector<G>& bases, const std::vector<Fr>& scalars) {
  return msm<CURVE>(bases.data(), bases.size(), scalars.data(), scalars.size());
}
template <int CURVE>
Fr ip(const Fr* a, size_t na, const Fr* b, size_t nb) {
  if (na != nb) throw Panic("assertion failed: a.len() == b.len()");  // util/mod.rs:18
  Fr out;
  check(bpgpu_ip(CURVE, na ? a->v : nullptr, na ? b->v : nullptr, na, out.v));
  return out;
}
inline size_t next_power_of_two(size_t n) {
  size_t p = 1;
  while (p < n) p <<= 1;
  return p;
}
template <class Z>
std::vector<Z> zero_pad_to_two_power(const std::vector<Z>& v, const Z& zero) {  // util/mod.rs:137-144
  std::vector<Z> out(v);
  out.resize(next_power_of_two(v.size()), zero);
  return out;
}
// util/mod.rs:103-111 (like the reference, undefined for an empty array: (len - 1) underflows there)
template <class Z>
std::vector<Z> zero_pad_to_multiple(const std::vector<Z>& v, size_t k, const Z& zero) {
  if (v.empty() || k == 0) throw Panic("attempt to subtract with overflow / divide by zero");
  std::vector<Z> out(v);
  const size_t want = ((v.size() - 1) / k + 1) * k;
  if (want > out.size()) out.resize(want, zero);
  return out;
}
// util/mod.rs:117-134
template <class Z>
std::vector<Z> zero_intersperse_to_multiple(const std::vector<Z>& v, size_t k, const Z& zero) {
  if (v.empty() || k == 0) throw Panic("attempt to subtract with overflow / divide by zero");
  const size_t block = (v.size() - 1) / k + 1, fin = block * k, n_short = fin - v.size();
  std::vector<Z> out;
  out.reserve(fin);
  for (size_t i = 0; i < k; i++) {
    if (i >= k - n_short) {
      const size_t start = block * (k - n_short) + (block - 1) * (i + n_short - k);
      out.insert(out.end(), v.begin() + start, v.begin() + start + block - 1);
      out.push_back(zero);
    } else {
      out.insert(out.end(), v.begin() + block * i, v.begin() + block * (i + 1));
    }
  }
  if (out.size() != fin) throw Panic("assertion failed: `(left == right)` in zero_intersperse_to_multiple");
  return out;
}
// powers(f, start..end), util/mod.rs:57-64
template <int CURVE>
std::vector<Fr> powers(const Fr& f, size_t start, size_t end) {
  std::vector<Fr> out(end > start ? end - start : 0);
  check(bpgpu_powers(CURVE, f.v, start, out.size(), out.empty() ? nullptr : out.data()->v));
  return out;
}
template <int CURVE>
std::vector<Fr> neg_powers(const Fr& f, size_t start, size_t end) {  // util/mod.rs:66-68
  return powers<CURVE>(ScalarOps<CURVE>::inverse(f), start, end);
}
// The k-ary fold  sum_vecs(chunks.zip(y).map(|(c, y)| scale_vec(y, c)), ck)  in one call
// (bp_rec_kary.rs:100-112): v holds the k chunks back to back.
template <int CURVE>
std::vector<Fr> fold_chunks(const std::vector<Fr>& v, size_t k, const std::vector<Fr>& y) {
  if (y.size() < k || (k && v.size() % k)) throw Panic("assertion failed: `(left == right)` (summand.len(), len)");
  std::vector<Fr> out(k ? v.size() / k : 0);
  check(bpgpu_fold_scalars_kary(CURVE, v.empty() ? nullptr : v.data()->v, out.size(), k, k ? y.data()->v : nullptr,
                                out.empty() ? nullptr : out.data()->v));
  return out;
}
template <int CURVE>
std::vector<G> fold_chunks(const std::vector<G>& v, size_t k, const std::vector<Fr>& y) {
  if (y.size() < k || (k && v.size() % k)) throw Panic("assertion failed: `(left == right)` (summand.len(), len)");
  std::vector<G> out(k ? v.size() / k : 0);
  check(bpgpu_fold_points_kary(CURVE, v.empty() ? nullptr : v.data()->v, out.size(), k, k ? y.data()->v : nullptr,
                               out.empty() ? nullptr : out.data()->v));
  return out;
}
// Cross terms of a k-ary round (bp_rec_kary.rs:52-76 == ipa_to_bp_unroll.rs:189-215) over padded vectors of k chunks
template <int CURVE>
void kary_cross_terms(const std::vector<G>& a_gen, const std::vector<G>& b_gen, const G& q, const std::vector<Fr>& a,
                      const std::vector<Fr>& b, size_t ck, size_t k, std::vector<G>& pos, std::vector<G>& neg) {
  const size_t m = k > 0 ? k - 1 : 0;
  std::vector<uint64_t> po(9 * (m ? m : 1)), ne(9 * (m ? m : 1));
  check(bpgpu_kary_cross_terms(CURVE, a_gen.empty() ? nullptr : a_gen.data()->v, b_gen.empty() ? nullptr : b_gen.data()->v, q.v,
                               a.empty() ? nullptr : a.data()->v, b.empty() ? nullptr : b.data()->v, ck, k, po.data(), ne.data()));
  pos.clear();
  neg.clear();
  for (size_t t = 0; t < m; t++) {
    pos.push_back(PointOps<CURVE>::from_xy_inf(&po[9 * t]));
    neg.push_back(PointOps<CURVE>::from_xy_inf(&ne[9 * t]));
  }
}
// scale_vec (util/mod.rs:25-38) and sum_vecs (:40-55) on their own: the fold with one chunk, and the
// fold with every multiplier one
template <int CURVE, class Z>
std::vector<Z> scale_vec(const Fr& s, const std::vector<Z>& b) {
  return fold_chunks<CURVE>(b, 1, std::vector<Fr>{s});
}
template <int CURVE, class Z>
std::vector<Z> sum_vecs(const std::vector<std::vector<Z>>& vs, size_t len, const Z& zero) {
  std::vector<Z> flat;
  for (const auto& v : vs) {
    if (v.size() != len) throw Panic("assertion failed: `(left == right)` (summand.len(), len)");
    flat.insert(flat.end(), v.begin(), v.end());
  }
  if (vs.empty()) return std::vector<Z>(len, zero);  // vec![F::zero(); len]
  return fold_chunks<CURVE>(flat, vs.size(), std::vector<Fr>(vs.size(), ScalarOps<CURVE>::one()));
}
template <int CURVE>
std::vector<Fr> hadamard(const std::vector<Fr>& a, const std::vector<Fr>& b) {  // util/mod.rs:80-85
  if (a.size() != b.size()) throw Panic("assertion failed: `(left == right)` (a.len(), b.len())");
  std::vector<Fr> out(a.size());
  check(bpgpu_hadamard(CURVE, a.empty() ? nullptr : a.data()->v, b.empty() ? nullptr : b.data()->v, a.size(),
                       out.empty() ? nullptr : out.data()->v));
  return out;
}
template <int CURVE>
std::vector<G> hadamard_gp(const std::vector<G>& bases, const std::vector<Fr>& scalars) {  // util/mod.rs:70-78
  if (bases.size() != scalars.size()) throw Panic("assertion failed: `(left == right)` (bases.len(), scalars.len())");
  std::vector<G> out(bases.size());
  check(bpgpu_hadamard_points(CURVE, bases.empty() ? nullptr : bases.data()->v, scalars.empty() ? nullptr : scalars.data()->v,
                              bases.size(), out.empty() ? nullptr : out.data()->v));
  return out;
}
// the naive msm (util/msm.rs:13-24, what `msm_bench` runs by default): n independent base.mul(scalar)
// on the device (hadamard_gp) and the sum of the results; `zip` truncates to the shorter input
template <int CURVE>
G naive_msm(const std::vector<G>& bases, const std::vector<Fr>& scalars);
// sum of group elements (Iterator::sum over G), through the library's point adder
template <int CURVE>
G sum_points(const std::vector<G>& v) {
  std::vector<uint64_t> xy(9 * (v.size() ? v.size() : 1), 0);
  for (size_t i = 0; i < v.size(); i++) {
    const bool inf = (v[i].v[8] | v[i].v[9] | v[i].v[10] | v[i].v[11]) == 0;
    if (inf) xy[9 * i + 8] = 1; else std::memcpy(&xy[9 * i], v[i].v, 64);  // inputs are normalised (Z = 1)
  }
  uint64_t out[9];
  check(bpgpu_points_sum(CURVE, xy.data(), v.size(), out));
  return PointOps<CURVE>::from_xy_inf(out);
}
template <int CURVE>
G naive_msm(const std::vector<G>& bases, const std::vector<Fr>& scalars) {
  const size_t n = bases.size() < scalars.size() ? bases.size() : scalars.size();
  std::vector<G> prods(n);
  check(bpgpu_hadamard_points(CURVE, n ? bases.data()->v : nullptr, n ? scalars.data()->v : nullptr, n,
                              n ? prods.data()->v : nullptr));
  return sum_points<CURVE>(prods);
}
// bos_coster_msm (util/msm.rs:34-81): a serial max-heap algorithm, CPU-only in the reference and here (SURVEY 8
// row a3: it has no data-parallel form; it exists for `msm_bench -m boscoster`).  Entries (x, g) are ordered
// by x alone (msm.rs:26-32); zero scalars are skipped (:42); the loop pops the two largest and either halves
// the first (:57-67) or subtracts the second from it while adding the points (:68-75); the last entry is
// multiplied out (:77-80).  Group operations run on the host (host_field.h); the result is normalised like
// every other group element that leaves this interface.
template <int CURVE>
G bos_coster_msm(const std::vector<G>& bases, const std::vector<Fr>& scalars) {
  using HFs = bpgpu::host::HF<1 - CURVE>;
  using HFq = bpgpu::host::HF<CURVE>;
  using E = bpgpu::host::HEc<CURVE>;
  using P = bpgpu::host::HPoint<CURVE>;
  using H4 = bpgpu::host::H4;
  struct Entry {
    H4 x;  // canonical scalar (into_repr)
    P g;
  };
  auto cmp = [](const H4& a, const H4& b) {
    for (int i = 3; i >= 0; i--)
      if (a.v[i] != b.v[i]) return a.v[i] < b.v[i] ? -1 : 1;
    return 0;
  };
  auto less = [&](const Entry& a, const Entry& b) { return cmp(a.x, b.x) < 0; };
  auto div2 = [](H4& a) {
    for (int i = 0; i < 4; i++) a.v[i] = (a.v[i] >> 1) | (i < 3 ? a.v[i + 1] << 63 : 0);
  };
  auto is_zero = [](const H4& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; };
  auto load = [](const uint64_t* w) {
    H4 r;
    std::memcpy(r.v, w, 32);
    return r;
  };
  std::vector<Entry> heap;
  const size_t n = bases.size() < scalars.size() ? bases.size() : scalars.size();  // zip
  for (size_t i = 0; i < n; i++) {
    const H4 x = HFs::to_canon(load(scalars[i].v));
    if (is_zero(x)) continue;
    heap.push_back(Entry{x, E::from_jacobian(load(bases[i].v), load(bases[i].v + 4), load(bases[i].v + 8))});
    std::push_heap(heap.begin(), heap.end(), less);
  }
  auto pop = [&] {
    std::pop_heap(heap.begin(), heap.end(), less);
    Entry e = heap.back();
    heap.pop_back();
    return e;
  };
  auto push = [&](const Entry& e) {
    heap.push_back(e);
    std::push_heap(heap.begin(), heap.end(), less);
  };
  while (heap.size() > 1) {
    Entry first = pop(), second = pop();
    H4 half_first = first.x;
    div2(half_first);
    if (cmp(half_first, second.x) > 0) {
      push(second);
      if (first.x.v[0] & 1) push(Entry{H4{{1, 0, 0, 0}}, first.g});
      div2(first.x);
      first.g = E::dbl(first.g);
      push(first);
    } else {
      unsigned __int128 br = 0;
      for (int i = 0; i < 4; i++) {
        unsigned __int128 d = (unsigned __int128)first.x.v[i] - second.x.v[i] - br;
        first.x.v[i] = (uint64_t)d;
        br = (d >> 64) & 1;
      }
      if (br) throw Panic("assertion failed: !first.x.sub_noborrow(&second.x)");  // msm.rs:69
      second.g = E::add(first.g, second.g);
      push(second);
      if (!is_zero(first.x)) push(first);
    }
  }
  P total = E::identity();
  if (!heap.empty()) total = E::mul_canon(heap[0].g, heap[0].x);
  G out;
  H4 x, y;
  const H4 one = HFq::one();
  if (E::to_affine(total, x, y)) {
    std::memcpy(out.v, x.v, 32);
    std::memcpy(out.v + 4, y.v, 32);
    std::memcpy(out.v + 8, one.v, 32);
  } else {  // zero() = (0, 1, 0)
    std::memset(out.v, 0, 96);
    std::memcpy(out.v + 4, one.v, 32);
  }
  return out;
}
// ---- sparse products of the Com-R1CS compiler (src/relations/com_r1cs.rs:43-83) -------------------
// ark-relations' Matrix<F>: a list of rows of (value, column) pairs
using Matrix = std::vector<std::vector<std::pair<Fr, size_t>>>;
struct Csr {
  std::vector<uint64_t> row_ptr, col_idx;
  std::vector<Fr> vals;
  explicit Csr(const Matrix& mat) : row_ptr(mat.size() + 1, 0) {
    for (size_t r = 0; r < mat.size(); r++) {
      row_ptr[r + 1] = row_ptr[r] + mat[r].size();
      for (const auto& e : mat[r]) {
        vals.push_back(e.first);
        col_idx.push_back(e.second);
      }
    }
  }
};
inline size_t mat_cols(const Matrix& mat) {  // com_r1cs.rs:59-65
  size_t m = 0;
  for (const auto& row : mat)
    for (const auto& e : row) m = e.second + 1 > m ? e.second + 1 : m;
  return m;
}
// mat_vec_mult(&mat, &vec), com_r1cs.rs:43-57 (panics like the reference's assert!(c < vec.len()))
template <int CURVE>
std::vector<Fr> mat_vec_mult(const Matrix& mat, const std::vector<Fr>& vec) {
  const Csr m(mat);
  std::vector<Fr> out(mat.size());
  check(bpgpu_mat_vec_mult(CURVE, m.row_ptr.data(), m.col_idx.data(), m.vals.empty() ? nullptr : m.vals.data()->v, mat.size(),
                           vec.empty() ? nullptr : vec.data()->v, vec.size(), out.empty() ? nullptr : out.data()->v));
  return out;
}
// vec_mat_mult(&vec, &mat, output_size), com_r1cs.rs:67-83
template <int CURVE>
std::vector<Fr> vec_mat_mult(const std::vector<Fr>& vec, const Matrix& mat, size_t output_size) {
  const Csr m(mat);
  std::vector<Fr> out(output_size);
  check(bpgpu_vec_mat_mult(CURVE, vec.empty() ? nullptr : vec.data()->v, vec.size(), m.row_ptr.data(), m.col_idx.data(),
                           m.vals.empty() ? nullptr : m.vals.data()->v, mat.size(), output_size,
                           out.empty() ? nullptr : out.data()->v));
  return out;
}
}  // namespace util

// ---- relations (src/relations/ipa.rs) ---------------------------------------------------------
struct IpaGens {
  size_t vec_size = 0;
  std::vector<G> a_gens, b_gens;
  G ip_gen{};
};
struct IpaInstance {
  IpaGens gens;
  G result{};
};
struct IpaWitness {
  std::vector<Fr> a, b;
};

// IpaGens held in HBM across proofs.  The generators are fixed public parameters
// (src/relations/ipa.rs:28-36 samples them once per size), so a prover that makes many proofs
// uploads them once; Bp::prove then moves only the witness.  No counterpart in the reference,
// which keeps everything in host memory.
struct DeviceGens {
  int curve;
  size_t n;
  void *a = nullptr, *b = nullptr, *q = nullptr;
  DeviceGens(int curve_, const IpaGens& g) : curve(curve_), n(g.a_gens.size()) {
    if (g.b_gens.size() != n) throw Panic("assertion failed: vector sizes");
    try {
      check(bpgpu_dev_alloc((n ? n : 1) * 64, &a));
      check(bpgpu_dev_alloc((n ? n : 1) * 64, &b));
      check(bpgpu_dev_alloc(64, &q));
      if (n) {
        check(bpgpu_points_to_device(curve, g.a_gens.data()->v, n, 0, a));
        check(bpgpu_points_to_device(curve, g.b_gens.data()->v, n, 0, b));
      }
      check(bpgpu_points_to_device(curve, g.ip_gen.v, 1, 0, q));
    } catch (...) {
      release();
      throw;
    }
  }
  DeviceGens(const DeviceGens&) = delete;
  DeviceGens& operator=(const DeviceGens&) = delete;
  ~DeviceGens() { release(); }

 private:
  void release() {
    if (a) (void)bpgpu_dev_free(a);
    if (b) (void)bpgpu_dev_free(b);
    if (q) (void)bpgpu_dev_free(q);
    a = b = q = nullptr;
  }
};

// ---- sampling (UniformRand), in the reference's draw order -------------------------------------
// G::ScalarField::rand
template <int CURVE, class Rng>
Fr rand_scalar(Rng& rng) {
  Fr r;
  bpgpu::host::H4 x = rand_field<1 - CURVE>(rng);
  std::memcpy(r.v, x.v, 32);
  return r;
}
// GroupProjective::rand for y^2 = x^3 + 5 (ark-ec 0.3 short Weierstrass): loop { x = Fq::rand;
// greatest = rng.gen::<bool>() (top bit of one next_u32); if x^3 + 5 is a square pick y or -y by
// (y < -y) XOR greatest on canonical integers }; cofactor 1, so Z = 1.
template <int CURVE, class Rng>
G rand_point(Rng& rng) {
  using HF = bpgpu::host::HF<CURVE>;
  using H4 = bpgpu::host::H4;
  const H4 five = HF::from_u64(5);
  for (;;) {
    const H4 x = rand_field<CURVE>(rng);
    const bool greatest = (rng.next_u32() >> 31) != 0;
    H4 y;
    if (!FieldSqrt<CURVE>::sqrt(HF::add(HF::mul(HF::sqr(x), x), five), y)) continue;
    const H4 negy = HF::neg(y);
    const H4 pick = (FieldSqrt<CURVE>::canon_less(y, negy) != greatest) ? y : negy;
    G g;
    std::memcpy(g.v, x.v, 32);
    std::memcpy(g.v + 4, pick.v, 32);
    std::memcpy(g.v + 8, HF::one().v, 32);
    return g;
  }
}
// rng.gen::<[u8; 32]>(): one next_u32 per byte (src/lib.rs:117)
template <class Rng>
std::array<uint8_t, 32> gen_seed32(Rng& rng) {
  std::array<uint8_t, 32> s;
  for (auto& b : s) b = (uint8_t)rng.next_u32();
  return s;
}

// n points by G::rand in one go: the stream is walked on the host (an attempt's draws do not depend on
// its outcome), the square roots of all candidates are taken on the device (bpgpu_curve_lift_x), and the
// generator is left exactly where n sequential rand_point calls would leave it.
template <int CURVE, class Rng>
std::vector<G> rand_points(Rng& rng, size_t n) {
  using HF = bpgpu::host::HF<CURVE>;
  std::vector<G> out;
  out.reserve(n);
  while (out.size() < n) {
    const size_t want = n - out.size(), m = 2 * want + want / 8 + 64;  // about half of the attempts succeed
    const Rng start = rng;
    std::vector<bpgpu::host::H4> xs(m), ys(m);
    std::vector<uint8_t> greatest(m), ok(m);
    for (size_t i = 0; i < m; i++) {
      xs[i] = rand_field<CURVE>(rng);
      greatest[i] = (uint8_t)(rng.next_u32() >> 31);
    }
    check(bpgpu_curve_lift_x(CURVE, xs.data()->v, greatest.data(), m, ys.data()->v, ok.data()));
    size_t used = m;
    for (size_t i = 0; i < m; i++) {
      if (!ok[i]) continue;
      G g;
      std::memcpy(g.v, xs[i].v, 32);
      std::memcpy(g.v + 4, ys[i].v, 32);
      std::memcpy(g.v + 8, HF::one().v, 32);
      out.push_back(g);
      if (out.size() == n) {
        used = i + 1;
        break;
      }
    }
    if (used < m) {  // rewind to the state right after the last attempt that was needed
      rng = start;
      for (size_t i = 0; i < used; i++) {
        (void)rand_field<CURVE>(rng);
        (void)rng.next_u32();
      }
    }
  }
  return out;
}

template <int CURVE>
G zero_point() {
  uint64_t o[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1};
  return PointOps<CURVE>::from_xy_inf(o);
}

template <int CURVE>
struct IpaRelation {
  // commit_for_witness (ipa.rs:37-41) as one MSM over a_gens | b_gens | ip_gen
  static G commit_for_witness(const IpaGens& g, const IpaWitness& w) {
    std::vector<G> pts(g.a_gens);
    pts.insert(pts.end(), g.b_gens.begin(), g.b_gens.end());
    pts.push_back(g.ip_gen);
    std::vector<Fr> sc(w.a);
    sc.insert(sc.end(), w.b.begin(), w.b.end());
    sc.push_back(util::ip<CURVE>(w.a.data(), w.a.size(), w.b.data(), w.b.size()));
    return util::msm<CURVE>(pts, sc);
  }
  // check (ipa.rs:134-146)
  static void check(const IpaInstance& x, const IpaWitness& w) {
    if (x.gens.a_gens.size() != x.gens.vec_size || x.gens.b_gens.size() != x.gens.vec_size ||
        w.a.size() != x.gens.vec_size || w.b.size() != x.gens.vec_size)
      throw Panic("assertion failed: vector sizes");
    if (!(commit_for_witness(x.gens, w) == x.result)) throw Panic("assertion failed: commitment");
  }
};

// IpaInstance::sample_from_length (ipa.rs:28-36,90-97,110-115): a_gens, b_gens, ip_gen, then a, b;
// the result is commit_for_witness
template <int CURVE, class Rng>
std::pair<IpaInstance, IpaWitness> sample_from_length(Rng& rng, size_t length) {
  IpaInstance x;
  IpaWitness w;
  x.gens.vec_size = length;
  std::vector<G> gs = rand_points<CURVE>(rng, 2 * length + 1);  // a_gens, b_gens, ip_gen: consecutive draws
  x.gens.a_gens.assign(gs.begin(), gs.begin() + length);
  x.gens.b_gens.assign(gs.begin() + length, gs.begin() + 2 * length);
  x.gens.ip_gen = gs[2 * length];
  for (size_t i = 0; i < length; i++) w.a.push_back(rand_scalar<CURVE>(rng));
  for (size_t i = 0; i < length; i++) w.b.push_back(rand_scalar<CURVE>(rng));
  x.result = IpaRelation<CURVE>::commit_for_witness(x.gens, w);
  return {x, w};
}

// ---- proofs::bp_iter (src/proofs/bp_iter.rs) --------------------------------------------------
struct BpProof {
  std::vector<G> ls, rs;
  Fr final_a{}, final_b{};
};

template <int CURVE>
struct Bp {
  using S = ScalarOps<CURVE>;
  using P = PointOps<CURVE>;

  // bp_iter.rs:49-128.  The vectors go to HBM once; per round only L, R come back and
  // x, x^-1 go down.  Optionally reports the challenges (not part of the reference's proof).
  // `cached`, when given, must hold instance.gens (DeviceGens above): the generators are then not
  // uploaded again.
  static BpProof prove(const IpaInstance& instance, const IpaWitness& witness, FiatShamirRng& fs,
                       std::vector<Fr>* challenges_out = nullptr, const DeviceGens* cached = nullptr) {
    const size_t n = witness.a.size();
    if (witness.b.size() != n || instance.gens.a_gens.size() != n || instance.gens.b_gens.size() != n)
      throw Panic("assertion failed: vector sizes");
    bpgpu_ipa* sess = nullptr;
    if (cached) {
      if (cached->n != n || cached->curve != CURVE) throw Panic("assertion failed: cached generators do not match");
      check(bpgpu_ipa_begin_device(CURVE, cached->a, cached->b, cached->q, n ? witness.a.data()->v : nullptr,
                                   n ? witness.b.data()->v : nullptr, n, &sess));
    } else {
      check(bpgpu_ipa_begin(CURVE, n ? instance.gens.a_gens.data()->v : nullptr,
                            n ? instance.gens.b_gens.data()->v : nullptr, instance.gens.ip_gen.v,
                            n ? witness.a.data()->v : nullptr, n ? witness.b.data()->v : nullptr, n, &sess));
    }
    BpProof proof;
    try {
      while (bpgpu_ipa_len(sess) > 1) {
        uint64_t lo[9], ro[9];
        check(bpgpu_ipa_round_lr(sess, lo, ro));
        G l = P::from_xy_inf(lo), r = P::from_xy_inf(ro);
        proof.ls.push_back(l);
        proof.rs.push_back(r);
        fs.absorb<CURVE>(l);
        fs.absorb<CURVE>(r);
        Fr x = fs.rand_scalar<CURVE>();
        Fr x_inv = S::inverse(x);
        if (challenges_out) challenges_out->push_back(x);
        check(bpgpu_ipa_round_fold(sess, x.v, x_inv.v));
      }
      check(bpgpu_ipa_finish(sess, proof.final_a.v, proof.final_b.v));
    } catch (...) {
      bpgpu_ipa_free(sess);
      throw;
    }
    bpgpu_ipa_free(sess);
    return proof;
  }

  // bp_iter.rs:25-42 (iterative form of the same recursion)
  static std::vector<Fr> scalars_from_challenges(const std::vector<Fr>& challs) {
    std::vector<Fr> left{S::one()};
    for (size_t k = challs.size(); k-- > 0;) {
      Fr ch = challs[k], ch_inv = S::inverse(ch);
      size_t n = left.size();
      left.resize(2 * n);
      for (size_t i = 0; i < n; i++) {
        left[n + i] = S::mul(left[i], ch);
        left[i] = S::mul(left[i], ch_inv);
      }
    }
    return left;
  }

  // bp_iter.rs:130-178: panics (throws) unless the single big MSM reproduces the commitment.  The
  // transcript is replayed here; the 2N + 1 + 2 log2 N scalars (scalars_from_challenges above and its
  // inverse twin, :160-173) are expanded on the device and the MSM runs there (bpgpu_ipa_verify_msm).
  // `cached`, when given, must hold instance.gens.
  static void verify(const IpaInstance& instance, const BpProof& proof, FiatShamirRng& fs,
                     const DeviceGens* cached = nullptr) {
    const size_t n = instance.gens.a_gens.size(), k = proof.ls.size();
    if (instance.gens.b_gens.size() != n || proof.rs.size() != k) throw Panic("assertion failed: vector sizes");
    std::vector<Fr> challenges;
    for (size_t i = 0; i < k; i++) {
      fs.absorb<CURVE>(proof.ls[i]);
      fs.absorb<CURVE>(proof.rs[i]);
      challenges.push_back(fs.rand_scalar<CURVE>());
    }
    if (k > 30 || util::next_power_of_two(n < 1 ? 1 : n) != ((size_t)1 << k))
      throw Panic("assertion failed: proof length does not match the generators");
    uint64_t out[9];
    const uint64_t* ls = k ? proof.ls.data()->v : nullptr;
    const uint64_t* rs = k ? proof.rs.data()->v : nullptr;
    const uint64_t* ch = k ? challenges.data()->v : nullptr;
    if (cached) {
      if (cached->n != n || cached->curve != CURVE) throw Panic("assertion failed: cached generators do not match");
      check(bpgpu_ipa_verify_msm_device(CURVE, cached->a, cached->b, cached->q, n, ls, rs, k, ch, proof.final_a.v,
                                        proof.final_b.v, out));
    } else {
      check(bpgpu_ipa_verify_msm(CURVE, n ? instance.gens.a_gens.data()->v : nullptr,
                                 n ? instance.gens.b_gens.data()->v : nullptr, instance.gens.ip_gen.v, n, ls, rs, k, ch,
                                 proof.final_a.v, proof.final_b.v, out));
    }
    if (!(instance.result == P::from_xy_inf(out))) throw Panic("assertion failed: `(left == right)` in Bp::verify");
  }
  static size_t proof_size(const BpProof& p) { return 2 * (1 + p.ls.size()); }  // bp_iter.rs:187-189
};

// ---- reductions::bp_2ary_step (src/reductions/bp_2ary_step.rs) --------------------------------
template <int CURVE>
struct Bp2aryStep {
  using S = ScalarOps<CURVE>;
  using P = PointOps<CURVE>;
  struct Proof {
    G l, r;
  };
  struct ProveOut {
    Proof proof;
    IpaInstance instance;
    IpaWitness witness;
  };
  static G p_next(const G& l, const G& r, const G& p, const Fr& x, const Fr& x_inv) {  // :59,105
    G pts[3] = {l, r, p};
    Fr sc[3] = {S::square(x), S::square(x_inv), S::one()};
    return util::msm<CURVE>(pts, 3, sc, 3);
  }
  // :24-87
  static ProveOut prove(const IpaInstance& instance, const IpaWitness& witness, FiatShamirRng& fs) {
    const size_t n0 = witness.a.size();
    if (util::next_power_of_two(n0) % 2 != 0) throw Panic("assertion failed: a.len() % 2 == 0");
    bpgpu_ipa* sess = nullptr;
    check(bpgpu_ipa_begin(CURVE, n0 ? instance.gens.a_gens.data()->v : nullptr, n0 ? instance.gens.b_gens.data()->v : nullptr,
                          instance.gens.ip_gen.v, n0 ? witness.a.data()->v : nullptr, n0 ? witness.b.data()->v : nullptr, n0,
                          &sess));
    ProveOut out;
    try {
      uint64_t lo[9], ro[9];
      check(bpgpu_ipa_round_lr(sess, lo, ro));
      out.proof.l = P::from_xy_inf(lo);
      out.proof.r = P::from_xy_inf(ro);
      fs.absorb<CURVE>(out.proof.l);
      fs.absorb<CURVE>(out.proof.r);
      Fr x = fs.rand_scalar<CURVE>(), x_inv = S::inverse(x);
      check(bpgpu_ipa_round_fold(sess, x.v, x_inv.v));
      const size_t n = bpgpu_ipa_len(sess);
      out.instance.gens.vec_size = n;
      out.instance.gens.ip_gen = instance.gens.ip_gen;
      out.instance.gens.a_gens.resize(n);
      out.instance.gens.b_gens.resize(n);
      out.witness.a.resize(n);
      out.witness.b.resize(n);
      check(bpgpu_ipa_export(sess, out.instance.gens.a_gens.data()->v, out.instance.gens.b_gens.data()->v,
                             out.witness.a.data()->v, out.witness.b.data()->v));
      out.instance.result = p_next(out.proof.l, out.proof.r, instance.result, x, x_inv);
    } catch (...) {
      bpgpu_ipa_free(sess);
      throw;
    }
    bpgpu_ipa_free(sess);
    return out;
  }
  // :88-125
  static IpaInstance verify(const IpaInstance& instance, const Proof& proof, FiatShamirRng& fs) {
    std::vector<G> a_gen = util::zero_pad_to_two_power(instance.gens.a_gens, zero_point<CURVE>());
    std::vector<G> b_gen = util::zero_pad_to_two_power(instance.gens.b_gens, zero_point<CURVE>());
    if (a_gen.size() % 2 != 0) throw Panic("assertion failed: a_gen.len() % 2 == 0");
    const size_t n = a_gen.size() / 2;
    fs.absorb<CURVE>(proof.l);
    fs.absorb<CURVE>(proof.r);
    Fr x = fs.rand_scalar<CURVE>(), x_inv = S::inverse(x);
    IpaInstance next;
    next.gens.vec_size = n;
    next.gens.ip_gen = instance.gens.ip_gen;
    next.gens.a_gens.resize(n);
    next.gens.b_gens.resize(n);
    check(bpgpu_fold_points(CURVE, a_gen.data()->v, n, x_inv.v, x.v, next.gens.a_gens.data()->v));
    check(bpgpu_fold_points(CURVE, b_gen.data()->v, n, x.v, x_inv.v, next.gens.b_gens.data()->v));
    next.result = p_next(proof.l, proof.r, instance.result, x, x_inv);
    return next;
  }
};

// ---- proofs::bp_rec_kary (src/proofs/bp_rec_kary.rs): one k-ary round ---------------------------
template <int CURVE>
struct KaryBp {
  using S = ScalarOps<CURVE>;
  using P = PointOps<CURVE>;
  struct StepOut {
    IpaInstance instance;
    IpaWitness witness;
    std::vector<G> pos_cross, neg_cross;
  };
  static void absorb_vec(FiatShamirRng& fs, const std::vector<G>& v) {  // ToBytes for Vec<G>: elements back to back
    std::vector<uint8_t> bytes(96 * v.size());
    for (size_t i = 0; i < v.size(); i++) P::to_bytes(v[i], bytes.data() + 96 * i);
    fs.absorb(bytes.data(), bytes.size());
  }
  // p_next = msm(pos | neg, x_pows[1..] | x_inv_pows[1..]) + p   (:113-116, :179-182)
  static G p_next(const std::vector<G>& pos, const std::vector<G>& neg, const std::vector<Fr>& xp,
                  const std::vector<Fr>& xip, const G& p) {
    std::vector<G> pts(pos);
    pts.insert(pts.end(), neg.begin(), neg.end());
    std::vector<Fr> sc(xp.begin() + 1, xp.end());
    sc.insert(sc.end(), xip.begin() + 1, xip.end());
    pts.push_back(p);
    sc.push_back(S::one());
    return util::msm<CURVE>(pts, sc);
  }
  // prove_step, :27-143
  static StepOut prove_step(size_t k, const IpaInstance& instance, const IpaWitness& witness, FiatShamirRng& fs) {
    Fr zero{};
    std::vector<Fr> a = util::zero_pad_to_multiple(witness.a, k, zero), b = util::zero_pad_to_multiple(witness.b, k, zero);
    std::vector<G> a_gen = util::zero_pad_to_multiple(instance.gens.a_gens, k, zero_point<CURVE>());
    std::vector<G> b_gen = util::zero_pad_to_multiple(instance.gens.b_gens, k, zero_point<CURVE>());
    if (b.size() != a.size() || a_gen.size() != a.size() || b_gen.size() != a.size()) throw Panic("assertion failed: vector sizes");
    const size_t ck = a.size() / k;
    const G& q = instance.gens.ip_gen;
    // X[i, j] = a[i] * A[j] + b[j] * B[i] + <a[i], b[j]> * Q   (:52-60) and the sums over it (:64-76): one call, which
    // merges the k (k - 1) MSMs into 2 (k - 1) over contiguous ranges and runs them as one batch
    StepOut out;
    util::kary_cross_terms<CURVE>(a_gen, b_gen, q, a, b, ck, k, out.pos_cross, out.neg_cross);
    absorb_vec(fs, out.pos_cross);
    absorb_vec(fs, out.neg_cross);
    const Fr xc = fs.rand_scalar<CURVE>(), x_inv = S::inverse(xc);
    const std::vector<Fr> x_pows = util::powers<CURVE>(xc, 0, k), x_inv_pows = util::powers<CURVE>(x_inv, 0, k);
    out.witness.a = util::fold_chunks<CURVE>(a, k, x_pows);               // :100
    out.witness.b = util::fold_chunks<CURVE>(b, k, x_inv_pows);           // :101-104
    out.instance.gens.a_gens = util::fold_chunks<CURVE>(a_gen, k, x_inv_pows);  // :105-108
    out.instance.gens.b_gens = util::fold_chunks<CURVE>(b_gen, k, x_pows);      // :109-112
    out.instance.gens.vec_size = ck;
    out.instance.gens.ip_gen = q;
    out.instance.result = p_next(out.pos_cross, out.neg_cross, x_pows, x_inv_pows, instance.result);
    return out;
  }
  // verify_step, :145-190
  static IpaInstance verify_step(size_t k, const IpaInstance& instance, const std::vector<G>& pos_cross,
                                 const std::vector<G>& neg_cross, FiatShamirRng& fs) {
    std::vector<G> a_gen = util::zero_pad_to_multiple(instance.gens.a_gens, k, zero_point<CURVE>());
    std::vector<G> b_gen = util::zero_pad_to_multiple(instance.gens.b_gens, k, zero_point<CURVE>());
    if (pos_cross.size() + 1 != k || neg_cross.size() + 1 != k) throw Panic("assertion failed: cross-term count");
    const size_t ck = a_gen.size() / k;
    absorb_vec(fs, pos_cross);
    absorb_vec(fs, neg_cross);
    const Fr xc = fs.rand_scalar<CURVE>(), x_inv = S::inverse(xc);
    const std::vector<Fr> x_pows = util::powers<CURVE>(xc, 0, k), x_inv_pows = util::powers<CURVE>(x_inv, 0, k);
    IpaInstance next;
    next.gens.vec_size = ck;
    next.gens.ip_gen = instance.gens.ip_gen;
    next.gens.a_gens = util::fold_chunks<CURVE>(a_gen, k, x_inv_pows);
    next.gens.b_gens = util::fold_chunks<CURVE>(b_gen, k, x_pows);
    next.result = p_next(pos_cross, neg_cross, x_pows, x_inv_pows, instance.result);
    return next;
  }
};

}  // namespace ark_bp
