// Host-side 4x64 Montgomery arithmetic (R = 2^256, i.e. ark-ff's in-memory form) and
// XYZZ point arithmetic for the *serial tails* of the GPU pipeline: the Horner
// combine over <= 64 window sums (src/util/msm.rs:177-190 in the reference), the
// final normalisation to affine, challenge inversion and the joint-sparse-form
// recoding of fold scalars.  These are a few hundred field operations per call;
// a single CPU core does one in ~25 ns whereas a lone GPU wave needs ~1 us, so
// they run here while the wide phases run on the device.
#pragma once
#include <cstdint>
#include <cstring>

#include "pasta_consts.h"

namespace bpgpu {
namespace host {

typedef unsigned __int128 u128;

struct H4 {
  uint64_t v[4];
};

template <int FID>
struct HF {
  using C = FieldConsts<FID>;

  static H4 zero() { return H4{{0, 0, 0, 0}}; }
  static H4 one() { return H4{{C::H_ONE[0], C::H_ONE[1], C::H_ONE[2], C::H_ONE[3]}}; }
  static H4 modulus() { return H4{{C::P64[0], C::P64[1], C::P64[2], C::P64[3]}}; }
  static bool is_zero(const H4& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
  static bool eq(const H4& a, const H4& b) {
    return a.v[0] == b.v[0] && a.v[1] == b.v[1] && a.v[2] == b.v[2] && a.v[3] == b.v[3];
  }
  static bool geq_p(const H4& a) {
    for (int i = 3; i >= 0; i--) {
      if (a.v[i] > C::P64[i]) return true;
      if (a.v[i] < C::P64[i]) return false;
    }
    return true;
  }
  static void sub_p(H4& a) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)a.v[i] - C::P64[i] - br;
      a.v[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
  }
  static H4 add(const H4& a, const H4& b) {
    H4 r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)a.v[i] + b.v[i];
      r.v[i] = (uint64_t)c;
      c >>= 64;
    }
    if (geq_p(r)) sub_p(r);  // a + b < 2p < 2^256: no carry out
    return r;
  }
  static H4 sub(const H4& a, const H4& b) {
    H4 r;
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)a.v[i] - b.v[i] - br;
      r.v[i] = (uint64_t)d;
      br = (d >> 64) & 1;
    }
    if (br) {
      u128 c = 0;
      for (int i = 0; i < 4; i++) {
        c += (u128)r.v[i] + C::P64[i];
        r.v[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    return r;
  }
  static H4 neg(const H4& a) { return is_zero(a) ? a : sub(zero(), a); }
  static H4 dbl(const H4& a) { return add(a, a); }
  // Montgomery product a*b/2^256 mod p (CIOS)
  static H4 mul(const H4& a, const H4& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (u128)a.v[j] * b.v[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (uint64_t)c;
      t[5] = (uint64_t)(c >> 64);
      uint64_t m = t[0] * C::INV64;
      c = (u128)m * C::P64[0] + t[0];
      c >>= 64;
      for (int j = 1; j < 4; j++) {
        c += (u128)m * C::P64[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = t[5] + (uint64_t)(c >> 64);
    }
    H4 r{{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_p(r)) sub_p(r);
    return r;
  }
  static H4 sqr(const H4& a) { return mul(a, a); }
  static H4 from_u64(uint64_t x) {  // x -> x*2^256
    H4 t{{x, 0, 0, 0}};
    H4 r2{{C::H_R2[0], C::H_R2[1], C::H_R2[2], C::H_R2[3]}};
    return mul(t, r2);
  }
  static H4 to_canon(const H4& a) {  // x*2^256 -> x
    H4 o{{1, 0, 0, 0}};
    return mul(a, o);
  }
  static H4 from_canon(const H4& a) {
    H4 r2{{C::H_R2[0], C::H_R2[1], C::H_R2[2], C::H_R2[3]}};
    return mul(a, r2);
  }
  // a^(p-2)
  static H4 inv(const H4& a) {
    H4 e = modulus();
    e.v[0] -= 2;  // p == 1 mod 2^64?  p low limb ends in ...00000001, so no borrow beyond limb 0
    H4 r = one();
    bool started = false;
    for (int i = 3; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        if (started) r = sqr(r);
        if ((e.v[i] >> b) & 1) {
          r = started ? mul(r, a) : a;
          started = true;
        }
      }
    return r;
  }
};

// XYZZ point over field FID, coordinates in ark form.
template <int FID>
struct HPoint {
  H4 x, y, zz, zzz;
  bool inf;
};

template <int FID>
struct HEc {
  using F = HF<FID>;
  using P = HPoint<FID>;
  static P identity() { return P{F::zero(), F::zero(), F::zero(), F::zero(), true}; }
  static P from_affine(const H4& x, const H4& y) { return P{x, y, F::one(), F::one(), false}; }
  static P neg(const P& a) {
    P r = a;
    r.y = F::neg(a.y);
    return r;
  }
  static P dbl(const P& p) {
    if (p.inf) return p;
    H4 U = F::dbl(p.y);
    H4 V = F::sqr(U);
    H4 W = F::mul(U, V);
    H4 S = F::mul(p.x, V);
    H4 x2 = F::sqr(p.x);
    H4 M = F::add(F::dbl(x2), x2);
    H4 X3 = F::sub(F::sub(F::sqr(M), S), S);
    H4 Y3 = F::sub(F::mul(M, F::sub(S, X3)), F::mul(W, p.y));
    return P{X3, Y3, F::mul(V, p.zz), F::mul(W, p.zzz), false};
  }
  static P add(const P& a, const P& b) {
    if (a.inf) return b;
    if (b.inf) return a;
    H4 U1 = F::mul(a.x, b.zz), U2 = F::mul(b.x, a.zz);
    H4 S1 = F::mul(a.y, b.zzz), S2 = F::mul(b.y, a.zzz);
    H4 Pp = F::sub(U2, U1), R = F::sub(S2, S1);
    if (F::is_zero(Pp)) {
      if (F::is_zero(R)) return dbl(a);
      return identity();
    }
    H4 PP = F::sqr(Pp), PPP = F::mul(Pp, PP), Q = F::mul(U1, PP);
    H4 X3 = F::sub(F::sub(F::sub(F::sqr(R), PPP), Q), Q);
    H4 Y3 = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(S1, PPP));
    return P{X3, Y3, F::mul(F::mul(a.zz, b.zz), PP), F::mul(F::mul(a.zzz, b.zzz), PPP), false};
  }
  // -> affine (x, y) ark form; returns false for identity
  static bool to_affine(const P& p, H4& x, H4& y) {
    if (p.inf || F::is_zero(p.zz)) {
      x = F::zero();
      y = F::zero();
      return false;
    }
    H4 i = F::inv(F::mul(p.zz, p.zzz));
    x = F::mul(p.x, F::mul(i, p.zzz));
    y = F::mul(p.y, F::mul(i, p.zz));
    return true;
  }
  // Jacobian (X, Y, Z) ark form -> XYZZ
  static P from_jacobian(const H4& X, const H4& Y, const H4& Z) {
    if (F::is_zero(Z)) return identity();
    H4 zz = F::sqr(Z);
    return P{X, Y, zz, F::mul(zz, Z), false};
  }
  static P mul_canon(const P& p, const H4& k_canon) {
    P acc = identity();
    for (int i = 3; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        acc = dbl(acc);
        if ((k_canon.v[i] >> b) & 1) acc = add(acc, p);
      }
    return acc;
  }
};

}  // namespace host
}  // namespace bpgpu
