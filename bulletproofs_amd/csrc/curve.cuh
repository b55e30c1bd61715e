// Pasta curve arithmetic (y^2 = x^3 + 5, a = 0) in extended Jacobian "XYZZ"
// coordinates:  x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2.  One coordinate system serves the
// bucket accumulation (mixed add 8M+2S), the bucket reduction / fix-ups (full add
// 12M+2S) and the shared-scalar generator fold (doubling 6M+3S).
//
// Formulas: EFD madd-2008-s, add-2008-s, dbl-2008-s-1, mdbl-2008-s-1, restated for the
// lazy signed-limb field of field.cuh (see the bound notes there); every special
// case (identity operands, P + P, P + (-P)) is handled, behind a cheap low-limb
// filter so the common path stays branch-free in practice.
//
// Reference semantics being reproduced: `buckets[..] += base`, `running_sum += &b`,
// `total.double_in_place()` in src/util/msm.rs:148,169-170,187 and
// `l.mul(&x_inv) + r.mul(&x)` in src/proofs/bp_iter.rs:99,103 -- as group operations;
// the projective representative differs from ark-ec's Jacobian one, which is why
// parity is defined on normalised (affine) elements.
#pragma once
#include "field.cuh"

namespace bpgpu {

// Affine point in device form; inf encodes the identity.
struct Aff {
  Fe x, y;
  bool inf;
};

struct Xyzz {
  Fe x, y, zz, zzz;
  bool inf;
};

// Jacobian point (x = X/Z^2, y = Y/Z^3): the coordinates of the shared-multiplier ladders.  A doubling costs
// 4S + 3M here against 3S + 4M + a two-product accumulation in XYZZ (738 against 981 multiply-adds), a mixed
// addition 3S + 6M + the accumulation against 2S + 6M + it (1233 against 1143): chains with more doublings than
// additions (the generator folds, the fixed-base table) run in Jacobian, the bucket work in XYZZ.
struct Jac {
  Fe x, y, z;
  bool inf;
};

template <int FID>
struct Ec {
  using F = Fp<FID>;

  static BP_HD Xyzz identity() {
    Xyzz r;
    r.x = F::zero();
    r.y = F::zero();
    r.zz = F::zero();
    r.zzz = F::zero();
    r.inf = true;
    return r;
  }
  static BP_HD Xyzz from_affine(const Aff& p) {
    if (p.inf) return identity();
    Xyzz r;
    r.x = p.x;
    r.y = F::carry(p.y);  // p.y may be a negated (non-positive-limb) value
    r.zz = F::one();
    r.zzz = F::one();
    r.inf = false;
    return r;
  }
  // -p: limbs of y become non-positive (|limb| < 2^29), fine as a mul operand
  static BP_HD Aff neg(const Aff& p) {
    Aff r = p;
    r.y = F::neg(p.y);
    return r;
  }
  static BP_HD Xyzz neg(const Xyzz& p) {
    Xyzz r = p;
    r.y = F::carry(F::neg(p.y));
    return r;
  }
  // XYZZ -> affine (one field inversion)
  static BP_HD Aff to_affine(const Xyzz& p) {
    Aff r;
    r.inf = p.inf;
    if (p.inf) {
      r.x = F::zero();
      r.y = F::zero();
      return r;
    }
    Fe i = F::inv(F::mul(p.zz, p.zzz));
    r.x = F::mul(p.x, F::mul(i, p.zzz));
    r.y = F::mul(p.y, F::mul(i, p.zz));
    return r;
  }

  // 2*(x, y) for an affine non-identity point (mdbl-2008-s-1); y != 0 always holds on
  // a prime-order curve.
  static BP_HD Xyzz dbl_affine(const Aff& p) {
    if (p.inf) return identity();
    Xyzz r;
    Fe U = F::carry(F::add(p.y, p.y));
    Fe V = F::sqr(U);
    Fe W = F::mul(U, V);
    Fe S = F::mul(p.x, V);
    Fe M = F::carry(F::mul_small(F::sqr(p.x), 3));
    Fe X3 = F::carry(F::sub(F::sub(F::sqr(M), S), S));
    Fe Y3 = F::mul_sub2(M, F::sub(S, X3), W, p.y);
    r.x = X3;
    r.y = Y3;
    r.zz = V;
    r.zzz = W;
    r.inf = false;
    return r;
  }

  // dbl-2008-s-1
  static BP_HD Xyzz dbl(const Xyzz& p) {
    if (p.inf) return p;
    Xyzz r;
    Fe U = F::carry(F::add(p.y, p.y));
    Fe V = F::sqr(U);
    Fe W = F::mul(U, V);
    Fe S = F::mul(p.x, V);
    Fe M = F::carry(F::mul_small(F::sqr(p.x), 3));
    Fe X3 = F::carry(F::sub(F::sub(F::sqr(M), S), S));
    Fe Y3 = F::mul_sub2(M, F::sub(S, X3), W, p.y);
    r.x = X3;
    r.y = Y3;
    r.zz = F::mul(V, p.zz);
    r.zzz = F::mul(W, p.zzz);
    r.inf = false;
    return r;
  }

  // acc += q  (q affine; madd-2008-s).  acc.x / acc.y are kept carried.
  static BP_HD void madd(Xyzz& acc, const Aff& q) {
    if (q.inf) return;
    if (acc.inf) {
      acc = from_affine(q);
      return;
    }
    Fe U2 = F::mul(q.x, acc.zz);
    Fe S2 = F::mul(q.y, acc.zzz);
    Fe P = F::sub(U2, acc.x);
    Fe R = F::sub(S2, acc.y);
    if (F::is_zero(P)) {
      if (F::is_zero(R)) {
        acc = dbl_affine(q);
      } else {
        acc = identity();
      }
      return;
    }
    Fe PP = F::sqr(P);
    Fe PPP = F::mul(P, PP);
    Fe Q = F::mul(acc.x, PP);
    Fe X3 = F::carry(F::sub(F::sub(F::sub(F::sqr(R), PPP), Q), Q));
    Fe Y3 = F::mul_sub2(R, F::sub(Q, X3), acc.y, PPP);
    acc.x = X3;
    acc.y = Y3;
    acc.zz = F::mul(acc.zz, PP);
    acc.zzz = F::mul(acc.zzz, PPP);
  }

  // ---- Jacobian ladder arithmetic -------------------------------------------------------------
  static BP_HD Jac jidentity() {
    Jac r;
    r.x = F::zero();
    r.y = F::zero();
    r.z = F::zero();
    r.inf = true;
    return r;
  }
  static BP_HD Jac jfrom_affine(const Aff& p) {
    if (p.inf) return jidentity();
    Jac r;
    r.x = p.x;
    r.y = F::carry(p.y);
    r.z = F::one();
    r.inf = false;
    return r;
  }
  // 2 p (a = 0; dbl-2009-l with X B in place of its squaring trick): A = X^2, B = Y^2, C = B^2, D = 4 X B,
  // E = 3 A, X3 = E^2 - 2 D, Y3 = E (D - X3) - 8 C, Z3 = 2 Y Z.  y != 0 on a prime-order curve.
  static BP_HD Jac jdbl(const Jac& p) {
    if (p.inf) return p;
    Jac r;
    const Fe A = F::sqr(p.x), B = F::sqr(p.y), C = F::sqr(B);
    const Fe D = F::carry(F::mul_small(F::mul(p.x, B), 4));
    const Fe E = F::carry(F::mul_small(A, 3));
    r.x = F::carry(F::sub(F::sqr(E), F::add(D, D)));
    const Fe C4 = F::carry(F::mul_small(C, 4));
    r.y = F::carry(F::sub(F::mul(E, F::sub(D, r.x)), F::add(C4, C4)));
    r.z = F::mul(F::add(p.y, p.y), p.z);
    r.inf = false;
    return r;
  }
  // acc += q (q affine): ZZ = Z^2, U2 = x ZZ, S2 = y Z ZZ, H = U2 - X, R = S2 - Y, X3 = R^2 - H^3 - 2 X H^2,
  // Y3 = R (X H^2 - X3) - Y H^3, Z3 = Z H
  static BP_HD void jmadd(Jac& acc, const Aff& q) {
    if (q.inf) return;
    if (acc.inf) {
      acc = jfrom_affine(q);
      return;
    }
    const Fe ZZ = F::sqr(acc.z);
    const Fe U2 = F::mul(q.x, ZZ);
    const Fe S2 = F::mul(q.y, F::mul(acc.z, ZZ));
    const Fe H = F::sub(U2, acc.x);
    const Fe R = F::sub(S2, acc.y);
    if (F::is_zero(H)) {
      if (F::is_zero(R)) {
        acc = jdbl(jfrom_affine(q));
      } else {
        acc = jidentity();
      }
      return;
    }
    const Fe HH = F::sqr(H);
    const Fe HHH = F::mul(H, HH);
    const Fe V = F::mul(acc.x, HH);
    const Fe X3 = F::carry(F::sub(F::sub(F::sub(F::sqr(R), HHH), V), V));
    const Fe Y3 = F::mul_sub2(R, F::sub(V, X3), acc.y, HHH);
    acc.z = F::mul(acc.z, H);
    acc.x = X3;
    acc.y = Y3;
  }
  // Jacobian -> affine given zi = 1 / Z
  static BP_HD Aff jto_affine_with(const Jac& p, const Fe& zi) {
    Aff r;
    r.inf = p.inf;
    if (p.inf) {
      r.x = F::zero();
      r.y = F::zero();
      return r;
    }
    const Fe zi2 = F::sqr(zi);
    r.x = F::mul(p.x, zi2);
    r.y = F::mul(p.y, F::mul(zi2, zi));
    return r;
  }

  // a + b, both XYZZ (add-2008-s)
  static BP_HD Xyzz add(const Xyzz& a, const Xyzz& b) {
    if (a.inf) return b;
    if (b.inf) return a;
    Fe U1 = F::mul(a.x, b.zz);
    Fe U2 = F::mul(b.x, a.zz);
    Fe S1 = F::mul(a.y, b.zzz);
    Fe S2 = F::mul(b.y, a.zzz);
    Fe P = F::sub(U2, U1);
    Fe R = F::sub(S2, S1);
    if (F::is_zero(P)) {
      if (F::is_zero(R)) return dbl(a);
      return identity();
    }
    Xyzz r;
    Fe PP = F::sqr(P);
    Fe PPP = F::mul(P, PP);
    Fe Q = F::mul(U1, PP);
    Fe X3 = F::carry(F::sub(F::sub(F::sub(F::sqr(R), PPP), Q), Q));
    Fe Y3 = F::mul_sub2(R, F::sub(Q, X3), S1, PPP);
    r.x = X3;
    r.y = Y3;
    r.zz = F::mul(F::mul(a.zz, b.zz), PP);
    r.zzz = F::mul(F::mul(a.zzz, b.zzz), PPP);
    r.inf = false;
    return r;
  }
};

// ---------------------------------------------------------------------------
// Memory formats (all little-endian u32 words)
//   ark affine   : x[8] y[8]           canonical, value*2^256  (0,0) = identity
//   ark jacobian : X[8] Y[8] Z[8]      canonical, value*2^256  Z = 0  = identity
//   dev affine   : x[8] y[8]           canonical, value*2^261  (0,0) = identity
//   dev xyzz     : X[8] Y[8] ZZ[8] ZZZ[8]  canonical, value*2^261, ZZ = 0 = identity
// ---------------------------------------------------------------------------
template <int FID>
struct PointIo {
  using F = Fp<FID>;

  static BP_HD bool all_zero(const uint32_t (&w)[8]) {
    uint32_t a = 0;
    BP_UNROLL
    for (int i = 0; i < 8; i++) a |= w[i];
    return a == 0;
  }

  static BP_HD Aff load_dev_affine(const uint32_t* p) {
    uint32_t wx[8], wy[8];
    BP_UNROLL
    for (int i = 0; i < 8; i++) {
      wx[i] = p[i];
      wy[i] = p[8 + i];
    }
    Aff r;
    r.inf = all_zero(wx) && all_zero(wy);
    r.x = F::unpack(wx);
    r.y = F::unpack(wy);
    return r;
  }
  static BP_HD void store_dev_affine(uint32_t* p, const Aff& a) {
    uint32_t wx[8], wy[8];
    F::store_packed(a.x, wx);
    F::store_packed(a.y, wy);
    BP_UNROLL
    for (int i = 0; i < 8; i++) {
      p[i] = a.inf ? 0u : wx[i];
      p[8 + i] = a.inf ? 0u : wy[i];
    }
  }
  static BP_HD Xyzz load_dev_xyzz(const uint32_t* p) {
    uint32_t w[4][8];
    BP_UNROLL
    for (int c = 0; c < 4; c++) {
      BP_UNROLL
      for (int i = 0; i < 8; i++) w[c][i] = p[8 * c + i];
    }
    Xyzz r;
    r.inf = all_zero(w[2]);
    r.x = F::unpack(w[0]);
    r.y = F::unpack(w[1]);
    r.zz = F::unpack(w[2]);
    r.zzz = F::unpack(w[3]);
    return r;
  }
  static BP_HD void store_dev_xyzz(uint32_t* p, const Xyzz& a) {
    uint32_t w[4][8];
    F::store_packed(a.x, w[0]);
    F::store_packed(a.y, w[1]);
    F::store_packed(a.zz, w[2]);
    F::store_packed(a.zzz, w[3]);
    BP_UNROLL
    for (int c = 0; c < 4; c++) {
      BP_UNROLL
      for (int i = 0; i < 8; i++) p[8 * c + i] = a.inf ? 0u : w[c][i];
    }
  }
  // dev jacobian: X[8] Y[8] Z[8], canonical, value*2^261, Z = 0 = identity (staging of the fixed-base table)
  static BP_HD Jac load_dev_jac(const uint32_t* p) {
    uint32_t w[3][8];
    BP_UNROLL
    for (int c = 0; c < 3; c++) {
      BP_UNROLL
      for (int i = 0; i < 8; i++) w[c][i] = p[8 * c + i];
    }
    Jac r;
    r.inf = all_zero(w[2]);
    r.x = F::unpack(w[0]);
    r.y = F::unpack(w[1]);
    r.z = F::unpack(w[2]);
    return r;
  }
  static BP_HD void store_dev_jac(uint32_t* p, const Jac& a) {
    uint32_t w[3][8];
    F::store_packed(a.x, w[0]);
    F::store_packed(a.y, w[1]);
    F::store_packed(a.z, w[2]);
    BP_UNROLL
    for (int c = 0; c < 3; c++) {
      BP_UNROLL
      for (int i = 0; i < 8; i++) p[8 * c + i] = a.inf ? 0u : w[c][i];
    }
  }
  // XYZZ in ark form (value*2^256) for the host tail
  static BP_HD void store_ark_xyzz(uint32_t* p, const Xyzz& a) {
    uint32_t w[4][8];
    F::store_ark(a.x, w[0]);
    F::store_ark(a.y, w[1]);
    F::store_ark(a.zz, w[2]);
    F::store_ark(a.zzz, w[3]);
    BP_UNROLL
    for (int c = 0; c < 4; c++) {
      BP_UNROLL
      for (int i = 0; i < 8; i++) p[8 * c + i] = a.inf ? 0u : w[c][i];
    }
  }
};

}  // namespace bpgpu
