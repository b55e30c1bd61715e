// 255-bit prime-field arithmetic for the two Pasta fields on gfx950.
//
// Representation ("Fe"): nine signed 32-bit limbs in radix 2^29,
//     value = sum_k l[k] * 2^(29k),
// held in Montgomery form with R' = 2^261.  On MI355X v_mad_i64_i32 /
// v_mad_u64_u32 issue at the same rate as a 32-bit add (measured:
// tools/ubench_valu.hip, ~35 T lane-ops/s), so the cheapest multiplication is the
// one with the fewest *instructions*: with 29-bit limbs every 64-bit column
// accumulator absorbs its <= 9 products and the interleaved Montgomery reduction
// without a single carry instruction.  Limbs are signed so that subtraction is nine
// v_sub's with no borrow chain; values float in (-100p, 100p) between reductions.
//
// Invariants (checked by tests/test_field_host.py through the host build of this
// header, which replaces int64 accumulators with overflow-checked __int128):
//   * "carried" form: limbs 0..7 in [0, 2^29), limb 8 signed.  mul/sqr return it.
//   * mul(a, b) needs  max|a.l[i]| * max|b.l[j]| <= 2^59.5  (i, j <= 7), which holds
//     when each operand is carried, or a sum/difference of two carried values and
//     the other operand is carried or a difference of two carried values.
//   * |value| < 100p everywhere (top limb stays far inside int32).
//
// Both moduli are p = 2^254 + t with t < 2^125 and p == 1 (mod 2^32); in radix 2^29
// p has limbs {1, p1, p2, p3, p4, 0, 0, 0, 2^22}, so one reduction step costs four
// multiply-adds and a shift instead of nine.
//
// The same header compiles for the host (g++) so the arithmetic is unit-tested on
// CPU; the product path only ever runs it on the GPU.
#pragma once
#include <cstdint>

#include "pasta_consts.h"

#if defined(__HIPCC__)
#define BP_HD __host__ __device__ __forceinline__
#define BP_UNROLL _Pragma("unroll")
#if defined(BP_NOINLINE_MUL)
#define BP_MUL __host__ __device__ __attribute__((noinline))
#else
#define BP_MUL BP_HD
#endif
#else
#define BP_HD inline
#define BP_MUL inline
#define BP_UNROLL
#endif

namespace bpgpu {

constexpr uint32_t MASK29 = (1u << 29) - 1u;

struct Fe {
  int32_t l[9];
};

#if defined(BP_CHECKED)
// Host-only checked build: accumulators are __int128 and every step asserts that
// the value would have fitted the int64 the device uses.
using acc_t = __int128;
void bp_check_fail(const char* what);
inline void bp_check_acc(acc_t v) {
  if (v > (acc_t)INT64_MAX || v < (acc_t)INT64_MIN) bp_check_fail("int64 accumulator overflow");
}
inline void bp_check_limb(int64_t v) {
  if (v > INT32_MAX || v < INT32_MIN) bp_check_fail("int32 limb overflow");
}
#else
using acc_t = int64_t;
#define bp_check_acc(v) ((void)0)
#define bp_check_limb(v) ((void)0)
#endif

template <int FID>
struct Fp {
  using C = FieldConsts<FID>;

  static BP_HD Fe from_const(const uint32_t (&c)[9]) {
    Fe r;
    BP_UNROLL
    for (int i = 0; i < 9; i++) r.l[i] = (int32_t)c[i];
    return r;
  }
  static BP_HD Fe zero() {
    Fe r;
    BP_UNROLL
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
  }
  static BP_HD Fe one() { return from_const(C::ONE261); }

  static BP_HD Fe add(const Fe& a, const Fe& b) {
    Fe r;
    BP_UNROLL
    for (int i = 0; i < 9; i++) {
      bp_check_limb((int64_t)a.l[i] + b.l[i]);
      r.l[i] = a.l[i] + b.l[i];
    }
    return r;
  }
  static BP_HD Fe sub(const Fe& a, const Fe& b) {
    Fe r;
    BP_UNROLL
    for (int i = 0; i < 9; i++) {
      bp_check_limb((int64_t)a.l[i] - b.l[i]);
      r.l[i] = a.l[i] - b.l[i];
    }
    return r;
  }
  static BP_HD Fe neg(const Fe& a) {
    Fe r;
    BP_UNROLL
    for (int i = 0; i < 9; i++) r.l[i] = -a.l[i];
    return r;
  }
  // limb-wise multiple by a small constant (caller owns the limb bound)
  static BP_HD Fe mul_small(const Fe& a, int32_t k) {
    Fe r;
    BP_UNROLL
    for (int i = 0; i < 9; i++) {
      bp_check_limb((int64_t)a.l[i] * k);
      r.l[i] = a.l[i] * k;
    }
    return r;
  }
  // carry propagation: limbs 0..7 -> [0, 2^29), limb 8 keeps the signed remainder
  static BP_HD Fe carry(const Fe& a) {
    Fe r = a;
    BP_UNROLL
    for (int i = 0; i < 8; i++) {
      int32_t c = r.l[i] >> 29;
      r.l[i] &= (int32_t)MASK29;
      bp_check_limb((int64_t)r.l[i + 1] + c);
      r.l[i + 1] += c;
    }
    return r;
  }

  // Montgomery reduction of 17 product columns t[0..16] (t[17] = 0 on entry).
  // p == 1 (mod 2^29), so column k is cleared by SUBTRACTING m p 2^(29k) with m = t[k] mod 2^29
  // (the plain low bits: one v_and).  t[k] - m is a multiple of 2^29 and (t[k] - m) >> 29 ==
  // t[k] >> 29 (arithmetic shift), so the cleared column itself is never written: the carry is a
  // shift and an add.  The result lies in (T/2^261 - p, T/2^261]: it may be negative, which the
  // signed top limb carries.
  static BP_HD Fe redc(acc_t (&t)[18]) {
#if defined(__HIP_DEVICE_COMPILE__)
    // keep 2^22 opaque so that m * 2^22 stays one multiply-add instead of a 64-bit shift and a
    // 64-bit subtract (three instructions)
    int32_t n8;
    asm("s_mov_b32 %0, 0xffc00000" : "=s"(n8));  // -2^22
#else
    const int32_t n8 = -(int32_t)C::P29[8];
#endif
    const int64_t n1 = -(int64_t)C::P29[1], n2 = -(int64_t)C::P29[2], n3 = -(int64_t)C::P29[3], n4 = -(int64_t)C::P29[4];
    BP_UNROLL
    for (int k = 0; k < 9; k++) {
      const int64_t m = (int64_t)((uint32_t)t[k] & MASK29);
      t[k + 1] += (acc_t)(m * n1);
      t[k + 2] += (acc_t)(m * n2);
      t[k + 3] += (acc_t)(m * n3);
      t[k + 4] += (acc_t)(m * n4);
      t[k + 8] += (acc_t)(m * (int64_t)n8);
      bp_check_acc(t[k + 1]);
      bp_check_acc(t[k + 2]);
      bp_check_acc(t[k + 3]);
      bp_check_acc(t[k + 4]);
      bp_check_acc(t[k + 8]);
      t[k + 1] += t[k] >> 29;
      bp_check_acc(t[k + 1]);
    }
    Fe r;
    BP_UNROLL
    for (int k = 9; k < 17; k++) {
      r.l[k - 9] = (int32_t)((uint32_t)t[k] & MASK29);
      t[k + 1] += t[k] >> 29;
      bp_check_acc(t[k + 1]);
    }
    bp_check_limb((int64_t)t[17]);
    r.l[8] = (int32_t)t[17];
    return r;
  }

  static BP_MUL Fe mul(const Fe& a, const Fe& b) {
    acc_t t[18];
    BP_UNROLL
    for (int k = 0; k < 18; k++) t[k] = 0;
    BP_UNROLL
    for (int i = 0; i < 9; i++) {
      BP_UNROLL
      for (int j = 0; j < 9; j++) {
        t[i + j] += (acc_t)((int64_t)a.l[i] * (int64_t)b.l[j]);
        bp_check_acc(t[i + j]);
      }
    }
    return redc(t);
  }

  // a*b - c*d with one Montgomery reduction (the two products share the column accumulators).
  // Needs 9 * (max|a| max|b| + max|c| max|d|) < 2^62.4: every operand carried or a difference of
  // two carried values.
  static BP_MUL Fe mul_sub2(const Fe& a, const Fe& b, const Fe& c, const Fe& d) {
    acc_t t[18];
    BP_UNROLL
    for (int k = 0; k < 18; k++) t[k] = 0;
    BP_UNROLL
    for (int i = 0; i < 9; i++) {
      BP_UNROLL
      for (int j = 0; j < 9; j++) {
        t[i + j] += (acc_t)((int64_t)a.l[i] * (int64_t)b.l[j]);
        bp_check_acc(t[i + j]);
      }
    }
    BP_UNROLL
    for (int i = 0; i < 9; i++) {
      const int32_t nc = -c.l[i];
      BP_UNROLL
      for (int j = 0; j < 9; j++) {
        t[i + j] += (acc_t)((int64_t)nc * (int64_t)d.l[j]);
        bp_check_acc(t[i + j]);
      }
    }
    return redc(t);
  }

  // a*b + c*d with one Montgomery reduction; same operand bounds as mul_sub2
  static BP_MUL Fe mul_add2(const Fe& a, const Fe& b, const Fe& c, const Fe& d) {
    acc_t t[18];
    BP_UNROLL
    for (int k = 0; k < 18; k++) t[k] = 0;
    BP_UNROLL
    for (int i = 0; i < 9; i++) {
      BP_UNROLL
      for (int j = 0; j < 9; j++) {
        t[i + j] += (acc_t)((int64_t)a.l[i] * (int64_t)b.l[j]);
        bp_check_acc(t[i + j]);
      }
    }
    BP_UNROLL
    for (int i = 0; i < 9; i++) {
      BP_UNROLL
      for (int j = 0; j < 9; j++) {
        t[i + j] += (acc_t)((int64_t)c.l[i] * (int64_t)d.l[j]);
        bp_check_acc(t[i + j]);
      }
    }
    return redc(t);
  }

  static BP_MUL Fe sqr(const Fe& a) {
    acc_t t[18];
    BP_UNROLL
    for (int k = 0; k < 18; k++) t[k] = 0;
    BP_UNROLL
    for (int i = 0; i < 9; i++) {
      t[2 * i] += (acc_t)((int64_t)a.l[i] * (int64_t)a.l[i]);
      bp_check_acc(t[2 * i]);
      bp_check_limb(2 * (int64_t)a.l[i]);
      int32_t a2 = 2 * a.l[i];
      BP_UNROLL
      for (int j = i + 1; j < 9; j++) {
        t[i + j] += (acc_t)((int64_t)a2 * (int64_t)a.l[j]);
        bp_check_acc(t[i + j]);
      }
    }
    return redc(t);
  }

  // w in [0, 3p) carried with non-negative top limb  ->  w - p if w >= p
  static BP_HD void sub_p_if_ge(Fe& w) {
    Fe d;
    int32_t c = 0;
    BP_UNROLL
    for (int i = 0; i < 8; i++) {
      int32_t s = w.l[i] - (int32_t)C::P29[i] + c;
      d.l[i] = s & (int32_t)MASK29;
      c = s >> 29;
    }
    d.l[8] = w.l[8] - (int32_t)C::P29[8] + c;
    bool ge = d.l[8] >= 0;
    BP_UNROLL
    for (int i = 0; i < 9; i++) w.l[i] = ge ? d.l[i] : w.l[i];
  }

  // Full reduction to the canonical representative in [0, p), carried limbs.
  static BP_HD Fe canon(const Fe& a) {
    Fe r = carry(a);
    // v = top*2^232 + low, low in [0, 2^232): q = floor(v / 2^254)
    int32_t k = (r.l[8] >> 22) - 1;  // subtract (q-1)*p so the result lands in (0, 3p)
    int64_t c = 0;
    BP_UNROLL
    for (int i = 0; i < 9; i++) {
      int64_t s = (int64_t)r.l[i] - (int64_t)k * (int64_t)C::P29[i] + c;
      if (i < 8) {
        r.l[i] = (int32_t)((uint32_t)s & MASK29);
        c = s >> 29;
      } else {
        r.l[i] = (int32_t)s;
      }
    }
    sub_p_if_ge(r);
    sub_p_if_ge(r);
    return r;
  }

  // canonical representative of a value known to lie in (-p, p): one conditional addition of p
  // (a product of a < 2^256 and a canonical b leaves redc in (-p, p/32]; sums of two such in (-p, p/16])
  static BP_HD Fe canon_small(const Fe& a) {
    Fe r = carry(a);
    const int32_t neg = r.l[8] >> 31;  // all ones iff negative
    BP_UNROLL
    for (int i = 0; i < 9; i++)
      if (C::P29[i] != 0) r.l[i] += (int32_t)C::P29[i] & neg;
    return carry(r);
  }

  // exact test  value == 0 (mod p)  for |value| < 128p
  static BP_HD bool is_zero(const Fe& a) {
    // k*p mod 2^29 = k mod 2^29 with |k| <= 128: cheap filter on the low limb
    uint32_t t = ((uint32_t)a.l[0] + 128u) & MASK29;
    if (t > 256u) return false;
    Fe r = canon(a);
    int32_t acc = 0;
    BP_UNROLL
    for (int i = 0; i < 9; i++) acc |= r.l[i];
    return acc == 0;
  }

  // canonical packed words (8 x u32, value < 2^256) -> carried limbs
  static BP_HD Fe unpack(const uint32_t (&w)[8]) {
    Fe r;
    BP_UNROLL
    for (int k = 0; k < 9; k++) {
      const int o = 29 * k, wi = o >> 5, sh = o & 31;
      uint32_t v = w[wi] >> sh;
      if (sh > 3 && wi + 1 < 8) v |= w[wi + 1] << (32 - sh);
      r.l[k] = (int32_t)(k < 8 ? (v & MASK29) : v);
    }
    return r;
  }
  // carried, non-negative limbs (e.g. canon output) -> packed words
  static BP_HD void pack(const Fe& a, uint32_t (&w)[8]) {
    BP_UNROLL
    for (int j = 0; j < 8; j++) {
      const int o = 32 * j, k = o / 29, sh = o % 29;
      uint32_t v = (uint32_t)a.l[k] >> sh;
      if (k + 1 < 9) v |= (uint32_t)a.l[k + 1] << (29 - sh);
      if (58 - sh < 32 && k + 2 < 9) v |= (uint32_t)a.l[k + 2] << (58 - sh);
      w[j] = v;
    }
  }

  // ---- boundary conversions (ark-ff memory form is value*2^256 mod p, canonical) ----
  static BP_HD Fe load_ark(const uint32_t (&w)[8]) {  // x*2^256 -> device form x*2^261
    return mul(unpack(w), from_const(C::C256TO261));
  }
  static BP_HD void store_ark(const Fe& a, uint32_t (&w)[8]) {  // x*2^261 -> x*2^256
    pack(canon(mul(a, from_const(C::C261TO256))), w);
  }
  static BP_HD void store_packed(const Fe& a, uint32_t (&w)[8]) {  // keep device form
    pack(canon(a), w);
  }

  // a^(p-2) by square-and-multiply over the bits of p-2 (p-2 = p - 2, low word
  // 0xffffffff, second word P32[1]-1 since p == 1 mod 2^32).  About 385 multiplications; kept as the
  // cross-check of inv() below (tests) -- the kernels call inv().
  static BP_HD Fe inv_fermat(const Fe& a) {
    Fe base = carry(a);
    Fe r = one();
    bool started = false;
    for (int wi = 7; wi >= 0; wi--) {
      uint32_t word = wi == 0 ? 0xffffffffu : (wi == 1 ? C::P32[1] - 1u : C::P32[wi]);
      for (int b = 31; b >= 0; b--) {
        if (started) r = sqr(r);
        if ((word >> b) & 1u) {
          r = started ? mul(r, base) : base;
          started = true;
        }
      }
    }
    return r;
  }

  // Modular inverse by Bernstein-Yang division steps ("safegcd"), about a sixth of the instructions of the
  // Fermat ladder above and, like it, branch-free (every lane of a wave runs the same schedule).
  //   divstep(delta, f, g), f odd:  delta > 0 and g odd -> (1 - delta, g, (g - f) / 2)
  //                                 otherwise           -> (1 + delta, f, (g + (g mod 2) f) / 2)
  // from (1, p, a) reaches g = 0, f = +-1 within 741 steps for 256-bit inputs (Bernstein-Yang 2019, Thm 11.2);
  // 26 batches of 29 steps = 754.  A batch only looks at the low 29 bits of f and g and yields an integer
  // matrix T (entries up to 2^29) with (f, g) <- T (f, g) / 2^29; the same T is applied to (d, e), kept
  // modulo p with d a = f, e a = g (mod p): the division by 2^29 there adds the multiple of p that clears the
  // low limb, which is the low limb itself negated because p == 1 (mod 2^29).  All big values are nine
  // signed 29-bit limbs (limbs 0..7 in [0, 2^29), limb 8 signed) -- the "carried" form of this header.
  static BP_HD Fe inv(const Fe& a) {
    const Fe g0 = canon(a);
    int32_t f[9], g[9], d[9], e[9];
    BP_UNROLL
    for (int i = 0; i < 9; i++) {
      f[i] = (int32_t)C::P29[i];
      g[i] = g0.l[i];
      d[i] = 0;
      e[i] = 0;
    }
    e[0] = 1;
    int32_t delta = 1;
    for (int it = 0; it < 26; it++) {
      // ---- 29 division steps on the low bits; T = [[u, v], [q, r]] scaled by 2^29
      uint32_t fl = (uint32_t)f[0] | ((uint32_t)f[1] << 29), gl = (uint32_t)g[0] | ((uint32_t)g[1] << 29);
      int32_t u = 1, v = 0, q = 0, r = 0;
      r = 1;
      for (int i = 0; i < 29; i++) {
        const int32_t dpos = (-delta) >> 31;          // -1 iff delta > 0
        const int32_t godd = -(int32_t)(gl & 1u);     // -1 iff g odd
        // g += +-f (minus when delta > 0), likewise its matrix row, if g is odd
        const uint32_t x = (fl ^ (uint32_t)dpos) - (uint32_t)dpos;
        const int32_t y = (u ^ dpos) - dpos, z = (v ^ dpos) - dpos;
        gl += x & (uint32_t)godd;
        q += y & godd;
        r += z & godd;
        // swap (f takes the old g) when delta > 0 and g was odd
        const int32_t sw = dpos & godd;
        delta = 1 + ((delta ^ sw) - sw);
        fl += gl & (uint32_t)sw;
        u += q & sw;
        v += r & sw;
        gl >>= 1;
        u <<= 1;
        v <<= 1;
      }
      // ---- (f, g) <- T (f, g) / 2^29 (exact)
      {
        acc_t cf = (acc_t)((int64_t)u * f[0]) + (acc_t)((int64_t)v * g[0]);
        acc_t cg = (acc_t)((int64_t)q * f[0]) + (acc_t)((int64_t)r * g[0]);
        cf >>= 29;
        cg >>= 29;
        BP_UNROLL
        for (int i = 1; i < 9; i++) {
          cf += (acc_t)((int64_t)u * f[i]) + (acc_t)((int64_t)v * g[i]);
          cg += (acc_t)((int64_t)q * f[i]) + (acc_t)((int64_t)r * g[i]);
          bp_check_acc(cf);
          bp_check_acc(cg);
          f[i - 1] = (int32_t)((uint32_t)cf & MASK29);
          g[i - 1] = (int32_t)((uint32_t)cg & MASK29);
          cf >>= 29;
          cg >>= 29;
        }
        bp_check_limb((int64_t)cf);
        bp_check_limb((int64_t)cg);
        f[8] = (int32_t)cf;
        g[8] = (int32_t)cg;
      }
      // ---- (d, e) <- T (d, e) / 2^29 (mod p); d, e stay in (-2p, p)
      {
        const int32_t sd = d[8] >> 31, se = e[8] >> 31;  // -1 iff negative
        int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
        acc_t cd = (acc_t)((int64_t)u * d[0]) + (acc_t)((int64_t)v * e[0]);
        acc_t ce = (acc_t)((int64_t)q * d[0]) + (acc_t)((int64_t)r * e[0]);
        // the multiple of p that makes the sum divisible by 2^29: p == 1 (mod 2^29), so it is minus the low limb
        md -= (int32_t)(((uint32_t)cd + (uint32_t)md) & MASK29);
        me -= (int32_t)(((uint32_t)ce + (uint32_t)me) & MASK29);
        cd += (acc_t)md;  // p limb 0 = 1
        ce += (acc_t)me;
        cd >>= 29;
        ce >>= 29;
        BP_UNROLL
        for (int i = 1; i < 9; i++) {
          cd += (acc_t)((int64_t)u * d[i]) + (acc_t)((int64_t)v * e[i]);
          ce += (acc_t)((int64_t)q * d[i]) + (acc_t)((int64_t)r * e[i]);
          if (C::P29[i] != 0) {
            cd += (acc_t)((int64_t)md * (int64_t)C::P29[i]);
            ce += (acc_t)((int64_t)me * (int64_t)C::P29[i]);
          }
          bp_check_acc(cd);
          bp_check_acc(ce);
          d[i - 1] = (int32_t)((uint32_t)cd & MASK29);
          e[i - 1] = (int32_t)((uint32_t)ce & MASK29);
          cd >>= 29;
          ce >>= 29;
        }
        // d, e in (-2p, p): top limbs within (-2^24, 2^23]
        bp_check_limb((int64_t)cd * 64);
        bp_check_limb((int64_t)ce * 64);
        d[8] = (int32_t)cd;
        e[8] = (int32_t)ce;
      }
    }
    // g = 0 and f = +-1 now (f = +-p, d = 0 when a = 0): the integer inverse is sign(f) d.  The value handed in
    // is x 2^261, its integer inverse x^-1 2^-261; one multiplication by 2^783 gives x^-1 2^261.
    const int32_t fneg = f[8] >> 31;
    Fe res;
    BP_UNROLL
    for (int i = 0; i < 9; i++) res.l[i] = (d[i] ^ fneg) - fneg;
    return mul(res, from_const(C::R3_261));
  }
};

}  // namespace bpgpu
