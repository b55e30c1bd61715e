// Host build of the device arithmetic headers with overflow-checked accumulators
// (BP_CHECKED).  Test-only: lets `pytest -m "not gpu"` exercise field.cuh / curve.cuh
// bit-for-bit on the CPU and prove the lazy-limb bounds.  Not part of the product path.
#include <cstdio>
#include <cstring>

#include "curve.cuh"

namespace bpgpu {
static long g_fail = 0;
void bp_check_fail(const char* what) {
  if (g_fail < 5) fprintf(stderr, "[bp_check] %s\n", what);
  g_fail++;
}
}  // namespace bpgpu

using namespace bpgpu;

template <int FID>
static void load8(const uint32_t* p, uint32_t (&w)[8]) {
  for (int i = 0; i < 8; i++) w[i] = p[i];
}

template <int FID>
static int fe_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  using F = Fp<FID>;
  uint32_t wa[8], wb[8], wo[8];
  load8<FID>(a, wa);
  load8<FID>(b, wb);
  Fe x = F::load_ark(wa), y = F::load_ark(wb), r;
  switch (op) {
    case 0: r = F::mul(x, y); break;
    case 1: r = F::sqr(x); break;
    case 2: r = F::add(x, y); break;
    case 3: r = F::sub(x, y); break;
    case 4: r = F::inv(x); break;
    case 5: r = F::mul(F::add(x, y), F::sub(x, y)); break;           // lazy operands
    case 6: r = F::sqr(F::sub(x, y)); break;
    case 7: r = F::mul(F::sub(y, x), F::sub(x, y)); break;
    case 8: r = F::neg(x); break;
    case 9: {  // a long lazy chain, stays < 100p
      r = x;
      for (int i = 0; i < 20; i++) r = F::carry(F::add(r, F::add(x, y)));
      r = F::mul(r, y);
      break;
    }
    case 10: r = F::is_zero(F::sub(x, y)) ? F::one() : F::zero(); break;
    case 11: r = F::inv_fermat(x); break;
    case 12: r = F::inv(F::sub(F::add(x, x), x)); break;  // lazy (uncarried) operand
    case 13: {  // a*a' + b*b' with one reduction, operands as the scalar fold has them: unpacked words and products
      uint32_t pa[8], pb[8];
      F::pack(F::canon(x), pa);
      F::pack(F::canon(y), pb);
      r = F::mul_add2(F::unpack(pa), F::mul(x, y), F::unpack(pb), F::sqr(y));
      break;
    }
    case 14: {  // the scalar fold's output path: packed words times canonical multipliers, conditional-add canonicalisation
      uint32_t pa[8], pb[8];
      F::pack(F::canon(x), pa);
      F::pack(F::canon(y), pb);
      r = F::canon_small(F::mul_add2(F::unpack(pa), F::canon(F::mul(x, y)), F::unpack(pb), F::canon(F::sqr(y))));
      Fe full = F::canon(r);
      for (int i = 0; i < 9; i++)
        if (full.l[i] != r.l[i]) bp_check_fail("canon_small is not canonical");
      break;
    }
    case 15: {  // ark form -> canonical integer as the digit kernel does it (output: that integer, as ark words of it / 2^256)
      Fe v = F::canon_small(F::mul(F::unpack(wa), F::from_const(F::C::C256TOCANON)));
      Fe full = F::canon(v);
      for (int i = 0; i < 9; i++)
        if (full.l[i] != v.l[i]) bp_check_fail("canon_small is not canonical");
      F::pack(v, wo);
      for (int i = 0; i < 8; i++) out[i] = wo[i];
      return 0;
    }
    default: return -1;
  }
  F::store_ark(r, wo);
  for (int i = 0; i < 8; i++) out[i] = wo[i];
  return 0;
}

template <int FID>
static Aff load_ark_affine(const uint32_t* p) {
  using F = Fp<FID>;
  uint32_t wx[8], wy[8];
  load8<FID>(p, wx);
  load8<FID>(p + 8, wy);
  Aff r;
  r.inf = PointIo<FID>::all_zero(wx) && PointIo<FID>::all_zero(wy);
  r.x = F::load_ark(wx);
  r.y = F::load_ark(wy);
  return r;
}
template <int FID>
static void store_ark_affine(uint32_t* p, const Aff& a) {
  using F = Fp<FID>;
  uint32_t wx[8], wy[8];
  F::store_ark(a.x, wx);
  F::store_ark(a.y, wy);
  for (int i = 0; i < 8; i++) {
    p[i] = a.inf ? 0 : wx[i];
    p[8 + i] = a.inf ? 0 : wy[i];
  }
}

// mode 0: acc = sum of +-points via madd; 1: pairwise tree via add(); 2: acc = [k]P via dbl/madd
template <int FID>
static int pt_op(int mode, const uint32_t* pts, const int* signs, int n, uint64_t k, uint32_t* out) {
  using E = Ec<FID>;
  Xyzz acc = E::identity();
  if (mode == 0) {
    for (int i = 0; i < n; i++) {
      Aff q = load_ark_affine<FID>(pts + 16 * i);
      if (signs && signs[i]) q = E::neg(q);
      E::madd(acc, q);
    }
  } else if (mode == 1) {
    Xyzz* v = new Xyzz[n > 0 ? n : 1];
    for (int i = 0; i < n; i++) {
      Aff q = load_ark_affine<FID>(pts + 16 * i);
      if (signs && signs[i]) q = E::neg(q);
      v[i] = E::from_affine(q);
      // push through the packed device format once, as kernels do
      uint32_t buf[32];
      PointIo<FID>::store_dev_xyzz(buf, v[i]);
      v[i] = PointIo<FID>::load_dev_xyzz(buf);
    }
    int m = n;
    while (m > 1) {
      int h = (m + 1) / 2;
      for (int i = 0; i + h < m; i++) v[i] = E::add(v[i], v[i + h]);
      m = h;
    }
    if (n > 0) acc = v[0];
    delete[] v;
  } else if (mode == 2) {
    Aff q = load_ark_affine<FID>(pts);
    for (int b = 63; b >= 0; b--) {
      acc = E::dbl(acc);
      if ((k >> b) & 1) E::madd(acc, q);
    }
  } else if (mode == 3) {  // xyzz neg + add:  P0 - P1 + ... via add(neg)
    for (int i = 0; i < n; i++) {
      Xyzz t = E::from_affine(load_ark_affine<FID>(pts + 16 * i));
      if (signs && signs[i]) t = E::neg(t);
      acc = E::add(acc, t);
    }
  }
  if (mode == 4 || mode == 5) {  // the Jacobian ladder arithmetic: 4 = sum of +-points via jmadd, 5 = [k]P via jdbl / jmadd
    Jac j = E::jidentity();
    if (mode == 4) {
      for (int i = 0; i < n; i++) {
        Aff q = load_ark_affine<FID>(pts + 16 * i);
        if (signs && signs[i]) q = E::neg(q);
        E::jmadd(j, q);
      }
    } else {
      Aff q = load_ark_affine<FID>(pts);
      for (int b = 63; b >= 0; b--) {
        j = E::jdbl(j);
        if ((k >> b) & 1) E::jmadd(j, q);
      }
      for (int i = 0; i < 200; i++) j = E::jdbl(j);  // a long run of doublings, as the fixed-base table walks
      (void)n;
    }
    store_ark_affine<FID>(out, E::jto_affine_with(j, j.inf ? Fp<FID>::one() : Fp<FID>::inv(j.z)));
    return 0;
  }
  store_ark_affine<FID>(out, E::to_affine(acc));
  return 0;
}

extern "C" {
int bpchk_fe_op(int fid, int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  return fid == 0 ? fe_op<0>(op, a, b, out) : fe_op<1>(op, a, b, out);
}
int bpchk_pt_op(int fid, int mode, const uint32_t* pts, const int* signs, int n, uint64_t k, uint32_t* out) {
  return fid == 0 ? pt_op<0>(mode, pts, signs, n, k, out) : pt_op<1>(mode, pts, signs, n, k, out);
}
long bpchk_failures() { return g_fail; }
void bpchk_reset() { g_fail = 0; }
}
