//! `src/gpu.rs` of ark-bp: dispatch of the MSM + IPA-folding hot path to libbpgpu (MI355X, gfx950).
//!
//! Add to the reference crate:
//!   * this file as `src/gpu.rs`, `pub mod gpu;` in `src/lib.rs` (after `pub mod util;`, src/lib.rs:12);
//!   * `bpgpu-sys = { path = "<this repository>/rust/bpgpu-sys", optional = true }` and a feature
//!     `gpu = ["bpgpu-sys"]` in `Cargo.toml` (dependencies at Cargo.toml:9-32, features at :46-50);
//!   * the three patches of `rust/patches/` (src/util/msm.rs, src/proofs/bp_iter.rs, src/reductions/bp_2ary_step.rs).
//! Traits (`Relation`, `Reduction`, `Proof`, src/lib.rs:33-101) and every public signature stay as they are.
//!
//! Every function here returns `None` when `G` is not Pallas / Vesta (BLS12-381, JubJub: the reference's own tests,
//! src/lib.rs:185-212) or the feature is off; the caller then runs its existing CPU body.
//!
//! Written against arkworks 0.3 (ark-ff / ark-ec / ark-pallas / ark-vesta 0.3.0, the versions of Cargo.lock); NOT
//! compiled in this repository's build image, which has no Rust toolchain.  The same call sequences are exercised through
//! the C++ mirror (`bulletproofs_amd/host/ark_bp.hpp`) and ctypes (`bulletproofs_amd/ffi.py`) by the test suite.
//!
//! Representation notes.  `Fp256<P>` is `(BigInteger256([u64; 4]), PhantomData)` holding the Montgomery form, which is
//! exactly the ABI's scalar / coordinate format; `GroupProjective<P>` is `{x, y, z}` (Jacobian).  Neither is `repr(C)`,
//! so values are copied limb by limb into contiguous `Vec<u64>` buffers -- never a transmute of a slice.
//! Results cross the boundary normalised (`Z = 1`): the transcript then absorbs `(x, y, 1)`, see INTEGRATION.md section 4.
#![cfg(feature = "gpu")]

use ark_ec::group::Group;
use ark_ec::short_weierstrass_jacobian::GroupProjective;
use ark_ec::SWModelParameters;
use ark_ff::{BigInteger256, Field, Fp256, Fp256Parameters, One, UniformRand, Zero};
use bpgpu_sys as sys;
use std::any::Any;
use std::sync::Once;

use crate::FiatShamirRng;

static INIT: Once = Once::new();
fn init() {
    INIT.call_once(|| {
        let dev = std::env::var("BPGPU_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        sys::check(unsafe { sys::bpgpu_init(dev) });
        // BPGPU_SHARDS=8: one context per GPU of the node for the large MSMs (bpgpu_msm_sharded)
        if let Some(p) = std::env::var("BPGPU_SHARDS").ok().and_then(|s| s.parse::<i32>().ok()) {
            if p > 1 {
                sys::check(unsafe { sys::bpgpu_sharded_init(p, std::ptr::null()) });
            }
        }
    });
}
/// MSMs at least this long are cut over the shard contexts when there are any
const SHARD_THRESHOLD: usize = 1 << 22;

// ---- limb plumbing ------------------------------------------------------------------------------------------------
fn fp_limbs<P: Fp256Parameters>(f: &Fp256<P>) -> [u64; 4] {
    (f.0).0
}
fn fp_from_limbs<P: Fp256Parameters>(l: &[u64]) -> Fp256<P> {
    Fp256::new(BigInteger256([l[0], l[1], l[2], l[3]]))
}
fn push_point<P>(out: &mut Vec<u64>, g: &GroupProjective<P>)
where
    P: SWModelParameters,
    P::BaseField: Limbs4,
{
    out.extend_from_slice(&g.x.limbs4());
    out.extend_from_slice(&g.y.limbs4());
    out.extend_from_slice(&g.z.limbs4());
}
/// A 256-bit prime field element as its four in-memory limbs (implemented for ark-ff's `Fp256`).
pub trait Limbs4: Sized {
    fn limbs4(&self) -> [u64; 4];
    fn from_limbs4(l: &[u64]) -> Self;
}
impl<P: Fp256Parameters> Limbs4 for Fp256<P> {
    fn limbs4(&self) -> [u64; 4] {
        fp_limbs(self)
    }
    fn from_limbs4(l: &[u64]) -> Self {
        fp_from_limbs(l)
    }
}
/// `xy_inf` (x | y | identity flag) -> `Projective { x, y, z: 1 }` or `zero()`
fn point_from_xy_inf<P>(o: &[u64; 9]) -> GroupProjective<P>
where
    P: SWModelParameters,
    P::BaseField: Limbs4 + One,
{
    if o[8] != 0 {
        return GroupProjective::<P>::zero();
    }
    GroupProjective::new(P::BaseField::from_limbs4(&o[0..4]), P::BaseField::from_limbs4(&o[4..8]), P::BaseField::one())
}
/// Jacobian triple with Z in {0, 1} as the library writes it (bpgpu_fold_points, bpgpu_ipa_export)
fn point_from_xyz<P>(o: &[u64]) -> GroupProjective<P>
where
    P: SWModelParameters,
    P::BaseField: Limbs4,
{
    GroupProjective::new(
        P::BaseField::from_limbs4(&o[0..4]),
        P::BaseField::from_limbs4(&o[4..8]),
        P::BaseField::from_limbs4(&o[8..12]),
    )
}

// ---- curve dispatch -------------------------------------------------------------------------------------------------
fn curve_id<G: Group>() -> Option<i32> {
    use std::any::TypeId;
    if TypeId::of::<G>() == TypeId::of::<ark_pallas::Projective>() {
        Some(sys::PALLAS)
    } else if TypeId::of::<G>() == TypeId::of::<ark_vesta::Projective>() {
        Some(sys::VESTA)
    } else {
        None
    }
}
fn down<G: Group, T: 'static>(g: &G) -> &T {
    (g as &dyn Any).downcast_ref::<T>().expect("curve_id checked the type")
}
fn up<G: Group, T: 'static + Copy>(t: T) -> G {
    *(&t as &dyn Any).downcast_ref::<G>().expect("curve_id checked the type")
}
fn gather_points<G: Group>(curve: i32, pts: &[&G]) -> Vec<u64> {
    let mut out = Vec::with_capacity(pts.len() * 12);
    for g in pts {
        if curve == sys::PALLAS {
            push_point(&mut out, down::<G, ark_pallas::Projective>(g));
        } else {
            push_point(&mut out, down::<G, ark_vesta::Projective>(g));
        }
    }
    out
}
fn gather_scalars<G: Group>(curve: i32, ss: &[&G::ScalarField]) -> Vec<u64> {
    let mut out = Vec::with_capacity(ss.len() * 4);
    for s in ss {
        let l = if curve == sys::PALLAS {
            fp_limbs(down_scalar::<G, ark_pallas::Fr>(s))
        } else {
            fp_limbs(down_scalar::<G, ark_vesta::Fr>(s))
        };
        out.extend_from_slice(&l);
    }
    out
}
fn down_scalar<G: Group, T: 'static>(s: &G::ScalarField) -> &T {
    (s as &dyn Any).downcast_ref::<T>().expect("curve_id checked the type")
}
fn scalar_limbs<G: Group>(curve: i32, s: &G::ScalarField) -> [u64; 4] {
    if curve == sys::PALLAS {
        fp_limbs(down_scalar::<G, ark_pallas::Fr>(s))
    } else {
        fp_limbs(down_scalar::<G, ark_vesta::Fr>(s))
    }
}
fn scalar_from<G: Group>(curve: i32, l: &[u64]) -> G::ScalarField {
    if curve == sys::PALLAS {
        let f: ark_pallas::Fr = fp_from_limbs(l);
        *(&f as &dyn Any).downcast_ref::<G::ScalarField>().unwrap()
    } else {
        let f: ark_vesta::Fr = fp_from_limbs(l);
        *(&f as &dyn Any).downcast_ref::<G::ScalarField>().unwrap()
    }
}
fn result_point<G: Group>(curve: i32, o: &[u64; 9]) -> G {
    if curve == sys::PALLAS {
        up::<G, ark_pallas::Projective>(point_from_xy_inf(o))
    } else {
        up::<G, ark_vesta::Projective>(point_from_xy_inf(o))
    }
}
fn points_from_xyz<G: Group>(curve: i32, buf: &[u64]) -> Vec<G> {
    buf.chunks_exact(12)
        .map(|c| {
            if curve == sys::PALLAS {
                up::<G, ark_pallas::Projective>(point_from_xyz(c))
            } else {
                up::<G, ark_vesta::Projective>(point_from_xyz(c))
            }
        })
        .collect()
}

// ---- util::msm (src/util/msm.rs:88-193) ---------------------------------------------------------------------------
/// `pippenger_msm` on the GPU: `bases` and `scalars` are the zipped (equal-length) operands of msm.rs:94-97.
pub fn msm<G: Group>(bases: &[&G], scalars: &[&G::ScalarField]) -> Option<G> {
    let curve = curve_id::<G>()?;
    init();
    let n = bases.len().min(scalars.len());
    let (p, s) = (gather_points::<G>(curve, &bases[..n]), gather_scalars::<G>(curve, &scalars[..n]));
    let mut out = [0u64; 9];
    let sharded = n >= SHARD_THRESHOLD && unsafe { sys::bpgpu_sharded_count() } > 1;
    sys::check(unsafe {
        if sharded {
            sys::bpgpu_msm_sharded(curve, p.as_ptr(), s.as_ptr(), n, out.as_mut_ptr())
        } else {
            sys::bpgpu_msm(curve, p.as_ptr(), s.as_ptr(), n, out.as_mut_ptr())
        }
    });
    Some(result_point::<G>(curve, &out))
}

// ---- Bp::prove (src/proofs/bp_iter.rs:49-128) ---------------------------------------------------------------------
/// The prover loop on a device-resident session: `(ls, rs, final_a, final_b)`.  Fiat-Shamir stays here, between
/// `round_lr` and `round_fold` (bp_iter.rs:84-87); zero padding (:59-62) happens inside `bpgpu_ipa_begin`.
pub fn bp_prove<G: Group>(
    a_gens: &[G],
    b_gens: &[G],
    ip_gen: &G,
    a: &[G::ScalarField],
    b: &[G::ScalarField],
    fs: &mut FiatShamirRng,
) -> Option<(Vec<G>, Vec<G>, G::ScalarField, G::ScalarField)> {
    let curve = curve_id::<G>()?;
    init();
    let n = a.len();
    assert_eq!(b.len(), n);
    assert_eq!(a_gens.len(), n);
    assert_eq!(b_gens.len(), n);
    let refs = |v: &[G]| v.iter().collect::<Vec<&G>>();
    let srefs = |v: &[G::ScalarField]| v.iter().collect::<Vec<&G::ScalarField>>();
    let (pa, pb, pq) = (
        gather_points::<G>(curve, &refs(a_gens)),
        gather_points::<G>(curve, &refs(b_gens)),
        gather_points::<G>(curve, &[ip_gen]),
    );
    let (sa, sb) = (gather_scalars::<G>(curve, &srefs(a)), gather_scalars::<G>(curve, &srefs(b)));
    let mut sess: *mut sys::bpgpu_ipa = std::ptr::null_mut();
    sys::check(unsafe {
        sys::bpgpu_ipa_begin(curve, pa.as_ptr(), pb.as_ptr(), pq.as_ptr(), sa.as_ptr(), sb.as_ptr(), n, &mut sess)
    });
    let (mut ls, mut rs) = (Vec::new(), Vec::new());
    while unsafe { sys::bpgpu_ipa_len(sess) } > 1 {
        let (mut l, mut r) = ([0u64; 9], [0u64; 9]);
        sys::check(unsafe { sys::bpgpu_ipa_round_lr(sess, l.as_mut_ptr(), r.as_mut_ptr()) }); // :68-81
        let (l, r) = (result_point::<G>(curve, &l), result_point::<G>(curve, &r));
        ls.push(l);
        rs.push(r);
        fs.absorb(&l); // :84-85
        fs.absorb(&r);
        let x: G::ScalarField = G::ScalarField::rand(fs); // :86-87
        let x_inv = x.inverse().unwrap();
        let (lx, lxi) = (scalar_limbs::<G>(curve, &x), scalar_limbs::<G>(curve, &x_inv));
        sys::check(unsafe { sys::bpgpu_ipa_round_fold(sess, lx.as_ptr(), lxi.as_ptr()) }); // :88-104
    }
    let (mut fa, mut fb) = ([0u64; 4], [0u64; 4]);
    sys::check(unsafe { sys::bpgpu_ipa_finish(sess, fa.as_mut_ptr(), fb.as_mut_ptr()) }); // :117-120
    unsafe { sys::bpgpu_ipa_free(sess) };
    Some((ls, rs, scalar_from::<G>(curve, &fa), scalar_from::<G>(curve, &fb)))
}

// ---- Bp::verify (src/proofs/bp_iter.rs:154-176) -------------------------------------------------------------------
/// Right-hand side of the verifier's check: the caller has replayed the transcript (:144-150) and compares the
/// returned point with `instance.result` (:176).  `a_gens` / `b_gens` are the UNPADDED generators.
pub fn bp_verify_msm<G: Group>(
    a_gens: &[G],
    b_gens: &[G],
    ip_gen: &G,
    ls: &[G],
    rs: &[G],
    challenges: &[G::ScalarField],
    final_a: &G::ScalarField,
    final_b: &G::ScalarField,
) -> Option<G> {
    let curve = curve_id::<G>()?;
    init();
    let refs = |v: &[G]| v.iter().collect::<Vec<&G>>();
    let (pa, pb, pq) = (
        gather_points::<G>(curve, &refs(a_gens)),
        gather_points::<G>(curve, &refs(b_gens)),
        gather_points::<G>(curve, &[ip_gen]),
    );
    let (pl, pr) = (gather_points::<G>(curve, &refs(ls)), gather_points::<G>(curve, &refs(rs)));
    let ch = gather_scalars::<G>(curve, &challenges.iter().collect::<Vec<_>>());
    let (fa, fb) = (scalar_limbs::<G>(curve, final_a), scalar_limbs::<G>(curve, final_b));
    let mut out = [0u64; 9];
    sys::check(unsafe {
        sys::bpgpu_ipa_verify_msm(
            curve, pa.as_ptr(), pb.as_ptr(), pq.as_ptr(), a_gens.len(), pl.as_ptr(), pr.as_ptr(), ls.len(), ch.as_ptr(),
            fa.as_ptr(), fb.as_ptr(), out.as_mut_ptr(),
        )
    });
    Some(result_point::<G>(curve, &out))
}

// ---- Bp2aryStep (src/reductions/bp_2ary_step.rs:24-125) -----------------------------------------------------------
/// One round behind the `Reduction` trait: `((l, r), a_gens', b_gens', a', b', x)`.  The caller computes `p_next`
/// (:59, three points) and assembles the next instance / witness (:72-86).
pub fn bp_2ary_prove<G: Group>(
    a_gens: &[G],
    b_gens: &[G],
    ip_gen: &G,
    a: &[G::ScalarField],
    b: &[G::ScalarField],
    fs: &mut FiatShamirRng,
) -> Option<((G, G), Vec<G>, Vec<G>, Vec<G::ScalarField>, Vec<G::ScalarField>, G::ScalarField)> {
    let curve = curve_id::<G>()?;
    init();
    let n = a.len();
    let refs = |v: &[G]| v.iter().collect::<Vec<&G>>();
    let srefs = |v: &[G::ScalarField]| v.iter().collect::<Vec<&G::ScalarField>>();
    let (pa, pb, pq) = (
        gather_points::<G>(curve, &refs(a_gens)),
        gather_points::<G>(curve, &refs(b_gens)),
        gather_points::<G>(curve, &[ip_gen]),
    );
    let (sa, sb) = (gather_scalars::<G>(curve, &srefs(a)), gather_scalars::<G>(curve, &srefs(b)));
    let mut sess: *mut sys::bpgpu_ipa = std::ptr::null_mut();
    sys::check(unsafe {
        sys::bpgpu_ipa_begin(curve, pa.as_ptr(), pb.as_ptr(), pq.as_ptr(), sa.as_ptr(), sb.as_ptr(), n, &mut sess)
    });
    // the reference asserts an even padded length (:41); a session of length 1 has no round to run
    assert!(unsafe { sys::bpgpu_ipa_len(sess) } % 2 == 0);
    let (mut l, mut r) = ([0u64; 9], [0u64; 9]);
    sys::check(unsafe { sys::bpgpu_ipa_round_lr(sess, l.as_mut_ptr(), r.as_mut_ptr()) }); // :43-46
    let (l, r) = (result_point::<G>(curve, &l), result_point::<G>(curve, &r));
    fs.absorb(&l); // :47-50
    fs.absorb(&r);
    let x: G::ScalarField = G::ScalarField::rand(fs);
    let x_inv = x.inverse().unwrap();
    let (lx, lxi) = (scalar_limbs::<G>(curve, &x), scalar_limbs::<G>(curve, &x_inv));
    sys::check(unsafe { sys::bpgpu_ipa_round_fold(sess, lx.as_ptr(), lxi.as_ptr()) }); // :51-67
    let m = unsafe { sys::bpgpu_ipa_len(sess) };
    let (mut ga, mut gb) = (vec![0u64; 12 * m], vec![0u64; 12 * m]);
    let (mut na, mut nb) = (vec![0u64; 4 * m], vec![0u64; 4 * m]);
    sys::check(unsafe {
        sys::bpgpu_ipa_export(sess, ga.as_mut_ptr(), gb.as_mut_ptr(), na.as_mut_ptr(), nb.as_mut_ptr())
    });
    unsafe { sys::bpgpu_ipa_free(sess) };
    let sc = |v: &[u64]| v.chunks_exact(4).map(|c| scalar_from::<G>(curve, c)).collect::<Vec<_>>();
    Some(((l, r), points_from_xyz::<G>(curve, &ga), points_from_xyz::<G>(curve, &gb), sc(&na), sc(&nb), x))
}

/// Generator fold of the verifier side (:106-113): `out[i] = g[i] * s_lo + g[i + n] * s_hi` over a padded vector of
/// length 2n (A: `(x_inv, x)`, B: `(x, x_inv)`).
pub fn fold_points<G: Group>(g: &[G], s_lo: &G::ScalarField, s_hi: &G::ScalarField) -> Option<Vec<G>> {
    let curve = curve_id::<G>()?;
    init();
    assert!(g.len() % 2 == 0);
    let n = g.len() / 2;
    let p = gather_points::<G>(curve, &g.iter().collect::<Vec<&G>>());
    let (lo, hi) = (scalar_limbs::<G>(curve, s_lo), scalar_limbs::<G>(curve, s_hi));
    let mut out = vec![0u64; 12 * n];
    sys::check(unsafe { sys::bpgpu_fold_points(curve, p.as_ptr(), n, lo.as_ptr(), hi.as_ptr(), out.as_mut_ptr()) });
    Some(points_from_xyz::<G>(curve, &out))
}

/// `ip` (src/util/mod.rs:17-23) for the scalar fields of the two curves; `None` for any other field.
pub fn ip<F: Field + 'static>(a: &[F], b: &[F]) -> Option<F> {
    use std::any::TypeId;
    // the scalar field of Pallas is the base field of Vesta and vice versa; ark-pallas / ark-vesta re-export one type
    let curve = if TypeId::of::<F>() == TypeId::of::<ark_pallas::Fr>() {
        sys::PALLAS
    } else if TypeId::of::<F>() == TypeId::of::<ark_vesta::Fr>() {
        sys::VESTA
    } else {
        return None;
    };
    init();
    assert_eq!(a.len(), b.len());
    let limbs = |v: &[F]| -> Vec<u64> {
        let mut out = Vec::with_capacity(4 * v.len());
        for f in v {
            if curve == sys::PALLAS {
                out.extend_from_slice(&fp_limbs((f as &dyn Any).downcast_ref::<ark_pallas::Fr>().unwrap()));
            } else {
                out.extend_from_slice(&fp_limbs((f as &dyn Any).downcast_ref::<ark_vesta::Fr>().unwrap()));
            }
        }
        out
    };
    let (la, lb) = (limbs(a), limbs(b));
    let mut out = [0u64; 4];
    sys::check(unsafe { sys::bpgpu_ip(curve, la.as_ptr(), lb.as_ptr(), a.len(), out.as_mut_ptr()) });
    Some(if curve == sys::PALLAS {
        let f: ark_pallas::Fr = fp_from_limbs(&out);
        *(&f as &dyn Any).downcast_ref::<F>().unwrap()
    } else {
        let f: ark_vesta::Fr = fp_from_limbs(&out);
        *(&f as &dyn Any).downcast_ref::<F>().unwrap()
    })
}
