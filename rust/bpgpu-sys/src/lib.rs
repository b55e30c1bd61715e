//! Raw declarations of the C ABI in `include/bpgpu.h` (one per prototype, same order), plus the two helpers every
//! caller needs: `check` (non-zero return -> `panic!`, the reference's only error convention on this path) and
//! `last_error`.
//!
//! Data formats (see the header): a scalar is ark-ff 0.3's in-memory `Fp256` (4 x u64, value * 2^256 mod p,
//! canonical); a Jacobian point is `X | Y | Z` (12 x u64, `Z = 0` the identity); results come back as `xy_inf`
//! (9 x u64: affine x, y and an identity flag).  `curve`: `PALLAS` or `VESTA`.
//!
//! This file is not compiled in the repository's build image (no Rust toolchain there); `tests/test_rust_ffi.py`
//! compares it with `include/bpgpu.h` name by name, argument by argument.
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_void};

pub const PALLAS: i32 = 0;
pub const VESTA: i32 = 1;
pub const BPGPU_OK: i32 = 0;
pub const BPGPU_E_ARG: i32 = -1;
pub const BPGPU_E_HIP: i32 = -2;
pub const BPGPU_E_NOINIT: i32 = -3;
pub const BPGPU_E_STATE: i32 = -4;
/// points per `bpgpu_msm*` call (and per shard of a sharded MSM)
pub const BPGPU_MSM_MAX_N: usize = 1 << 28;
/// elements per vector of an IPA session
pub const BPGPU_IPA_MAX_N: usize = 1 << 26;
pub const BPGPU_IP_NOALLOC_MAX_N: usize = 0xffff_ffff;

/// Opaque device-resident prover session (`bpgpu_ipa_begin` .. `bpgpu_ipa_free`).
#[repr(C)]
pub struct bpgpu_ipa {
    _private: [u8; 0],
}
/// Opaque handle of generators resident on every shard context (`bpgpu_sharded_gens_*`).
#[repr(C)]
pub struct bpgpu_sharded_gens {
    _private: [u8; 0],
}

extern "C" {
    pub fn bpgpu_init(device: i32) -> i32;
    pub fn bpgpu_shutdown();
    pub fn bpgpu_last_error() -> *const c_char;
    pub fn bpgpu_set_window_bits(c: i32) -> i32;
    pub fn bpgpu_set_freeze_len(len: usize) -> i32;
    pub fn bpgpu_set_pair_fold(enable: i32) -> i32;
    pub fn bpgpu_set_ipa_table(min_len: usize) -> i32;
    pub fn bpgpu_reset_tuning();
    pub fn bpgpu_get_tuning(out: *mut usize) -> i32;
    pub fn bpgpu_msm(curve: i32, points_xyz: *const u64, scalars: *const u64, n: usize, out_xy_inf: *mut u64) -> i32;
    pub fn bpgpu_msm_affine(curve: i32, points_xy: *const u64, scalars: *const u64, n: usize, out_xy_inf: *mut u64) -> i32;
    pub fn bpgpu_ip(curve: i32, a: *const u64, b: *const u64, n: usize, out: *mut u64) -> i32;
    pub fn bpgpu_fold_scalars(curve: i32, v: *const u64, n_out: usize, s_lo: *const u64, s_hi: *const u64, out: *mut u64) -> i32;
    pub fn bpgpu_fold_points(curve: i32, g_xyz: *const u64, n_out: usize, s_lo: *const u64, s_hi: *const u64, out_xyz: *mut u64) -> i32;
    pub fn bpgpu_dev_alloc(bytes: usize, dptr: *mut *mut c_void) -> i32;
    pub fn bpgpu_dev_free(dptr: *mut c_void) -> i32;
    pub fn bpgpu_dev_upload(dptr: *mut c_void, host: *const c_void, bytes: usize) -> i32;
    pub fn bpgpu_dev_download(host: *mut c_void, dptr: *const c_void, bytes: usize) -> i32;
    pub fn bpgpu_points_to_device(curve: i32, points: *const u64, n: usize, fmt: i32, d_out: *mut c_void) -> i32;
    pub fn bpgpu_points_from_device(curve: i32, d_points: *const c_void, n: usize, out_xyz: *mut u64) -> i32;
    pub fn bpgpu_msm_device(curve: i32, d_points: *const c_void, d_scalars: *const c_void, n: usize, out_xy_inf: *mut u64) -> i32;
    pub fn bpgpu_ip_device(curve: i32, d_a: *const c_void, d_b: *const c_void, n: usize, out: *mut u64) -> i32;
    pub fn bpgpu_ip_device_noalloc(curve: i32, d_a: *const c_void, d_b: *const c_void, n: usize, d_out: *mut c_void) -> i32;
    pub fn bpgpu_fold_scalars_device(curve: i32, d_v: *const c_void, n_out: usize, s_lo: *const u64, s_hi: *const u64, d_out: *mut c_void) -> i32;
    pub fn bpgpu_fold_points_device(curve: i32, d_g: *const c_void, n_out: usize, s_lo: *const u64, s_hi: *const u64, d_out: *mut c_void) -> i32;
    pub fn bpgpu_fold_scalars_kary(curve: i32, v: *const u64, ck: usize, k: usize, y: *const u64, out: *mut u64) -> i32;
    pub fn bpgpu_fold_points_kary(curve: i32, g_xyz: *const u64, ck: usize, k: usize, y: *const u64, out_xyz: *mut u64) -> i32;
    pub fn bpgpu_fold_scalars_kary_device(curve: i32, d_v: *const c_void, ck: usize, k: usize, y: *const u64, d_out: *mut c_void) -> i32;
    pub fn bpgpu_fold_points_kary_device(curve: i32, d_g: *const c_void, ck: usize, k: usize, y: *const u64, d_out: *mut c_void) -> i32;
    pub fn bpgpu_kary_cross_terms(curve: i32, a_gens_xyz: *const u64, b_gens_xyz: *const u64, ip_gen_xyz: *const u64, a: *const u64, b: *const u64, ck: usize, k: usize, pos_xy_inf: *mut u64, neg_xy_inf: *mut u64) -> i32;
    pub fn bpgpu_hadamard(curve: i32, a: *const u64, b: *const u64, n: usize, out: *mut u64) -> i32;
    pub fn bpgpu_hadamard_points(curve: i32, bases_xyz: *const u64, scalars: *const u64, n: usize, out_xyz: *mut u64) -> i32;
    pub fn bpgpu_powers(curve: i32, f: *const u64, start: u64, n: usize, out: *mut u64) -> i32;
    pub fn bpgpu_mat_vec_mult(curve: i32, row_ptr: *const u64, col_idx: *const u64, vals: *const u64, n_rows: usize, vec: *const u64, vec_len: usize, out: *mut u64) -> i32;
    pub fn bpgpu_vec_mat_mult(curve: i32, vec: *const u64, vec_len: usize, row_ptr: *const u64, col_idx: *const u64, vals: *const u64, n_rows: usize, output_size: usize, out: *mut u64) -> i32;
    pub fn bpgpu_curve_lift_x(curve: i32, xs: *const u64, greatest: *const u8, m: usize, ys: *mut u64, ok: *mut u8) -> i32;
    pub fn bpgpu_ipa_begin(curve: i32, a_gens_xyz: *const u64, b_gens_xyz: *const u64, ip_gen_xyz: *const u64, a: *const u64, b: *const u64, n: usize, out: *mut *mut bpgpu_ipa) -> i32;
    pub fn bpgpu_ipa_begin_device(curve: i32, d_a_gens: *const c_void, d_b_gens: *const c_void, d_ip_gen: *const c_void, a: *const u64, b: *const u64, n: usize, out: *mut *mut bpgpu_ipa) -> i32;
    pub fn bpgpu_ipa_len(s: *const bpgpu_ipa) -> usize;
    pub fn bpgpu_ipa_round_lr(s: *mut bpgpu_ipa, l_xy_inf: *mut u64, r_xy_inf: *mut u64) -> i32;
    pub fn bpgpu_ipa_round_fold(s: *mut bpgpu_ipa, x: *const u64, x_inv: *const u64) -> i32;
    pub fn bpgpu_ipa_finish(s: *mut bpgpu_ipa, final_a: *mut u64, final_b: *mut u64) -> i32;
    pub fn bpgpu_ipa_export(s: *mut bpgpu_ipa, a_gens_xyz: *mut u64, b_gens_xyz: *mut u64, a: *mut u64, b: *mut u64) -> i32;
    pub fn bpgpu_ipa_free(s: *mut bpgpu_ipa);
    pub fn bpgpu_ipa_verify_msm(curve: i32, a_gens_xyz: *const u64, b_gens_xyz: *const u64, ip_gen_xyz: *const u64, n: usize, ls_xyz: *const u64, rs_xyz: *const u64, k: usize, challenges: *const u64, final_a: *const u64, final_b: *const u64, out_xy_inf: *mut u64) -> i32;
    pub fn bpgpu_ipa_verify_msm_device(curve: i32, d_a_gens: *const c_void, d_b_gens: *const c_void, d_ip_gen: *const c_void, n: usize, ls_xyz: *const u64, rs_xyz: *const u64, k: usize, challenges: *const u64, final_a: *const u64, final_b: *const u64, out_xy_inf: *mut u64) -> i32;
    pub fn bpgpu_sharded_init(n_ctx: i32, devices: *const i32) -> i32;
    pub fn bpgpu_sharded_count() -> i32;
    pub fn bpgpu_msm_sharded(curve: i32, points_xyz: *const u64, scalars: *const u64, n: usize, out_xy_inf: *mut u64) -> i32;
    pub fn bpgpu_msm_sharded_affine(curve: i32, points_xy: *const u64, scalars: *const u64, n: usize, out_xy_inf: *mut u64) -> i32;
    pub fn bpgpu_sharded_gens_load(curve: i32, points: *const u64, n: usize, fmt: i32, out: *mut *mut bpgpu_sharded_gens) -> i32;
    pub fn bpgpu_sharded_gens_msm(g: *mut bpgpu_sharded_gens, scalars: *const u64, out_xy_inf: *mut u64) -> i32;
    pub fn bpgpu_sharded_gens_free(g: *mut bpgpu_sharded_gens);
    pub fn bpgpu_points_sum(curve: i32, xy_inf: *const u64, count: usize, out_xy_inf: *mut u64) -> i32;
    pub fn bpgpu_last_msm_timings(out_ms: *mut f32) -> i32;
    pub fn bpgpu_last_msm_entries(entries: *mut u64) -> i32;
    pub fn bpgpu_last_op_ms(ms: *mut f32) -> i32;
    pub fn bpgpu_last_fold_ops(out: *mut u64) -> i32;
    pub fn bpgpu_measure_mad_peak(mads_per_sec: *mut f64) -> i32;
}

/// Message of the last failed call on this thread.
pub fn last_error() -> String {
    unsafe { CStr::from_ptr(bpgpu_last_error()).to_string_lossy().into_owned() }
}

/// The reference panics on every error of this path (`assert!` / `unwrap`); so does the shim.
#[track_caller]
pub fn check(rc: i32) {
    if rc != BPGPU_OK {
        panic!("bpgpu error {}: {}", rc, last_error());
    }
}
