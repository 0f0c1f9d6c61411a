//! Raw bindings of `include/frw.h`, one declaration per exported function, same order as the header.
//! Checked against the header by `tests/test_rust_binding.py` (names and parameter counts).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const FRW_OK: c_int = 0;
pub const FRW_E_INVALID_ARG: c_int = -1;
pub const FRW_E_NO_DEVICE: c_int = -2;
pub const FRW_E_HIP: c_int = -3;
pub const FRW_E_OUT_OF_MEMORY: c_int = -4;
pub const FRW_E_RANGE: c_int = -5;

pub const FRW_ENC_CANONICAL: c_int = 0;
/// x * 2^256 mod p: the in-memory limbs of `ark_ff::Fp256` (ark-ff 0.3)
pub const FRW_ENC_MONTGOMERY: c_int = 1;
pub const FRW_ENC_COMPACT: c_int = 2;

pub const FRW_ST_OK: i32 = 0;
pub const FRW_ST_COEFF_RANGE: i32 = 1;
pub const FRW_ST_NORM_BOUND: i32 = 2;
pub const FRW_ST_DECODE: i32 = 3;

pub const FRW_CIRCUIT_NTT: c_int = 0;
pub const FRW_CIRCUIT_DUAL_NTT: c_int = 1;
pub const FRW_NONCE_LEN: usize = 40;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct frw_layout_t {
    pub logn: i32,
    pub n: i32,
    pub num_witness: i32,
    pub num_instance: i32,
    pub num_constraints: i32,
    pub seg_off: [i32; 8],
    pub seg_len: [i32; 8],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct frw_layout_dual_t {
    pub logn: i32,
    pub n: i32,
    pub num_witness: i32,
    pub num_instance: i32,
    pub num_constraints: i32,
    pub seg_off: [i32; 15],
    pub seg_len: [i32; 15],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct frw_compact_layout_t {
    pub logn: i32,
    pub n: i32,
    pub bytes_per_signature: u64,
    pub small_off: u64,
    pub num_small: u64,
    pub t_off: u64,
    pub num_t: u64,
    pub bits_off: u64,
    pub num_bit_words: u64,
    pub bit_seg_off: [u64; 6],
    pub instance_off: u64,
    pub num_instance_values: u64,
    pub status_off: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct frw_msm_info_t {
    pub num_points: u64,
    pub window_bits: i32,
    pub num_windows: i32,
    pub table_bytes: u64,
    pub workspace_bytes_per_signature: u64,
}

#[repr(C)]
#[derive(Debug, Clone, Copy, Default)]
pub struct frw_r1cs_info_t {
    pub num_statements: u64,
    pub count_logn9: u64,
    pub count_logn10: u64,
    pub num_instance: u64,
    pub num_witness: u64,
    pub num_constraints: u64,
    pub log_domain_size: i32,
    pub witness_map_on_device: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct frw_qap_info_t {
    pub log_domain_size: i32,
    pub domain_size: u64,
    pub num_constraints: u64,
    pub num_instance: u64,
    pub workspace_bytes_per_signature: u64,
}

#[repr(C)]
pub struct frw_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct frw_r1cs {
    _private: [u8; 0],
}
#[repr(C)]
pub struct frw_msm {
    _private: [u8; 0],
}
#[repr(C)]
pub struct frw_groth16_pk {
    _private: [u8; 0],
}
#[repr(C)]
pub struct frw_groth16_vk {
    _private: [u8; 0],
}
pub const FRW_VERIFY_POINTS_ARE_CHECKED: c_int = 1;
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct frw_groth16_pk_desc_t {
    pub num_instance: u64,
    pub num_witness: u64,
    pub domain_size: u64,
    pub alpha_g1: *const u64,
    pub beta_g1: *const u64,
    pub delta_g1: *const u64,
    pub beta_g2: *const u64,
    pub delta_g2: *const u64,
    pub a_query: *const u64,
    pub b_g1_query: *const u64,
    pub b_g2_query: *const u64,
    pub h_query: *const u64,
    pub l_query: *const u64,
}
/// round 5: keys beyond what window tables can hold (the points only), and keys in slices (frw.h FRW_KEY_*)
pub const FRW_KEY_AUTO: i32 = 0;
pub const FRW_KEY_TABLES: i32 = 1;
pub const FRW_KEY_BARE: i32 = 2;
pub const FRW_GROTH16_PARTIAL_WORDS: usize = 72;
pub const FRW_GROTH16_COMBINE_WORKSPACE: usize = 4096;
pub const FRW_VK_POINTS_ARE_CHECKED: c_int = 1;
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct frw_groth16_key_opts_t {
    pub mode: i32,
    pub rank: u32,
    pub world: u32,
}
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct frw_groth16_pk_info_t {
    pub mode: i32,
    pub rank: u32,
    pub world: u32,
    pub z_lo: u64,
    pub z_hi: u64,
    pub h_lo: u64,
    pub h_hi: u64,
    pub key_bytes: u64,
}

extern "C" {
    pub fn frw_layout(logn: c_int, out: *mut frw_layout_t) -> c_int;
    pub fn frw_strerror(code: c_int) -> *const c_char;
    pub fn frw_last_error() -> *const c_char;
    pub fn frw_device_count() -> c_int;
    pub fn frw_ctx_create(device: c_int, out: *mut *mut frw_ctx) -> c_int;
    pub fn frw_ctx_destroy(ctx: *mut frw_ctx);
    pub fn frw_witness_ntt_verify_dev(ctx: *mut frw_ctx, logn: c_int, batch: usize, d_sig: *const u16, d_pk: *const u16,
                                      d_hm: *const u16, encoding: c_int, d_witness: *mut u64, d_instance: *mut u64,
                                      d_status: *mut i32, stream: *mut c_void) -> c_int;
    pub fn frw_ntt_modq_dev(ctx: *mut frw_ctx, logn: c_int, batch: usize, d_poly: *const u16, encoding: c_int,
                            d_witness: *mut u64, d_ntt_out: *mut u16, d_status: *mut i32, stream: *mut c_void) -> c_int;
    pub fn frw_witness_ntt_verify(ctx: *mut frw_ctx, logn: c_int, batch: usize, sig: *const u16, pk: *const u16,
                                  hm: *const u16, encoding: c_int, witness: *mut u64, instance: *mut u64,
                                  status: *mut i32, strict: c_int) -> c_int;
    pub fn frw_ntt_modq(ctx: *mut frw_ctx, logn: c_int, batch: usize, poly: *const u16, encoding: c_int,
                        witness: *mut u64, ntt_out: *mut u16, status: *mut i32) -> c_int;
    pub fn frw_diag_host_allocations(ctx: *mut frw_ctx, count: *mut u64) -> c_int;
    pub fn frw_ctx_trim(ctx: *mut frw_ctx) -> c_int;
    pub fn frw_diag_valu_rates(ctx: *mut frw_ctx, out: *mut f64) -> c_int;
    pub fn frw_compact_layout(logn: c_int, out: *mut frw_compact_layout_t) -> c_int;
    pub fn frw_witness_ntt_verify_compact_dev(ctx: *mut frw_ctx, logn: c_int, batch: usize, d_sig: *const u16,
                                              d_pk: *const u16, d_hm: *const u16, d_compact: *mut c_void,
                                              d_status: *mut i32, stream: *mut c_void) -> c_int;
    pub fn frw_expand_dev(ctx: *mut frw_ctx, logn: c_int, batch: usize, d_compact: *const c_void, d_witness: *mut u64,
                          d_instance: *mut u64, stream: *mut c_void) -> c_int;
    pub fn frw_expand_host(logn: c_int, batch: usize, compact: *const c_void, witness: *mut u64,
                           instance: *mut u64) -> c_int;
    pub fn frw_layout_dual(logn: c_int, out: *mut frw_layout_dual_t) -> c_int;
    pub fn frw_witness_dual_ntt_verify_dev(ctx: *mut frw_ctx, logn: c_int, batch: usize, d_sig: *const u16,
                                           d_pk: *const u16, d_hm: *const u16, encoding: c_int, d_witness: *mut u64,
                                           d_instance: *mut u64, d_status: *mut i32, stream: *mut c_void) -> c_int;
    pub fn frw_witness_dual_ntt_verify(ctx: *mut frw_ctx, logn: c_int, batch: usize, sig: *const u16, pk: *const u16,
                                       hm: *const u16, encoding: c_int, witness: *mut u64, instance: *mut u64,
                                       status: *mut i32, strict: c_int) -> c_int;
    pub fn frw_r1cs_export(circuit: c_int, logn: c_int, path: *const c_char, counts: *mut u64) -> c_int;
    pub fn frw_r1cs_load(device: c_int, circuit: c_int, logn: c_int, out: *mut *mut frw_r1cs) -> c_int;
    pub fn frw_r1cs_free(r: *mut frw_r1cs);
    pub fn frw_r1cs_load_aggregate(device: c_int, count: usize, logn: *const i32, out: *mut *mut frw_r1cs) -> c_int;
    pub fn frw_r1cs_info(r: *const frw_r1cs, out: *mut frw_r1cs_info_t) -> c_int;
    pub fn frw_aggregate_assign_dev(aggregate: *const frw_r1cs, d_witness_512: *const u64, d_instance_512: *const u64,
                                    d_witness_1024: *const u64, d_instance_1024: *const u64, d_witness: *mut u64,
                                    d_instance: *mut u64, stream: *mut c_void) -> c_int;
    pub fn frw_groth16_setup_r1cs(r: *const frw_r1cs, toxic: *const u64, pk_out: *mut *mut frw_groth16_pk,
                                  vk_out: *mut u64) -> c_int;
    pub fn frw_r1cs_check_dev(r: *const frw_r1cs, batch: usize, d_witness: *const u64, d_instance: *const u64,
                              d_num_unsatisfied: *mut u32, stream: *mut c_void) -> c_int;
    pub fn frw_r1cs_eval_dev(r: *const frw_r1cs, batch: usize, d_witness: *const u64, d_instance: *const u64,
                             d_num_unsatisfied: *mut u32, d_abc: *mut u64, stream: *mut c_void) -> c_int;
    pub fn frw_r1cs_eval_scratch_bytes(r: *const frw_r1cs, batch: usize, with_products: c_int) -> usize;
    pub fn frw_r1cs_eval_scratch_dev(r: *const frw_r1cs, batch: usize, d_witness: *const u64, d_instance: *const u64,
                                     d_num_unsatisfied: *mut u32, d_abc: *mut u64, d_scratch: *mut c_void,
                                     scratch_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn frw_qap_info(r: *const frw_r1cs, out: *mut frw_qap_info_t) -> c_int;
    pub fn frw_qap_witness_map_dev(r: *const frw_r1cs, batch: usize, d_witness: *const u64, d_instance: *const u64,
                                   d_h: *mut u64, d_num_unsatisfied: *mut u32, d_workspace: *mut c_void,
                                   workspace_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn frw_qap_quotient_dev(r: *const frw_r1cs, batch: usize, d_witness: *const u64, d_instance: *const u64,
                                d_h: *mut u64, d_num_unsatisfied: *mut u32, d_workspace: *mut c_void, workspace_bytes: usize,
                                stream: *mut c_void) -> c_int;
    pub fn frw_qap_witness_map(r: *const frw_r1cs, batch: usize, witness: *const u64, instance: *const u64, h: *mut u64,
                               num_unsatisfied: *mut u32) -> c_int;
    pub fn frw_r1cs_diag_host_allocations(r: *const frw_r1cs, count: *mut u64) -> c_int;
    pub fn frw_msm_g1_load(device: c_int, num_points: usize, bases: *const u64, out: *mut *mut frw_msm) -> c_int;
    pub fn frw_msm_free(m: *mut frw_msm);
    pub fn frw_g1_fixed_base(device: c_int, count: usize, scalars: *const u64, out: *mut u64) -> c_int;
    pub fn frw_g2_fixed_base(device: c_int, count: usize, scalars: *const u64, out: *mut u64) -> c_int;
    pub fn frw_msm_g2_load(device: c_int, num_points: usize, bases: *const u64, out: *mut *mut frw_msm) -> c_int;
    pub fn frw_msm_g1_load_narrow(device: c_int, num_points: usize, bases: *const u64, out: *mut *mut frw_msm) -> c_int;
    pub fn frw_msm_g2_load_narrow(device: c_int, num_points: usize, bases: *const u64, out: *mut *mut frw_msm) -> c_int;
    pub fn frw_msm_g2_dev(m: *const frw_msm, batch: usize, d_scalars: *const u64, scalar_stride: usize, montgomery: c_int,
                          d_out: *mut u64, d_workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn frw_msm_info(m: *const frw_msm, out: *mut frw_msm_info_t) -> c_int;
    pub fn frw_msm_g1_dev(m: *const frw_msm, batch: usize, d_scalars: *const u64, scalar_stride: usize, montgomery: c_int,
                          d_out: *mut u64, d_workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn frw_groth16_msm_h_dev(m: *const frw_msm, batch: usize, d_h: *const u64, domain_size: usize, d_out: *mut u64,
                                 d_workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn frw_groth16_pk_load(device: c_int, desc: *const frw_groth16_pk_desc_t, out: *mut *mut frw_groth16_pk) -> c_int;
    pub fn frw_groth16_pk_load_opts(device: c_int, desc: *const frw_groth16_pk_desc_t, opts: *const frw_groth16_key_opts_t,
                                    out: *mut *mut frw_groth16_pk) -> c_int;
    pub fn frw_groth16_pk_info(pk: *const frw_groth16_pk, out: *mut frw_groth16_pk_info_t) -> c_int;
    pub fn frw_groth16_pk_query(pk: *const frw_groth16_pk, which: c_int) -> *const frw_msm;
    pub fn frw_groth16_setup_r1cs_opts(r: *const frw_r1cs, toxic: *const u64, opts: *const frw_groth16_key_opts_t,
                                       pk_out: *mut *mut frw_groth16_pk, vk_out: *mut u64) -> c_int;
    pub fn frw_groth16_prove_partial_dev(pk: *const frw_groth16_pk, r: *const frw_r1cs, batch: usize, d_witness: *const u64,
                                         d_instance: *const u64, rs: *const u64, d_partial: *mut u64, d_num_unsatisfied: *mut u32,
                                         d_workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn frw_groth16_prove_combine_dev(pk: *const frw_groth16_pk, world: usize, d_partials: *const u64, rs: *const u64,
                                         d_proof: *mut u64, d_workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn frw_msm_g1_load_bare(device: c_int, num_points: usize, bases: *const u64, narrow: c_int, out: *mut *mut frw_msm) -> c_int;
    pub fn frw_msm_g2_load_bare(device: c_int, num_points: usize, bases: *const u64, narrow: c_int, out: *mut *mut frw_msm) -> c_int;
    pub fn frw_diag_poly_eval_dev(device: c_int, n: u64, d_coeffs: *const u64, t: *const u64, out: *mut u64) -> c_int;
    pub fn frw_diag_groth16_side_counts(pk: *const frw_groth16_pk, d_z: *const u64, d_workspace: *mut c_void, workspace_bytes: usize,
                                        stream: *mut c_void, out: *mut u64) -> c_int;
    pub fn frw_groth16_pk_free(pk: *mut frw_groth16_pk);
    pub fn frw_groth16_setup(device: c_int, circuit: c_int, logn: c_int, toxic: *const u64, pk_out: *mut *mut frw_groth16_pk,
                             vk_out: *mut u64) -> c_int;
    pub fn frw_groth16_workspace_bytes(pk: *const frw_groth16_pk, r: *const frw_r1cs, batch_in_flight: usize) -> usize;
    pub fn frw_groth16_prove_dev(pk: *const frw_groth16_pk, r: *const frw_r1cs, batch: usize, d_witness: *const u64,
                                 d_instance: *const u64, rs: *const u64, d_proofs: *mut u64, d_num_unsatisfied: *mut u32,
                                 d_workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn frw_groth16_prove_rs_dev(pk: *const frw_groth16_pk, r: *const frw_r1cs, batch: usize, d_witness: *const u64,
                                    d_instance: *const u64, d_rs: *const u64, d_proofs: *mut u64, d_num_unsatisfied: *mut u32,
                                    d_workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn frw_groth16_vk_load(vk: *const u64, num_instance: usize, out: *mut *mut frw_groth16_vk) -> c_int;
    pub fn frw_groth16_vk_load_opts(vk: *const u64, num_instance: usize, flags: c_int, out: *mut *mut frw_groth16_vk) -> c_int;
    pub fn frw_groth16_vk_free(vk: *mut frw_groth16_vk);
    pub fn frw_groth16_verify(vk: *const frw_groth16_vk, batch: usize, instance: *const u64, encoding: c_int, proofs: *const u64,
                              flags: c_int, accepted: *mut i32) -> c_int;
    pub fn frw_diag_pairing(g1: *const u64, g2: *const u64, out: *mut u64) -> c_int;
    pub fn frw_hash_to_point_dev(ctx: *mut frw_ctx, logn: c_int, batch: usize, d_nonces: *const u8, d_msgs: *const u8,
                                 d_msg_off: *const u64, d_hm: *mut u16, stream: *mut c_void) -> c_int;
    pub fn frw_decode_public_keys_dev(ctx: *mut frw_ctx, logn: c_int, batch: usize, d_pk_bytes: *const u8,
                                      d_pk: *mut u16, d_status: *mut i32, stream: *mut c_void) -> c_int;
    pub fn frw_decode_signatures_dev(ctx: *mut frw_ctx, logn: c_int, batch: usize, d_sig_bytes: *const u8,
                                     sig_len: usize, d_sig: *mut u16, d_nonce_out: *mut u8, d_status: *mut i32,
                                     stream: *mut c_void) -> c_int;
    pub fn frw_prepare_inputs(ctx: *mut frw_ctx, logn: c_int, batch: usize, pk_bytes: *const u8, sig_bytes: *const u8,
                              sig_len: usize, msgs: *const u8, msg_off: *const u64, sig: *mut u16, pk: *mut u16,
                              hm: *mut u16, status: *mut i32) -> c_int;
    pub fn frw_gadget_block_len(kind: c_int) -> c_int;
    pub fn frw_gadget_dev(ctx: *mut frw_ctx, kind: c_int, count: usize, d_a: *const c_void, d_b: *const u64,
                          encoding: c_int, d_out: *mut u64, d_status: *mut i32, stream: *mut c_void) -> c_int;
    pub fn frw_gadget(ctx: *mut frw_ctx, kind: c_int, count: usize, a: *const c_void, b: *const u64, encoding: c_int,
                      out: *mut u64, status: *mut i32) -> c_int;
    pub fn frw_digest_dev(ctx: *mut frw_ctx, d_buf: *const u64, words_per_item: usize, items: usize, d_out: *mut u64,
                          stream: *mut c_void) -> c_int;
    pub fn frw_diag_launch_shape(ctx: *mut frw_ctx, logn: c_int, encoding: c_int, batch: usize, out: *mut i32) -> c_int;
    pub fn frw_diag_write_stream_dev(ctx: *mut frw_ctx, d_buf: *mut c_void, bytes: usize, slab_bytes: usize,
                                     stream: *mut c_void) -> c_int;
    pub fn frw_synth_triples(logn: c_int, batch: usize, seed: u64, first_index: u64, sig: *mut u16, pk: *mut u16,
                             hm: *mut u16) -> c_int;
    pub fn frw_host_alloc(ctx: *mut frw_ctx, bytes: usize, ptr: *mut *mut c_void) -> c_int;
    pub fn frw_host_free(ctx: *mut frw_ctx, ptr: *mut c_void) -> c_int;
    pub fn frw_malloc(ctx: *mut frw_ctx, bytes: usize, d_ptr: *mut *mut c_void) -> c_int;
    pub fn frw_free(ctx: *mut frw_ctx, d_ptr: *mut c_void) -> c_int;
    pub fn frw_memcpy_h2d(ctx: *mut frw_ctx, d_dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn frw_memcpy_d2h(ctx: *mut frw_ctx, dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    pub fn frw_synchronize(ctx: *mut frw_ctx, stream: *mut c_void) -> c_int;
}
