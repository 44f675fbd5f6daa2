//! Raw declarations of include/zkt_plonk.h (the subset the shim uses).  Field elements cross as arkworks'
//! in-memory Montgomery limbs (`Fp256` = `BigInteger256([u64; 4])`), points as x limbs || y limbs, (0, 0) = infinity.
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct ZktCtx {
    _private: [u8; 0],
}

#[repr(C)]
pub struct ZktTranscriptVtable {
    pub user: *mut c_void,
    pub append_u64: extern "C" fn(*mut c_void, *const c_char, u64),
    pub append_scalars: extern "C" fn(*mut c_void, *const c_char, *const u64, usize, c_int),
    pub append_commitment: extern "C" fn(*mut c_void, *const c_char, *const u64, c_int),
    pub challenge_scalar: extern "C" fn(*mut c_void, *const c_char, *mut u64),
}

#[repr(C)]
pub struct ZktCommVtable {
    pub user: *mut c_void,
    pub rank: c_int,
    pub world: c_int,
    pub device_buffers: c_int,
    pub all_gather: extern "C" fn(*mut c_void, *const c_void, *mut c_void, usize, c_int, *mut c_void) -> c_int,
    /// optional stream-ordered form for device buffers (None: the library uses the blocking callback)
    pub all_gather_async: Option<extern "C" fn(*mut c_void, *const c_void, *mut c_void, usize, *mut c_void) -> c_int>,
}

#[repr(C)]
pub struct ZktProveInputs {
    pub a_evals: *const u64,
    pub b_evals: *const u64,
    pub c_evals: *const u64,
    pub n_rows: usize,
    pub table: *const u64,
    pub table_len: usize,
    pub pi_pos: *const usize,
    pub pi_vals: *const u64,
    pub n_pi: usize,
    pub blinders: *const u64,
    pub wires_on_device: c_int,
    pub variables: *const u64,
    pub n_vars: usize,
    pub w_l: *const u32,
    pub w_r: *const u32,
    pub w_o: *const u32,
}

pub const ZKT_VARIABLE_ZERO: u32 = 0xFFFF_FFFF;
pub const ZKT_CURVE_BN254: c_int = 0;
pub const ZKT_CURVE_BLS12_381: c_int = 1;

extern "C" {
    pub fn zkt_ctx_create(curve_id: c_int, device_id: c_int, out: *mut *mut ZktCtx) -> c_int;
    pub fn zkt_ctx_destroy(ctx: *mut ZktCtx);
    pub fn zkt_last_error(ctx: *const ZktCtx) -> *const c_char;
    pub fn zkt_ctx_set_comm(ctx: *mut ZktCtx, comm: *const ZktCommVtable) -> c_int;
    pub fn zkt_shard_range(total: usize, rank: c_int, world: c_int, lo: *mut usize, hi: *mut usize) -> c_int;
    pub fn zkt_ntt(ctx: *mut ZktCtx, log_n: c_int, inverse: c_int, coset: c_int, input: *const u64, in_len: usize,
                   out: *mut u64) -> c_int;
    pub fn zkt_srs_load(ctx: *mut ZktCtx, g1_xy_mont: *const u64, count: usize) -> c_int;
    pub fn zkt_srs_load_slice(ctx: *mut ZktCtx, g1_xy_mont: *const u64, offset: usize, count: usize, total: usize) -> c_int;
    pub fn zkt_srs_load_file(ctx: *mut ZktCtx, ck_path: *const c_char, max_powers: usize) -> c_int;
    pub fn zkt_msm_g1(ctx: *mut ZktCtx, scalars: *const u64, len: usize, base_offset: usize, scalars_montgomery: c_int,
                      out_xy_mont: *mut u64, out_is_infinity: *mut c_int) -> c_int;
    pub fn zkt_ctx_fork(ctx: *mut ZktCtx, out: *mut *mut ZktCtx) -> c_int;
    pub fn zkt_commit_evals_dev(ctx: *mut ZktCtx, d_evals: *const c_void, blinders: *const u64, k: c_int, path: c_int,
                                out_xy_mont: *mut u64, out_is_infinity: *mut c_int) -> c_int;
    pub fn zkt_ctx_set_lagrange(ctx: *mut ZktCtx, on: c_int) -> c_int;
    pub fn zkt_lagrange_info(ctx: *mut ZktCtx, log_n: *mut c_int, bases: *mut usize) -> c_int;
    pub fn zkt_circuit_load(ctx: *mut ZktCtx, log_n: c_int, pk_polys: *const *const u64, pk_lens: *const usize) -> c_int;
    pub fn zkt_circuit_load_file(ctx: *mut ZktCtx, pk_path: *const c_char, log_n: c_int) -> c_int;
    pub fn zkt_keyfile_extended_prover_key(path: *const c_char, curve_id: c_int, which: c_int, out_mont: *mut u64, cap: usize,
                                           lens17: *mut usize) -> c_int;
    pub fn zkt_circuit_check_epk_file(ctx: *mut ZktCtx, epk_path: *const c_char, first_mismatch_vector: *mut c_int,
                                      mismatch_at: *mut usize) -> c_int;
    pub fn zkt_circuit_setup(ctx: *mut ZktCtx, log_n: c_int, evals: *const *const u64, eval_lens: *const usize,
                             evals_on_device: c_int, out_commitments: *mut u64, out_is_infinity: *mut c_int) -> c_int;
    pub fn zkt_prove_with(ctx: *mut ZktCtx, inputs: *const ZktProveInputs, transcript: *const ZktTranscriptVtable,
                          proof_out: *mut u8, proof_cap: usize, proof_len: *mut usize) -> c_int;
    pub fn zkt_prove_set_next(ctx: *mut ZktCtx, next: *const ZktProveInputs) -> c_int;
    pub fn zkt_g1_msm_host(curve_id: c_int, points_xy_mont: *const u64, scalars: *const u64, n: usize, scalars_montgomery: c_int,
                           out_xy_mont: *mut u64, out_is_infinity: *mut c_int) -> c_int;
    pub fn zkt_verify(curve_id: c_int, inputs: *const ZktVerifyInputs, transcript: *mut c_void, h_g2_mont: *const u64,
                      beta_h_g2_mont: *const u64, accepted: *mut c_int) -> c_int;
    pub fn zkt_verify_batch(curve_id: c_int, inputs: *const ZktVerifyInputs, transcripts: *const *mut c_void, count: usize,
                            h_g2_mont: *const u64, beta_h_g2_mont: *const u64, accepted: *mut c_int) -> c_int;
    pub fn zkt_poseidon_load(ctx: *mut ZktCtx, params: *const ZktPoseidonParams, out: *mut *mut c_void) -> c_int;
    pub fn zkt_poseidon_free(ctx: *mut ZktCtx, params: *mut c_void);
    pub fn zkt_poseidon_hash_batch_dev(ctx: *mut ZktCtx, params: *const c_void, d_inputs: *const c_void, batch: usize,
                                       arity: c_int, d_out_hashes: *mut c_void, d_out_states: *mut c_void) -> c_int;
    pub fn zkt_poseidon_gadget_vars_per_hash(params: *const c_void) -> usize;
    pub fn zkt_poseidon_gadget_witness_dev(ctx: *mut ZktCtx, params: *const c_void, args: *const ZktPoseidonGadgetArgs) -> c_int;
    pub fn zkt_poseidon_gadget_check(ctx: *mut ZktCtx, params: *const c_void) -> c_int;
    pub fn zkt_dev_alloc(ctx: *mut ZktCtx, bytes: usize, dptr: *mut *mut c_void) -> c_int;
    pub fn zkt_dev_free(ctx: *mut ZktCtx, dptr: *mut c_void) -> c_int;
    pub fn zkt_dev_upload(ctx: *mut ZktCtx, dptr: *mut c_void, host: *const c_void, bytes: usize) -> c_int;
}

/// include/zkt_comm_rccl.h: the optional RCCL transport (libzkt_comm_rccl.so), INTEGRATION.md section 3c
/// Behind the `rccl` cargo feature: a host without librccl builds and links the shim without it.
#[cfg(feature = "rccl")]
#[link(name = "zkt_comm_rccl")]
extern "C" {
    pub fn zkt_comm_rccl_unique_id(out: *mut u8) -> c_int;
    pub fn zkt_comm_rccl_create(id: *const u8, rank: c_int, world: c_int, device: c_int, out: *mut *mut c_void) -> c_int;
    pub fn zkt_comm_rccl_vtable(comm: *mut c_void, out: *mut ZktCommVtable) -> c_int;
    pub fn zkt_comm_rccl_destroy(comm: *mut c_void);
    pub fn zkt_comm_rccl_last_error(comm: *const c_void) -> *const c_char;
}

#[repr(C)]
pub struct ZktPoseidonGadgetArgs {
    pub batch: usize,
    pub arity: c_int,
    pub d_inputs: *const c_void,
    pub d_input_vars: *const u32,
    pub d_variables: *mut c_void,
    pub n_vars: usize,
    pub d_trace_base: *const u32,
    pub trace_base0: usize,
    pub d_out_hashes: *mut c_void,
    pub kernel: c_int,
}

#[repr(C)]
pub struct ZktVerifyInputs {
    pub n: u64,
    pub vk_commitments: *const u64,
    pub vk_is_infinity: *const c_int,
    pub pi_roots: *const u64,
    pub pub_inputs: *const u64,
    pub n_pi: usize,
    pub proof: *const u8,
    pub proof_len: usize,
    pub g: *const u64,
}

#[repr(C)]
pub struct ZktPoseidonParams {
    pub width: c_int,
    pub half_full_rounds: c_int,
    pub partial_rounds: c_int,
    pub round_constants: *const u64,
    pub mds: *const u64,
    pub domain_tag: *const u64,
}
