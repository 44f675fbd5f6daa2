//! The body of `proof_system::prove` (plonk-core/src/proof_system/prove.rs:75-470) as ONE call into the library.
//! `ProvingComposer` is crate-private in plonk-core, so this file is meant to live inside plonk-core as
//! `src/gpu/prover.rs` under `cfg(feature = "gpu")`; it is kept here so that the binding is complete as source.
//! UNCOMPILED (see lib.rs).  No stubs: the callbacks convert commitments through the function the caller supplies
//! (`kzg::commitment_from_limbs::<E>`) and never unwind across the C boundary.
use crate::{check, ffi, with_ctx};
use ark_ff::{FftField, PrimeField};
use ark_serialize::CanonicalDeserialize;
use plonk_core::{commitment::HomomorphicCommitment, error::Error, transcript::TranscriptProtocol};
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

/// The labels the library passes are the literals of prove.rs / keys/mod.rs:264-274: map the C string back to the
/// `&'static str` the trait wants.
fn label(l: *const c_char) -> &'static str {
    const LABELS: &[&str] = &["pi", "a_commit", "b_commit", "c_commit", "t_commit", "h1_commit", "h2_commit", "beta", "gamma",
        "delta", "epsilon", "z1_commit", "z2_commit", "alpha", "q_lo_commit", "q_mid_commit", "q_hi_commit", "xi", "a_eval",
        "b_eval", "c_eval", "sigma1_eval", "sigma2_eval", "z1_next_eval", "q_lookup_eval", "t_eval", "t_next_eval",
        "z2_next_eval", "h1_next_eval", "h2_eval", "eta"];
    let s = unsafe { CStr::from_ptr(l) }.to_str().expect("ascii label");
    LABELS.iter().copied().find(|x| *x == s).expect("label of prove.rs")
}

unsafe fn frs<'a, F: PrimeField>(p: *const u64, k: usize) -> &'a [F] { core::slice::from_raw_parts(p as *const F, k) }

/// What the callbacks find behind `user`: the transcript and the way from the library's (x, y) limbs to `PC::Commitment`
/// (`kzg::commitment_from_limbs::<E>` for KZG10<E> / GpuKZG10<E>).  A callback that panics must not unwind into C: the
/// panic is caught, remembered here and re-raised by `prove_gpu` after the library has returned.
pub struct CallbackState<'t, F: PrimeField, PC: HomomorphicCommitment<F>, T: TranscriptProtocol<F, PC::Commitment>> {
    transcript: &'t mut T,
    commitment_from_limbs: fn(*const u64, c_int) -> PC::Commitment,
    panic: Option<Box<dyn std::any::Any + Send + 'static>>,
    _f: core::marker::PhantomData<F>,
}

fn guarded<F, PC, T>(u: *mut c_void, body: impl FnOnce(&mut CallbackState<'_, F, PC, T>))
where
    F: PrimeField,
    PC: HomomorphicCommitment<F>,
    T: TranscriptProtocol<F, PC::Commitment>,
{
    let st = unsafe { &mut *(u as *mut CallbackState<'_, F, PC, T>) };
    if st.panic.is_some() {
        return;   // already failed: do nothing more, prove_gpu reports it
    }
    let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| body(st)));
    if let Err(e) = r {
        let st = unsafe { &mut *(u as *mut CallbackState<'_, F, PC, T>) };
        st.panic = Some(e);
    }
}

/// `T: TranscriptProtocol<F, PC::Commitment>` behind the four callbacks of `zkt_transcript_vtable`.  Rust closures cannot
/// be `extern "C"`: each callback is a monomorphised function that recovers the state from `user`.
pub fn vtable<F, PC, T>(state: &mut CallbackState<'_, F, PC, T>) -> ffi::ZktTranscriptVtable
where
    F: PrimeField,
    PC: HomomorphicCommitment<F>,
    T: TranscriptProtocol<F, PC::Commitment>,
{
    ffi::ZktTranscriptVtable {
        user: state as *mut CallbackState<'_, F, PC, T> as *mut c_void,
        append_u64: cb_append_u64::<F, PC, T>,
        append_scalars: cb_append_scalars::<F, PC, T>,
        append_commitment: cb_append_commitment::<F, PC, T>,
        challenge_scalar: cb_challenge_scalar::<F, PC, T>,
    }
}
extern "C" fn cb_append_u64<F: PrimeField, PC: HomomorphicCommitment<F>, T: TranscriptProtocol<F, PC::Commitment>>(u: *mut c_void, l: *const c_char, v: u64) {
    guarded::<F, PC, T>(u, |st| st.transcript.append_u64(label(l), v))
}
extern "C" fn cb_append_scalars<F: PrimeField, PC: HomomorphicCommitment<F>, T: TranscriptProtocol<F, PC::Commitment>>(u: *mut c_void, l: *const c_char, p: *const u64, k: usize, single: c_int) {
    guarded::<F, PC, T>(u, |st| {
        if single != 0 { st.transcript.append_scalar(label(l), &unsafe { frs::<F>(p, 1) }[0]) }
        else { st.transcript.append_scalars(label(l), unsafe { frs::<F>(p, k) }.iter()) }
    })
}
extern "C" fn cb_append_commitment<F: PrimeField, PC: HomomorphicCommitment<F>, T: TranscriptProtocol<F, PC::Commitment>>(u: *mut c_void, l: *const c_char, xy: *const u64, inf: c_int) {
    guarded::<F, PC, T>(u, |st| {
        let c = (st.commitment_from_limbs)(xy, inf);
        st.transcript.append_commitment(label(l), &c)
    })
}
extern "C" fn cb_challenge_scalar<F: PrimeField, PC: HomomorphicCommitment<F>, T: TranscriptProtocol<F, PC::Commitment>>(u: *mut c_void, l: *const c_char, out: *mut u64) {
    // on a panic the output keeps the zero the library initialised it with; the library then fails with
    // ZKT_ERR_EQUAL_CHALLENGES at the next distinctness check and prove_gpu re-raises the panic
    guarded::<F, PC, T>(u, |st| unsafe { *(out as *mut F) = st.transcript.challenge_scalar(label(l)) })
}

/// What `prove` hands over: wire evaluations (or the composer's variables + indices), the lookup table in IndexSet
/// order, the public inputs in BTreeMap order and the 19 blinders in the reference's draw order
/// (prove.rs:125-127,170-171,225,244,296).  Returns the CanonicalSerialize bytes of the Proof (proof.rs:106-155).
#[allow(clippy::too_many_arguments)]
pub fn prove_gpu<F, PC, T>(
    a: &[F], b: &[F], c: &[F], table: &[F], pi_pos: &[usize], pi_vals: &[F], blinders: &[F; 19],
    transcript: &mut T, commitment_from_limbs: fn(*const u64, c_int) -> PC::Commitment,
) -> Result<Vec<u8>, Error>
where
    F: PrimeField + FftField,
    PC: HomomorphicCommitment<F>,
    T: TranscriptProtocol<F, PC::Commitment>,
{
    let inputs = ffi::ZktProveInputs {
        a_evals: a.as_ptr() as *const u64, b_evals: b.as_ptr() as *const u64, c_evals: c.as_ptr() as *const u64,
        n_rows: a.len(), table: table.as_ptr() as *const u64, table_len: table.len(),
        pi_pos: pi_pos.as_ptr(), pi_vals: pi_vals.as_ptr() as *const u64, n_pi: pi_pos.len(),
        blinders: blinders.as_ptr() as *const u64, wires_on_device: 0,
        variables: core::ptr::null(), n_vars: 0, w_l: core::ptr::null(), w_r: core::ptr::null(), w_o: core::ptr::null(),
    };
    let mut state = CallbackState::<F, PC, T> { transcript, commitment_from_limbs, panic: None, _f: core::marker::PhantomData };
    let vt = vtable::<F, PC, T>(&mut state);
    let mut bytes = vec![0u8; 1024];
    let mut len = 0usize;
    let rc = with_ctx::<F, _>(|ctx| check(ctx, unsafe { ffi::zkt_prove_with(ctx, &inputs, &vt, bytes.as_mut_ptr(), bytes.len(), &mut len) }));
    if let Some(p) = state.panic.take() {
        std::panic::resume_unwind(p);   // a transcript callback panicked: the panic continues here, on the Rust side
    }
    rc?;
    bytes.truncate(len);
    Ok(bytes)   // the caller: Proof::<F, D, PC>::deserialize(&bytes[..]) (proof.rs:98-155)
}

#[allow(dead_code)]
fn _deserialize_hint<P: CanonicalDeserialize>(bytes: &[u8]) -> Result<P, Error> {
    P::deserialize(bytes).map_err(|e| Error::PCError { error: e.to_string() })
}
