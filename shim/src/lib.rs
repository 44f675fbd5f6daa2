//! `zkt-plonk-gpu`: the two generic seams of `ZKTPlonk<F, D, PC, T, C, TABLE_SIZE>` (plonk-core/src/plonk.rs:39-52)
//! bound to the MI355X library.  SOURCE ONLY: UNCOMPILED AND UNTESTED -- the authoring image has no Rust toolchain and
//! none of the crates; expect to fix signatures against ark-poly / ark-poly-commit 0.3 when it first meets a compiler.
//! The C-ABI underneath is what the parity tests exercise.  INTEGRATION.md walks through the pieces.
pub mod ffi;

use ark_ff::{FftField, PrimeField};
use ark_poly::{domain::DomainCoeff, EvaluationDomain, Radix2EvaluationDomain};
use std::cell::RefCell;
use std::ffi::CStr;
use std::os::raw::c_int;

thread_local! {
    // one context per proving thread (`prove` is !Send: Rc<ExtendedProverKey>, prove.rs:62)
    static CTX: RefCell<Option<*mut ffi::ZktCtx>> = RefCell::new(None);
}

/// Maps the scalar field to the library's curve id by its modulus size / two-adicity.
fn curve_of<F: PrimeField + FftField>() -> c_int {
    if F::size_in_bits() == 254 { ffi::ZKT_CURVE_BN254 } else { ffi::ZKT_CURVE_BLS12_381 }
}

pub fn with_ctx<F: PrimeField + FftField, R>(f: impl FnOnce(*mut ffi::ZktCtx) -> R) -> R {
    CTX.with(|slot| {
        let mut slot = slot.borrow_mut();
        if slot.is_none() {
            let mut ctx = std::ptr::null_mut();
            let rc = unsafe { ffi::zkt_ctx_create(curve_of::<F>(), 0, &mut ctx) };
            assert_eq!(rc, 0, "zkt_ctx_create failed: no MI355X visible (there is no CPU fallback)");
            *slot = Some(ctx);
        }
        f(slot.unwrap())
    })
}

pub fn check(ctx: *mut ffi::ZktCtx, rc: c_int) -> Result<(), plonk_core::error::Error> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::zkt_last_error(ctx)) }.to_string_lossy().into_owned();
    Err(match rc {
        2 => plonk_core::error::Error::InvalidEvalDomainSize { log_size_of_group: 0, adicity: 0 },
        8 => plonk_core::error::Error::ElementNotIndexedInTable,
        _ => plonk_core::error::Error::PCError { error: msg },
    })
}

/// `D: EvaluationDomain<F> + EvaluationDomainExt<F>` (prove.rs:70, util.rs:27-59): constants stay arkworks', the four
/// transforms of util.rs:63-140 run on the GPU.  Satisfies the bound of `prove` / `ZKTPlonk` as it stands: both traits
/// are implemented below.
#[derive(Copy, Clone, Hash, Eq, PartialEq, Debug, ark_serialize::CanonicalSerialize, ark_serialize::CanonicalDeserialize)]
pub struct GpuDomain<F: FftField> {
    pub inner: Radix2EvaluationDomain<F>,
}

impl<F: FftField + PrimeField> GpuDomain<F> {
    /// `DomainCoeff<F>` also admits group elements, whose layout the library does not speak: only a vector of the scalar
    /// field itself (4 limbs per element, arkworks' Montgomery form) goes to the device, anything else stays on ark-poly.
    /// `fft_in_place<T: DomainCoeff<F>>` cannot ask for `T: 'static`, so `core::any::TypeId` is out; `typeid::of` (the
    /// `typeid` crate, sound for non-'static types) decides T == F exactly -- no name comparison, no size heuristics.
    fn is_scalar_vec<T>() -> bool {
        typeid::of::<T>() == typeid::of::<F>()
    }

    /// inverse / coset as in include/zkt_plonk.h zkt_ntt; ark-poly resizes to n, the library zero-pads
    fn run<T: DomainCoeff<F>>(&self, v: &mut Vec<T>, inverse: c_int, coset: c_int) {
        assert!(Self::is_scalar_vec::<T>(), "GpuDomain::run is for vectors of the scalar field itself");
        let n = self.inner.size();
        let in_len = v.len();
        v.resize(n, T::zero());
        with_ctx::<F, _>(|ctx| unsafe {
            let rc = ffi::zkt_ntt(ctx, self.inner.log_size_of_group as c_int, inverse, coset,
                                  v.as_ptr() as *const u64, in_len, v.as_mut_ptr() as *mut u64);
            check(ctx, rc).expect("zkt_ntt");
        });
    }
}

impl<F: FftField + PrimeField> EvaluationDomain<F> for GpuDomain<F> {
    type Elements = <Radix2EvaluationDomain<F> as EvaluationDomain<F>>::Elements;

    fn new(num_coeffs: usize) -> Option<Self> { Radix2EvaluationDomain::new(num_coeffs).map(|inner| Self { inner }) }
    fn compute_size_of_domain(num_coeffs: usize) -> Option<usize> { Radix2EvaluationDomain::<F>::compute_size_of_domain(num_coeffs) }
    fn size(&self) -> usize { self.inner.size() }
    fn fft_in_place<T: DomainCoeff<F>>(&self, c: &mut Vec<T>) {                                   // util.rs:104-113
        if Self::is_scalar_vec::<T>() { self.run(c, 0, 0) } else { self.inner.fft_in_place(c) }
    }
    fn ifft_in_place<T: DomainCoeff<F>>(&self, e: &mut Vec<T>) {                                  // util.rs:63-86
        if Self::is_scalar_vec::<T>() { self.run(e, 1, 0) } else { self.inner.ifft_in_place(e) }
    }
    fn coset_fft_in_place<T: DomainCoeff<F>>(&self, c: &mut Vec<T>) {                             // util.rs:117-140
        if Self::is_scalar_vec::<T>() { self.run(c, 0, 1) } else { self.inner.coset_fft_in_place(c) }
    }
    fn coset_ifft_in_place<T: DomainCoeff<F>>(&self, e: &mut Vec<T>) {                            // util.rs:90-100
        if Self::is_scalar_vec::<T>() { self.run(e, 1, 1) } else { self.inner.coset_ifft_in_place(e) }
    }
    fn evaluate_all_lagrange_coefficients(&self, tau: F) -> Vec<F> { self.inner.evaluate_all_lagrange_coefficients(tau) }
    fn vanishing_polynomial(&self) -> ark_poly::univariate::SparsePolynomial<F> { self.inner.vanishing_polynomial() }
    fn evaluate_vanishing_polynomial(&self, tau: F) -> F { self.inner.evaluate_vanishing_polynomial(tau) }
    fn element(&self, i: usize) -> F { self.inner.element(i) }
    fn elements(&self) -> Self::Elements { self.inner.elements() }
}

impl<F: FftField + PrimeField> plonk_core::util::EvaluationDomainExt<F> for GpuDomain<F> {      // util.rs:27-59
    fn log_size_of_group(&self) -> u32 { self.inner.log_size_of_group }
    fn group_gen(&self) -> F { self.inner.group_gen }
}

/// Repacks `GroupAffine { x, y, infinity }` (repr(Rust)) into x limbs || y limbs, (0, 0) for the identity (generic form;
/// kzg::load_committer_key does the same through G1FromXY).
pub fn pack_g1<G: ark_ec::AffineCurve>(pts: &[G], limbs: usize, xy: impl Fn(&G) -> (Vec<u64>, Vec<u64>)) -> Vec<u64> {
    let mut out = vec![0u64; pts.len() * 2 * limbs];
    for (i, p) in pts.iter().enumerate() {
        if p.is_zero() { continue; }
        let (x, y) = xy(p);
        out[2 * limbs * i..2 * limbs * i + limbs].copy_from_slice(&x);
        out[2 * limbs * i + limbs..2 * limbs * (i + 1)].copy_from_slice(&y);
    }
    out
}

pub mod kzg;
pub mod prover;
