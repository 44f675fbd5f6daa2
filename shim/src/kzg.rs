//! `PC: HomomorphicCommitment<F>` (plonk-core/src/commitment.rs:10-46) on the GPU: `GpuKZG10<E>` has every associated type
//! of `SonicKZG10<E, DensePolynomial<E::Fr>>` (keys, commitments and proofs serialise identically; `setup`, `trim` and the
//! checks are arkworks'), `commit` and `open` run their multi-scalar multiplications through `zkt_msm_g1`.
//!
//! UNCOMPILED (no Rust toolchain in the authoring image, see lib.rs): written against ark-ec / ark-poly-commit 0.3.0 as
//! recalled -- in particular the trait's required items (`open_individual_opening_challenges`,
//! `check_individual_opening_challenges`) and `GroupAffine::new(x, y, infinity)`.  Nothing here is a stub: every function
//! has a body a compiler can be pointed at; expect signature-level adjustments, not missing logic.
use crate::{check, ffi, with_ctx};
use ark_ec::{models::SWModelParameters, short_weierstrass_jacobian::GroupAffine, AffineCurve, PairingEngine};
use ark_ff::{FftField, PrimeField, Zero};
use ark_poly::{univariate::DensePolynomial, UVPolynomial};
use ark_poly_commit::{
    kzg10, sonic_pc::SonicKZG10, LabeledCommitment, LabeledPolynomial, PCRandomness, PolynomialCommitment, QuerySet,
    Evaluations,
};
use ark_std::rand::RngCore;
use core::marker::PhantomData;
use plonk_core::commitment::HomomorphicCommitment;
use std::os::raw::c_int;

type Poly<E> = DensePolynomial<<E as PairingEngine>::Fr>;
type Sonic<E> = SonicKZG10<E, Poly<E>>;
type SonicPC<E> = <E as PairingEngine>::Fr;   // (readability of the bounds below)

pub struct GpuKZG10<E: PairingEngine>(PhantomData<E>);

/// Affine G1 points the library's (x, y) limb pairs can be turned into.  Both curves of the reference (ark-bn254,
/// ark-bls12-381) have `G1Affine = GroupAffine<P>` with a short Weierstrass `P`, which is the one impl below; the bound
/// `E::G1Affine: G1FromXY` on `GpuKZG10` is therefore satisfied by exactly the engines `bin/src/instance.rs:7-15` names.
pub trait G1FromXY: AffineCurve {
    fn from_xy(x: Self::BaseField, y: Self::BaseField) -> Self;
    fn xy(&self) -> (Self::BaseField, Self::BaseField);
}
impl<P: SWModelParameters> G1FromXY for GroupAffine<P> {
    fn from_xy(x: P::BaseField, y: P::BaseField) -> Self { GroupAffine::new(x, y, false) }
    fn xy(&self) -> (P::BaseField, P::BaseField) { (self.x, self.y) }
}

/// `Fp256` / `Fp384` are `struct Fp<P>(pub BigInteger256|384, PhantomData<P>)` and the big integer is `[u64; N]`: the
/// in-memory value IS the Montgomery limbs the C-ABI speaks (NOT `into_repr()`, which leaves Montgomery form).
fn fq_limbs<Fq: PrimeField>(x: &Fq) -> &[u64] {
    debug_assert_eq!(core::mem::size_of::<Fq>(), 8 * Fq::BigInt::NUM_LIMBS);
    unsafe { core::slice::from_raw_parts(x as *const Fq as *const u64, Fq::BigInt::NUM_LIMBS) }
}
/// The inverse: a field element whose in-memory limbs are `limbs` (already reduced, Montgomery form: what the library returns).
fn fq_from_limbs<Fq: PrimeField>(limbs: &[u64]) -> Fq {
    assert_eq!(limbs.len(), Fq::BigInt::NUM_LIMBS);
    let mut x = Fq::zero();
    unsafe { core::ptr::copy_nonoverlapping(limbs.as_ptr(), &mut x as *mut Fq as *mut u64, limbs.len()) };
    x
}

/// (x limbs || y limbs, infinity flag) as written by `zkt_msm_g1` / the transcript callbacks -> `E::G1Affine`.
pub fn g1_from_limbs<G: G1FromXY>(xy: &[u64], infinity: bool) -> G
where
    G::BaseField: PrimeField,
{
    if infinity {
        return G::zero();
    }
    let l = <G::BaseField as PrimeField>::BigInt::NUM_LIMBS;
    G::from_xy(fq_from_limbs(&xy[..l]), fq_from_limbs(&xy[l..2 * l]))
}

/// `ck.powers_of_g[..count]` -> the library (once per key: the window table is built on the device and stays there).
/// The prover never commits to more than n + 7 coefficients, so `count = n + 8` of the 4n + 1 powers is enough.
/// `GroupAffine { x, y, infinity }` is `repr(Rust)`: repacked explicitly, never transmuted.
pub fn load_committer_key<E: PairingEngine>(ck: &<Sonic<E> as PolynomialCommitment<E::Fr, Poly<E>>>::CommitterKey, count: usize)
where
    E::Fr: FftField,
    E::G1Affine: G1FromXY,
    <E::G1Affine as AffineCurve>::BaseField: PrimeField,
{
    let l = <<E::G1Affine as AffineCurve>::BaseField as PrimeField>::BigInt::NUM_LIMBS;
    let mut xy = vec![0u64; count * 2 * l];
    for (i, p) in ck.powers_of_g[..count].iter().enumerate() {
        if p.is_zero() {
            continue;   // (0, 0) is the library's identity
        }
        let (x, y) = p.xy();
        xy[2 * l * i..2 * l * i + l].copy_from_slice(fq_limbs(&x));
        xy[2 * l * i + l..2 * l * (i + 1)].copy_from_slice(fq_limbs(&y));
    }
    with_ctx::<E::Fr, _>(|ctx| check(ctx, unsafe { ffi::zkt_srs_load(ctx, xy.as_ptr(), count) }).expect("zkt_srs_load"));
}

impl<E: PairingEngine> GpuKZG10<E>
where
    E::Fr: FftField,
    E::G1Affine: G1FromXY,
    <E::G1Affine as AffineCurve>::BaseField: PrimeField,
{
    /// kzg10::commit without hiding = MSM(powers_of_g[..len], coeffs); the scalars go over as they lie in memory
    /// (Montgomery form, `montgomery = 1`): `into_repr()` is the device's business.
    fn commit_one(coeffs: &[E::Fr]) -> Result<kzg10::Commitment<E>, ark_poly_commit::Error> {
        let mut xy = [0u64; 12];   // 2 x 6 limbs covers Fq381
        let mut inf: c_int = 0;
        with_ctx::<E::Fr, _>(|ctx| {
            check(ctx, unsafe { ffi::zkt_msm_g1(ctx, coeffs.as_ptr() as *const u64, coeffs.len(), 0, 1, xy.as_mut_ptr(), &mut inf) })
        })
        .map_err(|_| ark_poly_commit::Error::TooManyCoefficients { num_coefficients: coeffs.len(), num_powers: 0 })?;
        Ok(kzg10::Commitment(g1_from_limbs::<E::G1Affine>(&xy, inf != 0)))
    }
}

impl<E: PairingEngine> PolynomialCommitment<E::Fr, Poly<E>> for GpuKZG10<E>
where
    E::Fr: FftField,
    E::G1Affine: G1FromXY,
    <E::G1Affine as AffineCurve>::BaseField: PrimeField,
{
    type UniversalParams = <Sonic<E> as PolynomialCommitment<E::Fr, Poly<E>>>::UniversalParams;
    type CommitterKey = <Sonic<E> as PolynomialCommitment<E::Fr, Poly<E>>>::CommitterKey;
    type VerifierKey = <Sonic<E> as PolynomialCommitment<E::Fr, Poly<E>>>::VerifierKey;
    type PreparedVerifierKey = <Sonic<E> as PolynomialCommitment<E::Fr, Poly<E>>>::PreparedVerifierKey;
    type Commitment = <Sonic<E> as PolynomialCommitment<E::Fr, Poly<E>>>::Commitment;            // kzg10::Commitment<E>
    type PreparedCommitment = <Sonic<E> as PolynomialCommitment<E::Fr, Poly<E>>>::PreparedCommitment;
    type Randomness = <Sonic<E> as PolynomialCommitment<E::Fr, Poly<E>>>::Randomness;
    type Proof = <Sonic<E> as PolynomialCommitment<E::Fr, Poly<E>>>::Proof;                      // kzg10::Proof<E>
    type BatchProof = <Sonic<E> as PolynomialCommitment<E::Fr, Poly<E>>>::BatchProof;
    type Error = <Sonic<E> as PolynomialCommitment<E::Fr, Poly<E>>>::Error;

    fn setup<R: RngCore>(max_degree: usize, num_vars: Option<usize>, rng: &mut R) -> Result<Self::UniversalParams, Self::Error> {
        Sonic::<E>::setup(max_degree, num_vars, rng)
    }

    fn trim(pp: &Self::UniversalParams, supported_degree: usize, supported_hiding_bound: usize,
            enforced_degree_bounds: Option<&[usize]>) -> Result<(Self::CommitterKey, Self::VerifierKey), Self::Error> {
        Sonic::<E>::trim(pp, supported_degree, supported_hiding_bound, enforced_degree_bounds)
    }

    fn commit<'a>(
        ck: &Self::CommitterKey,
        polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<E::Fr, Poly<E>>>,
        _rng: Option<&mut dyn RngCore>,
    ) -> Result<(Vec<LabeledCommitment<Self::Commitment>>, Vec<Self::Randomness>), Self::Error>
    where
        Poly<E>: 'a,
    {
        let _ = ck;   // on the device since load_committer_key
        let mut commits = Vec::new();
        let mut rands = Vec::new();
        for p in polynomials {
            // plonk-core commits without degree bounds or hiding (prove.rs:133-135,178-180,249-251,306-308,373-375);
            // anything else is not this path's business
            assert!(p.degree_bound().is_none() && p.hiding_bound().is_none(), "GpuKZG10: plain commitments only");
            commits.push(LabeledCommitment::new(p.label().clone(), Self::commit_one(p.polynomial().coeffs())?, None));
            rands.push(Self::Randomness::empty());
        }
        Ok((commits, rands))
    }

    /// `PC::open(ck, polys, comms, &point, eta, rands, None)` (prove.rs:381-420, 427-451) is the trait's provided wrapper
    /// around this with `opening_challenges = |k| eta^k`: combined = sum_k eta^k p_k, witness = combined / (X - z),
    /// `kzg10::Proof { w: commit(witness), random_v: None }`.
    fn open_individual_opening_challenges<'a>(
        ck: &Self::CommitterKey,
        labeled_polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<E::Fr, Poly<E>>>,
        _commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>,
        point: &'a E::Fr,
        opening_challenges: &dyn Fn(u64) -> E::Fr,
        _rands: impl IntoIterator<Item = &'a Self::Randomness>,
        _rng: Option<&mut dyn RngCore>,
    ) -> Result<Self::Proof, Self::Error>
    where
        Self::Randomness: 'a,
        Self::Commitment: 'a,
        Poly<E>: 'a,
    {
        let _ = ck;
        let mut combined = Poly::<E>::zero();
        for (k, p) in labeled_polynomials.into_iter().enumerate() {
            combined += (opening_challenges(k as u64), p.polynomial());
        }
        let divisor = Poly::<E>::from_coefficients_vec(vec![-*point, E::Fr::from(1u64)]);
        let witness = &combined / &divisor;
        Ok(kzg10::Proof { w: Self::commit_one(&witness.coeffs)?.0, random_v: None })
    }

    fn check_individual_opening_challenges<'a>(
        vk: &Self::VerifierKey,
        commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>,
        point: &'a E::Fr,
        values: impl IntoIterator<Item = E::Fr>,
        proof: &Self::Proof,
        opening_challenges: &dyn Fn(u64) -> E::Fr,
        rng: Option<&mut dyn RngCore>,
    ) -> Result<bool, Self::Error>
    where
        Self::Commitment: 'a,
    {
        Sonic::<E>::check_individual_opening_challenges(vk, commitments, point, values, proof, opening_challenges, rng)
    }

    fn batch_check_individual_opening_challenges<'a, R: RngCore>(
        vk: &Self::VerifierKey,
        commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>,
        query_set: &QuerySet<E::Fr>,
        values: &Evaluations<E::Fr, E::Fr>,
        proof: &Self::BatchProof,
        opening_challenges: &dyn Fn(u64) -> E::Fr,
        rng: &mut R,
    ) -> Result<bool, Self::Error>
    where
        Self::Commitment: 'a,
    {
        Sonic::<E>::batch_check_individual_opening_challenges(vk, commitments, query_set, values, proof, opening_challenges, rng)
    }
}

impl<E: PairingEngine> HomomorphicCommitment<E::Fr> for GpuKZG10<E>
where
    E::Fr: FftField,
    E::G1Affine: G1FromXY,
    <E::G1Affine as AffineCurve>::BaseField: PrimeField,
{
    /// commitment.rs:32-45: 13 arbitrary points, verifier only; stays on arkworks (`zkt_g1_msm_host` is there for a host
    /// without it -- a device launch would cost more than the sum)
    fn multi_scalar_mul(commitments: &[Self::Commitment], scalars: &[E::Fr]) -> Self::Commitment {
        <Sonic<E> as HomomorphicCommitment<E::Fr>>::multi_scalar_mul(commitments, scalars)
    }
}

/// The transcript callbacks' way from limbs to `PC::Commitment` (prover.rs): for `GpuKZG10<E>` and `KZG10<E>` alike the
/// commitment is `kzg10::Commitment<E>`.
pub fn commitment_from_limbs<E: PairingEngine>(xy: *const u64, infinity: c_int) -> kzg10::Commitment<E>
where
    E::G1Affine: G1FromXY,
    <E::G1Affine as AffineCurve>::BaseField: PrimeField,
{
    let l = <<E::G1Affine as AffineCurve>::BaseField as PrimeField>::BigInt::NUM_LIMBS;
    let s = unsafe { core::slice::from_raw_parts(xy, 2 * l) };
    kzg10::Commitment(g1_from_limbs::<E::G1Affine>(s, infinity != 0))
}

#[allow(dead_code)]
type _Unused<E> = SonicPC<E>;
