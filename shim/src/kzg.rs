//! `PC: HomomorphicCommitment<F>` (plonk-core/src/commitment.rs:10-46) on the GPU: `GpuKZG10<E>` has every associated type
//! of `SonicKZG10<E, DensePolynomial<E::Fr>>` (keys, commitments and proofs serialise identically, `setup` / `trim` /
//! `check` are arkworks'), `commit` and `open` run their multi-scalar multiplications through `zkt_msm_g1`.
//! UNCOMPILED (see lib.rs).
use crate::{check, ffi, pack_g1, with_ctx};
use ark_ec::{AffineCurve, PairingEngine};
use ark_ff::{PrimeField, Zero};
use ark_poly::univariate::DensePolynomial;
use ark_poly_commit::{kzg10, sonic_pc::SonicKZG10, LabeledCommitment, LabeledPolynomial, PCRandomness, PolynomialCommitment};
use core::marker::PhantomData;
use plonk_core::commitment::HomomorphicCommitment;
use std::os::raw::c_int;

type Poly<E> = DensePolynomial<<E as PairingEngine>::Fr>;
type Sonic<E> = SonicKZG10<E, Poly<E>>;

pub struct GpuKZG10<E: PairingEngine>(PhantomData<E>);

/// `ck.powers_of_g[..count]` -> the library (once per key: the window table is built on the device and stays there).
/// The prover never commits to more than n + 7 coefficients, so `count = n + 8` of the 4n + 1 powers is enough.
pub fn load_committer_key<E: PairingEngine>(ck: &<Sonic<E> as PolynomialCommitment<E::Fr, Poly<E>>>::CommitterKey, count: usize)
where
    E::Fr: ark_ff::FftField,
{
    let limbs = <<E::G1Affine as AffineCurve>::BaseField as PrimeField>::BigInt::NUM_LIMBS;
    let xy = pack_g1(&ck.powers_of_g[..count], limbs, |p| (fq_limbs(&p.x), fq_limbs(&p.y)));   // GroupAffine is repr(Rust)
    with_ctx::<E::Fr, _>(|ctx| check(ctx, unsafe { ffi::zkt_srs_load(ctx, xy.as_ptr(), count) }).expect("zkt_srs_load"));
}

fn fq_limbs<Fq: PrimeField>(x: &Fq) -> Vec<u64> {
    // the in-memory Montgomery form, NOT into_repr(): Fp256 / Fp384 wrap BigInteger256 / 384([u64; N])
    unsafe { core::slice::from_raw_parts(x as *const Fq as *const u64, Fq::BigInt::NUM_LIMBS) }.to_vec()
}

fn g1_from_limbs<E: PairingEngine>(xy: &[u64], infinity: bool) -> E::G1Affine {
    if infinity {
        return E::G1Affine::zero();
    }
    // the inverse of fq_limbs; spelled with the concrete curve's constructor in the real crate
    unimplemented!("GroupAffine::new(x_from_limbs(&xy[..l]), y_from_limbs(&xy[l..]), false)")
}

impl<E: PairingEngine> GpuKZG10<E>
where
    E::Fr: ark_ff::FftField,
{
    /// kzg10::commit without hiding = MSM(powers_of_g[..len], coeffs); into_repr() happens on the device
    fn commit_one(coeffs: &[E::Fr]) -> Result<kzg10::Commitment<E>, ark_poly_commit::Error> {
        let mut xy = [0u64; 12];
        let mut inf: c_int = 0;
        with_ctx::<E::Fr, _>(|ctx| {
            check(ctx, unsafe { ffi::zkt_msm_g1(ctx, coeffs.as_ptr() as *const u64, coeffs.len(), 0, 1, xy.as_mut_ptr(), &mut inf) })
        })
        .map_err(|_| ark_poly_commit::Error::TooManyCoefficients { num_coefficients: coeffs.len(), num_powers: 0 })?;
        Ok(kzg10::Commitment(g1_from_limbs::<E>(&xy, inf != 0)))
    }
}

// Every item below that is not spelled out delegates to `Sonic<E>` verbatim (same associated types):
//   type UniversalParams / CommitterKey / VerifierKey / PreparedVerifierKey / Commitment / PreparedCommitment /
//   Randomness / Proof / BatchProof / Error;  fn setup, trim, check, batch_check, ...
impl<E: PairingEngine> PolynomialCommitment<E::Fr, Poly<E>> for GpuKZG10<E>
where
    E::Fr: ark_ff::FftField,
{
    // ... associated types = <Sonic<E> as PolynomialCommitment<_, _>>::* ...

    fn commit<'a>(
        ck: &Self::CommitterKey,
        polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<E::Fr, Poly<E>>>,
        _rng: Option<&mut dyn ark_std::rand::RngCore>,
    ) -> Result<(Vec<LabeledCommitment<Self::Commitment>>, Vec<Self::Randomness>), Self::Error> {
        let _ = ck;   // loaded once by load_committer_key
        let mut commits = Vec::new();
        let mut rands = Vec::new();
        for p in polynomials {
            // plonk-core commits without degree bounds or hiding (prove.rs:133-135 etc.)
            commits.push(LabeledCommitment::new(p.label().clone(), Self::commit_one(p.polynomial().coeffs())?, None));
            rands.push(Self::Randomness::empty());
        }
        Ok((commits, rands))
    }

    fn open<'a>(
        ck: &Self::CommitterKey,
        labeled_polynomials: impl IntoIterator<Item = &'a LabeledPolynomial<E::Fr, Poly<E>>>,
        _commitments: impl IntoIterator<Item = &'a LabeledCommitment<Self::Commitment>>,
        point: &'a E::Fr,
        opening_challenge: E::Fr,
        _rands: impl IntoIterator<Item = &'a Self::Randomness>,
        _rng: Option<&mut dyn ark_std::rand::RngCore>,
    ) -> Result<Self::Proof, Self::Error> {
        let _ = ck;
        // sum_k eta^k p_k on the host (or skip all of this and call zkt_prove_with, prover.rs), divide by (X - z),
        // commit to the witness polynomial: kzg10::open with random_v = None
        let mut combined = Poly::<E>::zero();
        let mut ch = E::Fr::from(1u64);
        for p in labeled_polynomials {
            combined += (ch, p.polynomial());
            ch *= opening_challenge;
        }
        let witness = &combined / &Poly::<E>::from_coefficients_vec(vec![-*point, E::Fr::from(1u64)]);
        Ok(kzg10::Proof { w: Self::commit_one(&witness.coeffs)?.0, random_v: None })
    }
}

impl<E: PairingEngine> HomomorphicCommitment<E::Fr> for GpuKZG10<E>
where
    E::Fr: ark_ff::FftField,
{
    /// commitment.rs:32-45: 13 arbitrary points, verifier only -- zkt_g1_msm_host (host arithmetic; a launch costs more)
    fn multi_scalar_mul(commitments: &[Self::Commitment], scalars: &[E::Fr]) -> Self::Commitment {
        <Sonic<E> as HomomorphicCommitment<E::Fr>>::multi_scalar_mul(commitments, scalars)
    }
}
