// lib.rs -- the reference crate's public API (gogoex/BulletProofsPlus src/lib.rs:11-13: PublicKey, RangeProof,
// RangeProver) over libbpp_amd.so, plus the README's RangeVerifier (README.md:47-55) and the MulVec seam
// (src/bls12_381/building_block/mulvec.rs:7-53).  Same names, argument meaning and error behaviour as the reference:
// `verify` returns Err(ProofError::VerificationError), `prove` and `MulVec::calculate` panic on bad lengths.
//
// SOURCE ONLY: the image this repository is built in has no Rust toolchain, so this file has never been compiled.
// The `extern "C"` block (ffi.rs) is generated from include/bpp_amd.h and checked against it by
// tests/test_rust_binding.py, which also checks that every ffi function called below exists with the arity used here.
#![allow(non_snake_case)]
pub mod ffi;

use std::os::raw::c_int;
use std::sync::Once;

/// u64 limbs per base-field element of BLS12-381 (the curve the reference's range proof is wired to,
/// src/range/mod.rs:10-15); a wire point is x | y | infinity flag.
pub const L: usize = 6;
pub const PW: usize = 2 * L + 1;

/// reference src/errors.rs:14-50 (only these two are ever produced on this path)
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum ProofError {
    VerificationError,
    FormatError,
}

/// reference src/bls12_381/building_block/point/point.rs:12 -- here the canonical affine wire image
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub struct Point(pub [u64; PW]);
/// reference src/bls12_381/building_block/scalar/prime_field_elem.rs:13-15 -- canonical little-endian limbs
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub struct PrimeFieldElem(pub [u64; 4]);

impl Point {
    pub fn zero() -> Point {
        let mut w = [0u64; PW];
        w[2 * L] = 1;
        Point(w)
    }
    pub fn is_zero(&self) -> bool {
        self.0[2 * L] != 0
    }
}
impl PrimeFieldElem {
    /// PrimeFieldElem::new(i32) for non-negative values (prime_field_elem.rs:191-195)
    pub fn new(n: u32) -> PrimeFieldElem {
        PrimeFieldElem([n as u64, 0, 0, 0])
    }
}

static INIT: Once = Once::new();
static mut CTX: *mut ffi::BppCtx = std::ptr::null_mut();

/// reference src/bls12_381/building_block/arith.rs:6-19
pub struct Arith;
impl Arith {
    pub fn init() {
        INIT.call_once(|| unsafe {
            let mut c: *mut ffi::BppCtx = std::ptr::null_mut();
            let rc = ffi::bpp_init(ffi::BPP_BLS12_381_G1, 0, &mut c);
            if rc != 0 {
                panic!("bpp_init failed: {}", rc);
            }
            CTX = c;
        });
    }
}
fn ctx() -> *mut ffi::BppCtx {
    Arith::init();
    unsafe { CTX }
}
fn flat_points(ps: &[Point]) -> Vec<u64> {
    ps.iter().flat_map(|p| p.0.iter().cloned()).collect()
}
fn flat_scalars(ss: &[PrimeFieldElem]) -> Vec<u64> {
    ss.iter().flat_map(|s| s.0.iter().cloned()).collect()
}
fn unflat_points(w: &[u64]) -> Vec<Point> {
    w.chunks(PW).map(|c| {
        let mut a = [0u64; PW];
        a.copy_from_slice(c);
        Point(a)
    }).collect()
}

/// reference src/bls12_381/building_block/mulvec.rs:7-53
pub struct MulVec {
    pub scalars: Vec<PrimeFieldElem>,
    pub points: Vec<Point>,
}
impl MulVec {
    pub fn new() -> MulVec {
        MulVec { scalars: vec![], points: vec![] }
    }
    pub fn add_scalar(&mut self, s: &PrimeFieldElem) {
        self.scalars.push(*s);
    }
    pub fn add_scalars(&mut self, ss: &[PrimeFieldElem]) {
        self.scalars.extend_from_slice(ss);
    }
    pub fn add_point(&mut self, p: &Point) {
        self.points.push(*p);
    }
    pub fn add_points(&mut self, ps: &[Point]) {
        self.points.extend_from_slice(ps);
    }
    pub fn calculate(&self) -> Point {
        if self.scalars.len() != self.points.len() {
            panic!("mulvec: lengths of scalars and points must match"); // mulvec.rs:23-25
        }
        let sc = flat_scalars(&self.scalars);
        let pt = flat_points(&self.points);
        let mut out = [0u64; PW];
        let rc = unsafe { ffi::bpp_msm(ctx(), sc.as_ptr(), pt.as_ptr(), self.scalars.len(), out.as_mut_ptr()) };
        assert!(rc == 0, "bpp_msm failed: {}", rc);
        Point(out)
    }
}

/// reference src/publickey.rs:13-52
pub struct PublicKey {
    pub g: Point,
    pub h: Point,
    pub G_vec: Vec<Point>,
    pub H_vec: Vec<Point>,
}
impl PublicKey {
    pub fn new(length: usize) -> PublicKey {
        let mut gh = vec![0u64; 2 * PW];
        let mut gv = vec![0u64; length.max(1) * PW];
        let mut hv = vec![0u64; length.max(1) * PW];
        let rc = unsafe { ffi::bpp_pk_new(ctx(), length, gh.as_mut_ptr(), gv.as_mut_ptr(), hv.as_mut_ptr()) };
        assert!(rc == 0, "bpp_pk_new failed: {}", rc);
        let ghp = unflat_points(&gh);
        PublicKey { g: ghp[0], h: ghp[1], G_vec: unflat_points(&gv[..length * PW]), H_vec: unflat_points(&hv[..length * PW]) }
    }
    /// generators hashed from a label instead of the reference's test generators (bpp_pk_hashed)
    pub fn from_label(length: usize, label: &[u8]) -> PublicKey {
        let mut gh = vec![0u64; 2 * PW];
        let mut gv = vec![0u64; length.max(1) * PW];
        let mut hv = vec![0u64; length.max(1) * PW];
        let rc = unsafe {
            ffi::bpp_pk_hashed(ctx(), label.as_ptr(), label.len(), length, gh.as_mut_ptr(), gv.as_mut_ptr(), hv.as_mut_ptr())
        };
        assert!(rc == 0, "bpp_pk_hashed failed: {}", rc);
        let ghp = unflat_points(&gh);
        PublicKey { g: ghp[0], h: ghp[1], G_vec: unflat_points(&gv[..length * PW]), H_vec: unflat_points(&hv[..length * PW]) }
    }
    /// g * v + h * gamma (publickey.rs:50-52)
    pub fn commitment(&self, v: u64, gamma: &PrimeFieldElem) -> Point {
        let gh = flat_points(&[self.g, self.h]);
        let mut out = [0u64; PW];
        let rc = unsafe { ffi::bpp_commit(ctx(), gh.as_ptr(), v, gamma.0.as_ptr(), out.as_mut_ptr()) };
        assert!(rc == 0, "bpp_commit failed: {}", rc);
        Point(out)
    }
}

/// reference src/range/prover.rs:13-42
pub struct RangeProver {
    pub v_vec: Vec<u64>,
    pub gamma_vec: Vec<PrimeFieldElem>,
    pub commitment_vec: Vec<Point>,
}
impl RangeProver {
    pub fn new() -> RangeProver {
        RangeProver { v_vec: vec![], gamma_vec: vec![], commitment_vec: vec![] }
    }
    pub fn commit(&mut self, pk: &PublicKey, v: u64, gamma: PrimeFieldElem) {
        self.v_vec.push(v);
        self.gamma_vec.push(gamma);
        self.commitment_vec.push(pk.commitment(v, &gamma)); // keeps the `v as i32` truncation of prover.rs:37
    }
}

/// README.md:47-55 (the reference's code has no such type; `verify` takes the slice, src/range/mod.rs:57-62)
pub struct RangeVerifier {
    pub commitment_vec: Vec<Point>,
}
impl RangeVerifier {
    pub fn new() -> RangeVerifier {
        RangeVerifier { commitment_vec: vec![] }
    }
    pub fn allocate(&mut self, commitment_vec: &[Point]) {
        self.commitment_vec = commitment_vec.to_vec();
    }
}

/// reference src/weighted_inner_product_proof.rs:25-33
pub struct WeightedInnerProductProof {
    pub L_vec: Vec<Point>,
    pub R_vec: Vec<Point>,
    pub A: Point,
    pub B: Point,
    pub r_prime: PrimeFieldElem,
    pub s_prime: PrimeFieldElem,
    pub d_prime: PrimeFieldElem,
}

/// reference src/range/mod.rs:25-28
pub struct RangeProof {
    pub A: Point,
    pub proof: WeightedInnerProductProof,
}
impl RangeProof {
    /// reference src/range/mod.rs:31-55
    pub fn prove(pk: &PublicKey, n: usize, prover: &RangeProver) -> RangeProof {
        let m = prover.v_vec.len();
        let mn = n * m;
        assert!(mn.is_power_of_two());                          // wip.rs:67
        assert_eq!(pk.G_vec.len(), mn);                         // range/mod.rs:90-91, :252-253
        assert_eq!(pk.H_vec.len(), mn);
        let k = mn.trailing_zeros() as usize;
        let gh = flat_points(&[pk.g, pk.h]);
        let (gv, hv) = (flat_points(&pk.G_vec), flat_points(&pk.H_vec));
        let gam = flat_scalars(&prover.gamma_vec);
        let cv = flat_points(&prover.commitment_vec);
        let mut pts = vec![0u64; (3 + 2 * k) * PW];
        let mut sc = vec![0u64; 12];
        let rc = unsafe {
            ffi::bpp_range_prove(ctx(), gh.as_ptr(), gv.as_ptr(), hv.as_ptr(), n, m, prover.v_vec.as_ptr(), gam.as_ptr(),
                                 cv.as_ptr(), pts.as_mut_ptr(), sc.as_mut_ptr())
        };
        assert!(rc == 0, "bpp_range_prove failed: {}", rc);
        let p = unflat_points(&pts);
        let s = |i: usize| PrimeFieldElem([sc[4 * i], sc[4 * i + 1], sc[4 * i + 2], sc[4 * i + 3]]);
        RangeProof {
            A: p[0],
            proof: WeightedInnerProductProof { A: p[1], B: p[2], L_vec: p[3..3 + k].to_vec(), R_vec: p[3 + k..3 + 2 * k].to_vec(),
                                               r_prime: s(0), s_prime: s(1), d_prime: s(2) },
        }
    }
    /// reference src/range/mod.rs:57-78
    pub fn verify(&self, pk: &PublicKey, n: usize, commitment_vec: &[Point]) -> Result<(), ProofError> {
        let k = self.proof.L_vec.len();
        let mut pts = vec![self.A, self.proof.A, self.proof.B];
        pts.extend_from_slice(&self.proof.L_vec);
        pts.extend_from_slice(&self.proof.R_vec);
        let pw = flat_points(&pts);
        let sc = flat_scalars(&[self.proof.r_prime, self.proof.s_prime, self.proof.d_prime]);
        let gh = flat_points(&[pk.g, pk.h]);
        let (gv, hv) = (flat_points(&pk.G_vec), flat_points(&pk.H_vec));
        let cv = flat_points(commitment_vec);
        let rc: c_int = unsafe {
            ffi::bpp_range_verify(ctx(), gh.as_ptr(), gv.as_ptr(), hv.as_ptr(), n, commitment_vec.len(), pw.as_ptr(), k,
                                  sc.as_ptr(), cv.as_ptr())
        };
        match rc {
            0 => Ok(()),
            1 => Err(ProofError::VerificationError),
            2 => Err(ProofError::FormatError),
            e => panic!("bpp_range_verify: {}", e),
        }
    }
    /// the README's calling convention (README.md:55)
    pub fn verify_with(&self, pk: &PublicKey, n: usize, verifier: &RangeVerifier) -> Result<(), ProofError> {
        self.verify(pk, n, &verifier.commitment_vec)
    }
}

/// Not in the reference (it verifies one proof at a time, src/range/mod.rs:57-78): the engine's batch verifier for ONE
/// (public key, n, m) -- window tables of the 2mn + 2 generators in HBM, built once; `verify_batch` then judges any number of
/// proofs per call and returns the reference's verdict for each of them (include/bpp_amd.h: bpp_verifier_create,
/// bpp_range_verify_batch).  `window_bits` = 0 lets the engine choose; 13 needs ~15 GB at (64,16), 17 ~204 GB.
pub struct BatchVerifier {
    handle: *mut ffi::BppVerifier,
    m: usize,
    k: usize,
}
impl BatchVerifier {
    pub fn new(pk: &PublicKey, n: usize, m: usize, window_bits: i32) -> BatchVerifier {
        assert!(pk.G_vec.len() == n * m && pk.H_vec.len() == n * m, "public key of length n*m");
        let gh = flat_points(&[pk.g, pk.h]);
        let (gv, hv) = (flat_points(&pk.G_vec), flat_points(&pk.H_vec));
        let mut h: *mut ffi::BppVerifier = std::ptr::null_mut();
        let rc = unsafe { ffi::bpp_verifier_create(ctx(), gh.as_ptr(), gv.as_ptr(), hv.as_ptr(), n, m, window_bits as c_int, &mut h) };
        assert!(rc == 0, "bpp_verifier_create: {}", rc);
        BatchVerifier { handle: h, m, k: (n * m).trailing_zeros() as usize }
    }
    /// one `Result` per (proof, its commitments), in order: Ok / Err(VerificationError), as RangeProof::verify would return
    pub fn verify_batch(&self, batch: &[(&RangeProof, &[Point])]) -> Vec<Result<(), ProofError>> {
        let mut pts: Vec<Point> = Vec::with_capacity(batch.len() * (3 + 2 * self.k + self.m));
        let mut scs: Vec<PrimeFieldElem> = Vec::with_capacity(batch.len() * 3);
        for (proof, commitment_vec) in batch {
            assert!(proof.proof.L_vec.len() == self.k && commitment_vec.len() == self.m, "proof of another shape");
            pts.extend_from_slice(&[proof.A, proof.proof.A, proof.proof.B]);
            pts.extend_from_slice(&proof.proof.L_vec);
            pts.extend_from_slice(&proof.proof.R_vec);
            pts.extend_from_slice(commitment_vec);
            scs.extend_from_slice(&[proof.proof.r_prime, proof.proof.s_prime, proof.proof.d_prime]);
        }
        let (pw, sw) = (flat_points(&pts), flat_scalars(&scs));
        let mut ok = vec![0u32; batch.len()];
        let rc = unsafe { ffi::bpp_range_verify_batch(self.handle, pw.as_ptr(), sw.as_ptr(), batch.len(), ok.as_mut_ptr()) };
        assert!(rc == 0, "bpp_range_verify_batch: {}", rc);
        ok.iter().map(|&v| if v == 0 { Ok(()) } else { Err(ProofError::VerificationError) }).collect()
    }
}
impl Drop for BatchVerifier {
    fn drop(&mut self) {
        unsafe { ffi::bpp_verifier_destroy(self.handle) };
    }
}
