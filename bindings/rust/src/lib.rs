//! Host shim: the reference's four public entry points -- `SecretKey::sign` (src/sign.rs:32-60), `PublicKey::verify`
//! (src/verify.rs:18-50), `proof_gen` (src/proof_gen.rs:78-113), `proof_verify` (src/proof_verify.rs:19-61) -- re-pointed
//! at the MI355X engine through the C ABI of include/bbs_sign_amd.h, for BATCHES of items of one issuer.
//!
//! SOURCE ONLY: the build image has no Rust toolchain; this file has never been compiled.  The ABI it binds IS
//! exercised by a non-Rust client that performs exactly this file's call sequence, argument for argument:
//! tests/cpp/ffi_sequence.c (plain C against include/bbs_sign_amd.h; tests/test_ffi_sequence.py runs it on the CPU test
//! build and on the GPU).  Every `unsafe` call below carries the step number of that file.
//!
//! Shape: a `GpuIssuer<E, F, C>` owns one engine context = (curve, generators for `l` messages, issuer key).  The
//! reference recomputes `create_generators` on every call (sign.rs:49, verify.rs:35, proof_gen.rs:98,
//! proof_verify.rs:40-43); here the library computes them once (`bbs_create_generators`) when the context is built.
//! Types are the reference's own: `Signature<E, F>` (sign.rs:18-22), `Proof<E, F>` with `challenge: Challenge<F>`
//! (proof_gen.rs:24-39), `PublicKey<E>` / `SecretKey<F>` (key_gen.rs:12-32), `SignatureError`, `ProofGenError`.
use ark_ec::{pairing::Pairing, AffineRepr, CurveGroup};
use ark_ff::{BigInteger, PrimeField};
use std::marker::PhantomData;
use std::os::raw::c_int;

use bbs_plus::constants::Constants;
use bbs_plus::key_gen::{PublicKey, SecretKey};
use bbs_plus::proof_gen::{Challenge, Proof, ProofGenError};
use bbs_plus::sign::{Signature, SignatureError};
use bbs_plus::utils::core_utilities::calculate_random_scalars;
use bbs_plus::utils::utilities_helper::FromOkm;

#[repr(C)]
pub struct BbsCtx { _p: [u8; 0] }
#[repr(C)]
pub struct BbsJob { _p: [u8; 0] }

#[link(name = "bbs_sign_amd")]
extern "C" {
    fn bbs_fp_bytes(curve: c_int) -> usize;
    fn bbs_create_generators(curve: c_int, count: usize, api_id: *const u8, api_id_len: usize, out_affine: *mut u8) -> c_int;
    fn bbs_ctx_create(curve: c_int, device_id: c_int, out: *mut *mut BbsCtx) -> c_int;
    fn bbs_ctx_destroy(ctx: *mut BbsCtx);
    fn bbs_ctx_set_window_bits(ctx: *mut BbsCtx, bits: c_int) -> c_int;
    fn bbs_ctx_set_generators(ctx: *mut BbsCtx, gens: *const u8, count: usize, api_id: *const u8, api_id_len: usize) -> c_int;
    fn bbs_ctx_set_public_key(ctx: *mut BbsCtx, pk: *const u8, is_identity: c_int) -> c_int;
    fn bbs_ctx_set_secret_key(ctx: *mut BbsCtx, sk32: *const u8) -> c_int;
    fn bbs_ctx_set_points_in_subgroup(ctx: *mut BbsCtx, vouched: c_int) -> c_int;
    fn bbs_hash_to_scalar_batch(ctx: *mut BbsCtx, n: usize, msgs: *const u8, msg_off: *const u64,
                                dst: *const u8, dst_len: usize, scalars_out: *mut u8) -> c_int;
    fn bbs_core_sign_batch(ctx: *mut BbsCtx, n: usize, messages: *const u8, msg_off: *const u64,
                           headers: *const u8, hdr_off: *const u64, signatures_out: *mut u8, status: *mut i8) -> c_int;
    fn bbs_core_verify_batch(ctx: *mut BbsCtx, n: usize, signatures: *const u8, messages: *const u8, msg_off: *const u64,
                             headers: *const u8, hdr_off: *const u64, status: *mut i8) -> c_int;
    fn bbs_core_proof_gen_batch(ctx: *mut BbsCtx, n: usize, signatures: *const u8, messages: *const u8, msg_off: *const u64,
                                disclosed_idx: *const u64, didx_off: *const u64, random_scalars: *const u8, rnd_off: *const u64,
                                headers: *const u8, hdr_off: *const u64, ph: *const u8, ph_off: *const u64,
                                proofs_fixed_out: *mut u8, commitments_out: *mut u8, commit_off_out: *mut u64, status: *mut i8) -> c_int;
    fn bbs_core_proof_verify_submit(ctx: *mut BbsCtx, n: usize, proofs_fixed: *const u8,
        commitments: *const u8, commit_off: *const u64, disclosed_msgs: *const u8, dmsg_off: *const u64,
        disclosed_idx: *const u64, didx_off: *const u64, headers: *const u8, hdr_off: *const u64,
        ph: *const u8, ph_off: *const u64, status: *mut i8, job_out: *mut *mut BbsJob) -> c_int;
    fn bbs_job_wait(job: *mut BbsJob) -> c_int;
    fn bbs_jobs_wait_any(jobs: *const *mut BbsJob, n: usize, index_out: *mut usize) -> c_int;
    fn bbs_job_free(job: *mut BbsJob);
    // items of ANY number of messages in one call (one context per count inside the library): see `GpuIssuerAnyLength`
    fn bbs_issuer_create(curve: c_int, device_id: c_int, api_id: *const u8, api_id_len: usize, out: *mut *mut BbsIssuer) -> c_int;
    fn bbs_issuer_destroy(issuer: *mut BbsIssuer);
    fn bbs_issuer_set_public_key(issuer: *mut BbsIssuer, pk: *const u8, is_identity: c_int) -> c_int;
    fn bbs_issuer_set_limits(issuer: *mut BbsIssuer, max_messages: usize, window_bits: c_int) -> c_int;
    fn bbs_issuer_proof_verify(issuer: *mut BbsIssuer, n: usize, proof_octets: *const u8, oct_off: *const u64,
        msg_bytes: *const u8, msg_byte_off: *const u64, msg_item_off: *const u64, disclosed_idx: *const u64, didx_off: *const u64,
        headers: *const u8, hdr_off: *const u64, ph: *const u8, ph_off: *const u64, status: *mut i8) -> c_int;
    // a list of proofs over SEVERAL GPUs from one process (the multi-GPU fan-out is inside the library): see `GpuPool`
    fn bbs_pool_create(device_ids: *const c_int, n_devices: usize, out: *mut *mut BbsPool) -> c_int;
    fn bbs_pool_destroy(pool: *mut BbsPool);
    fn bbs_pool_device_count(pool: *const BbsPool) -> usize;
    fn bbs_pool_set_window_bits(pool: *mut BbsPool, curve: c_int, bits: c_int) -> c_int;
    fn bbs_pool_set_generators(pool: *mut BbsPool, curve: c_int, gens: *const u8, count: usize, api_id: *const u8, api_id_len: usize) -> c_int;
    fn bbs_pool_set_public_key(pool: *mut BbsPool, curve: c_int, pk: *const u8, is_identity: c_int) -> c_int;
    fn bbs_pool_set_inflight(pool: *mut BbsPool, jobs_per_member: c_int) -> c_int;
    fn bbs_pool_proof_verify_submit(pool: *mut BbsPool, lists: *const BbsPvList, n_lists: usize, max_batch: usize, job_out: *mut *mut BbsPoolJob) -> c_int;
    fn bbs_pool_job_wait(job: *mut BbsPoolJob) -> c_int;
    fn bbs_pool_job_free(job: *mut BbsPoolJob);
    fn bbs_pool_proof_verify(pool: *mut BbsPool, lists: *const BbsPvList, n_lists: usize, max_batch: usize) -> c_int;
    fn bbs_runtime_queue_budget(device_id: c_int, total: *mut c_int, pool: *mut c_int, dedicated_cap: *mut c_int, scratch_bytes_per_lane: *mut usize) -> c_int;
}
#[repr(C)] pub struct BbsIssuer { _p: [u8; 0] }
#[repr(C)] pub struct BbsPool { _p: [u8; 0] }
#[repr(C)] pub struct BbsPoolJob { _p: [u8; 0] }
/// `struct bbs_pv_list` of the header: the items of ONE curve of a list, in the layout of `bbs_core_proof_verify_submit`.
#[repr(C)]
pub struct BbsPvList {
    pub curve: c_int, pub n: usize,
    pub proofs_fixed: *const u8, pub commitments: *const u8, pub commit_off: *const u64,
    pub disclosed_msgs: *const u8, pub dmsg_off: *const u64, pub disclosed_idx: *const u64, pub didx_off: *const u64,
    pub headers: *const u8, pub hdr_off: *const u64, pub ph: *const u8, pub ph_off: *const u64,
    pub global_index: *const u64, pub status: *mut i8,
}

/// `core_proof_verify` (src/proof_verify.rs:64-116) over a LIST of proofs on SEVERAL GPUs from one process -- north_star's
/// "Rust host -> C ABI -> 8 GPUs".  The library owns one context per (curve, device) and one submitting thread per device; a
/// list is one `PackedProofs` per curve (what `GpuIssuer::proof_verify_submit` packs for one context); the statuses come back
/// in the caller's order.  Partition: by curve, then contiguous ceil(n / devices) items per device.
pub struct GpuPool { pool: *mut BbsPool }
/// one curve's items of a list, packed (owned buffers: they must outlive the call / the job)
pub struct PackedProofs {
    pub curve: c_int, pub n: usize, pub proofs_fixed: Vec<u8>, pub commitments: Vec<u8>, pub commit_off: Vec<u64>,
    pub disclosed_msgs: Vec<u8>, pub dmsg_off: Vec<u64>, pub disclosed_idx: Vec<u64>, pub didx_off: Vec<u64>,
    pub headers: Vec<u8>, pub hdr_off: Vec<u64>, pub ph: Vec<u8>, pub ph_off: Vec<u64>,
    /// position of item k of this section in the caller's whole list
    pub global_index: Vec<u64>,
}
impl GpuPool {
    pub fn new(devices: &[c_int]) -> Self {
        assert!(!devices.is_empty(), "GpuPool: at least one device");
        let mut pool = std::ptr::null_mut();
        let rc = unsafe { bbs_pool_create(devices.as_ptr(), devices.len(), &mut pool) };
        assert_eq!(rc, 0, "bbs_pool_create: {rc}");
        let me = GpuPool { pool };
        assert_eq!(unsafe { bbs_pool_device_count(me.pool) }, devices.len());
        me
    }
    /// the same generators and issuer key on every device (`generators_le` / `pk_affine_le`: the records of the header)
    pub fn configure<E: GpuCurve>(&self, generators_le: &[u8], count: usize, api_id: &[u8], pk_affine_le: &[u8], jobs_per_device: c_int) {
        let fpb = unsafe { bbs_fp_bytes(E::CURVE_ID) };
        assert_eq!(generators_le.len(), count * 2 * fpb, "generators: {count} affine points of {} bytes", 2 * fpb);
        assert_eq!(pk_affine_le.len(), 4 * fpb, "public key record: 4 x {fpb} bytes");
        unsafe {
            assert_eq!(bbs_pool_set_window_bits(self.pool, E::CURVE_ID, 0), 0);          // width by free device memory
            assert_eq!(bbs_pool_set_generators(self.pool, E::CURVE_ID, generators_le.as_ptr(), count, api_id.as_ptr(), api_id.len()), 0);
            assert_eq!(bbs_pool_set_public_key(self.pool, E::CURVE_ID, pk_affine_le.as_ptr(), 0), 0);
            assert_eq!(bbs_pool_set_inflight(self.pool, jobs_per_device), 0);
        }
    }
    /// The whole list, blocking; `total` = length of the caller's list (every section's `global_index` points into it).
    pub fn proof_verify(&self, sections: &[PackedProofs], total: usize) -> Vec<Result<bool, ProofGenError>> {
        let mut status = vec![-128i8; total.max(1)];
        let lists = Self::lists(sections, total, &mut status);
        let rc = unsafe { bbs_pool_proof_verify(self.pool, lists.as_ptr(), lists.len(), 0) };
        assert_eq!(rc, 0, "bbs_pool_proof_verify: {rc}");
        status[..total].iter().map(|&s| match s { 1 => Ok(true), 0 => Ok(false), e => Err(proof_error(e)) }).collect()
    }
    /// The same without waiting: several lists may be in flight (a device does not drain between them).
    pub fn proof_verify_submit<'a>(&'a self, sections: &'a [PackedProofs], total: usize) -> PendingList<'a> {
        let mut status = vec![-128i8; total.max(1)];
        let lists = Self::lists(sections, total, &mut status);
        let mut job = std::ptr::null_mut();
        let rc = unsafe { bbs_pool_proof_verify_submit(self.pool, lists.as_ptr(), lists.len(), 0, &mut job) };
        assert_eq!(rc, 0, "bbs_pool_proof_verify_submit: {rc}");
        PendingList { job, status, total, _lists: lists, _sections: sections }
    }
    fn lists(sections: &[PackedProofs], total: usize, status: &mut Vec<i8>) -> Vec<BbsPvList> {
        sections.iter().map(|s| {
            // every length the library will read, checked before the call
            assert!(s.commit_off.len() == s.n + 1 && s.dmsg_off.len() == s.n + 1 && s.didx_off.len() == s.n + 1
                    && s.hdr_off.len() == s.n + 1 && s.ph_off.len() == s.n + 1 && s.global_index.len() == s.n, "PackedProofs: offsets per item");
            assert!(s.global_index.iter().all(|&g| (g as usize) < total), "PackedProofs: global index out of the list");
            BbsPvList { curve: s.curve, n: s.n, proofs_fixed: s.proofs_fixed.as_ptr(), commitments: s.commitments.as_ptr(), commit_off: s.commit_off.as_ptr(),
                        disclosed_msgs: s.disclosed_msgs.as_ptr(), dmsg_off: s.dmsg_off.as_ptr(), disclosed_idx: s.disclosed_idx.as_ptr(), didx_off: s.didx_off.as_ptr(),
                        headers: s.headers.as_ptr(), hdr_off: s.hdr_off.as_ptr(), ph: s.ph.as_ptr(), ph_off: s.ph_off.as_ptr(),
                        global_index: s.global_index.as_ptr(), status: status.as_mut_ptr() }
        }).collect()
    }
    /// hardware queues the library will use on a device (pooled + dedicated) and the kernel frame that sets the bound
    pub fn queue_budget(device: c_int) -> (c_int, c_int, c_int, usize) {
        let (mut t, mut p, mut d, mut s) = (0, 0, 0, 0usize);
        let rc = unsafe { bbs_runtime_queue_budget(device, &mut t, &mut p, &mut d, &mut s) };
        assert_eq!(rc, 0, "bbs_runtime_queue_budget: {rc}");
        (t, p, d, s)
    }
}
impl Drop for GpuPool { fn drop(&mut self) { unsafe { bbs_pool_destroy(self.pool) } } }
/// a list in flight on the pool; borrows the pool and the packed sections, which therefore outlive it
pub struct PendingList<'a> { job: *mut BbsPoolJob, status: Vec<i8>, total: usize, _lists: Vec<BbsPvList>, _sections: &'a [PackedProofs] }
impl<'a> PendingList<'a> {
    pub fn wait(mut self) -> Vec<Result<bool, ProofGenError>> {
        let rc = unsafe { bbs_pool_job_wait(self.job) };
        unsafe { bbs_pool_job_free(self.job) };
        self.job = std::ptr::null_mut();
        assert_eq!(rc, 0, "bbs_pool_job_wait: {rc}");
        self.status[..self.total].iter().map(|&s| match s { 1 => Ok(true), 0 => Ok(false), e => Err(proof_error(e)) }).collect()
    }
}
impl<'a> Drop for PendingList<'a> { fn drop(&mut self) { if !self.job.is_null() { unsafe { bbs_pool_job_free(self.job) } } } }

/// `proof_verify` (src/proof_verify.rs:19-61) for a list of proofs whose numbers of messages differ: the reference derives the
/// generators from `proof.commitments.len() + disclosed_indexes.len() + 1` on every call (:40-43); here the library keeps
/// one table set per message count and routes the proofs.  Proofs travel as the octet strings a verifier receives
/// (`CanonicalSerialize` of the reference's `Proof`: 3 compressed G1 points, then the scalars), messages as raw bytes.
pub struct GpuIssuerAnyLength { issuer: *mut BbsIssuer }
impl GpuIssuerAnyLength {
    pub fn new<E: GpuCurve>(device: c_int, api_id: &[u8], pk_affine_le: &[u8], max_messages: usize) -> Self {
        // every length the FFI will read is checked BEFORE the first unsafe call: the library reads exactly 4 * fp_bytes of the key
        let fpb = unsafe { bbs_fp_bytes(E::CURVE_ID) };
        assert_eq!(pk_affine_le.len(), 4 * fpb, "public key record: x.c0 | x.c1 | y.c0 | y.c1, {} bytes each", fpb);
        let mut issuer = std::ptr::null_mut();
        let rc = unsafe { bbs_issuer_create(E::CURVE_ID, device, api_id.as_ptr(), api_id.len(), &mut issuer) };
        assert_eq!(rc, 0, "bbs_issuer_create: {rc}");
        // the owning value exists before the fallible calls: if one of them panics, Drop destroys the issuer
        let me = GpuIssuerAnyLength { issuer };
        unsafe {
            assert_eq!(bbs_issuer_set_limits(me.issuer, max_messages, 0), 0);       // table width by free device memory
            assert_eq!(bbs_issuer_set_public_key(me.issuer, pk_affine_le.as_ptr(), 0), 0);
        }
        me
    }
    /// One call for the whole list; `Err(InvalidMessageAndGeneratorsLength)` only for a proof above `max_messages`.
    pub fn proof_verify(&self, proof_octets: &[&[u8]], headers: &[&[u8]], phs: &[&[u8]], disclosed_messages: &[&[&[u8]]],
                        disclosed_indexes: &[&[usize]]) -> Vec<Result<bool, ProofGenError>> {
        let n = proof_octets.len();
        assert!(headers.len() == n && phs.len() == n && disclosed_messages.len() == n && disclosed_indexes.len() == n,
                "proof_verify: every input slice holds one entry per proof ({n})");
        let (ob, oo) = ragged(proof_octets);
        let (mut mb, mut mbo, mut mio, mut di, mut dio) = (Vec::new(), vec![0u64], vec![0u64], Vec::new(), vec![0u64]);
        for i in 0..n {
            for m in disclosed_messages[i] { mb.extend_from_slice(m); mbo.push(mb.len() as u64); }
            mio.push((mbo.len() - 1) as u64);
            di.extend(disclosed_indexes[i].iter().map(|&x| x as u64));
            dio.push(di.len() as u64);
        }
        let (hb, ho) = ragged(headers);
        let (pb, po) = ragged(phs);
        let mut status = vec![-128i8; n];
        let rc = unsafe { bbs_issuer_proof_verify(self.issuer, n, ob.as_ptr(), oo.as_ptr(), mb.as_ptr(), mbo.as_ptr(), mio.as_ptr(),
                                                  di.as_ptr(), dio.as_ptr(), hb.as_ptr(), ho.as_ptr(), pb.as_ptr(), po.as_ptr(), status.as_mut_ptr()) };
        assert_eq!(rc, 0, "bbs_issuer_proof_verify: {rc}");
        status.iter().map(|&s| match s { 1 => Ok(true), 0 => Ok(false), e => Err(proof_error(e)) }).collect()
    }
}
impl Drop for GpuIssuerAnyLength { fn drop(&mut self) { unsafe { bbs_issuer_destroy(self.issuer) } } }

/// Curves the engine knows (BBS_CURVE_* of the header).
pub trait GpuCurve: Pairing {
    const CURVE_ID: c_int;
}
impl GpuCurve for ark_bls12_381::Bls12_381 { const CURVE_ID: c_int = 0; }
impl GpuCurve for ark_bn254::Bn254 { const CURVE_ID: c_int = 1; }

// ---- canonical little-endian encodings of include/bbs_sign_amd.h ("Data formats") ------------------------------------
fn put_fr<F: PrimeField>(x: &F, out: &mut Vec<u8>) { out.extend(x.into_bigint().to_bytes_le()); }          // 32 B
fn get_fr<F: PrimeField>(b: &[u8]) -> F { F::from_le_bytes_mod_order(b) }                                   // canonical in, exact

fn put_g1<E: Pairing>(p: &E::G1, fpb: usize, out: &mut Vec<u8>)
where <E::G1Affine as AffineRepr>::BaseField: PrimeField {
    match p.into_affine().xy() {
        None => out.extend(std::iter::repeat(0u8).take(2 * fpb)),                                          // identity = all zero
        Some((x, y)) => { out.extend(x.into_bigint().to_bytes_le()); out.extend(y.into_bigint().to_bytes_le()); }
    }
}
fn get_g1<E: Pairing>(b: &[u8], fpb: usize) -> E::G1
where <E::G1Affine as AffineRepr>::BaseField: PrimeField {
    if b.iter().all(|&v| v == 0) { return E::G1::default(); }
    let x = <<E::G1Affine as AffineRepr>::BaseField>::from_le_bytes_mod_order(&b[..fpb]);
    let y = <<E::G1Affine as AffineRepr>::BaseField>::from_le_bytes_mod_order(&b[fpb..2 * fpb]);
    E::G1Affine::new_unchecked(x, y).into_group()           // the engine only returns points it computed on the curve
}

/// Ragged byte strings -> flat buffer + n + 1 offsets.
fn ragged(items: &[&[u8]]) -> (Vec<u8>, Vec<u64>) {
    let (mut flat, mut off) = (Vec::new(), vec![0u64]);
    for it in items { flat.extend_from_slice(it); off.push(flat.len() as u64); }
    (flat, off)
}

/// Per-item status of the header -> the reference's errors.  BBS_ST_* values: -1 InvalidMessageAndGeneratorsLength,
/// -2 InvalidDisclosedIndicesLength, -3 InvalidDisclosedIndex, -4 InvalidRandomScalarsAndUndisclosedIndicesLength,
/// -5 InvalidUndisclosedIndicesLength, -6 InvalidIndicesAndMessagesLength; -20 .. -23 are the reference's panics
/// (sign.rs:129, proof_gen.rs:346, proof_verify.rs:177-179, utilities_helper.rs:46-52) and are re-raised as panics.
fn proof_error(st: i8) -> ProofGenError {
    match st {
        -1 => ProofGenError::InvalidMessageAndGeneratorsLength,
        -2 => ProofGenError::InvalidDisclosedIndicesLength,
        -3 => ProofGenError::InvalidDisclosedIndex,
        -4 => ProofGenError::InvalidRandomScalarsAndUndisclosedIndicesLength,
        -5 => ProofGenError::InvalidUndisclosedIndicesLength,
        -6 => ProofGenError::InvalidIndicesAndMessagesLength,
        other => panic!("bbs_sign_amd status {other}"),
    }
}
fn signature_error(st: i8) -> SignatureError {
    match st {
        -1 => SignatureError::InvalidMessageAndGeneratorsLength,
        other => panic!("bbs_sign_amd status {other}"),
    }
}

pub struct GpuIssuer<E: GpuCurve, F: PrimeField, C> {
    ctx: *mut BbsCtx,
    api_id: Vec<u8>,
    fpb: usize,
    l: usize,
    _m: PhantomData<(E, F, C)>,
}

impl<E, F, C> GpuIssuer<E, F, C>
where
    E: GpuCurve,
    F: PrimeField + FromOkm<48, F>,
    C: for<'a> Constants<'a, E>,
    <E::G1Affine as AffineRepr>::BaseField: PrimeField,
    <E::G2Affine as AffineRepr>::BaseField: ark_ff::Field,
{
    /// Context for items with `l` messages.  Call `set_public_key` (verify / proof_gen / proof_verify) or
    /// `set_secret_key` (sign; also sets the public key) next.
    pub fn new(device: i32, l: usize, window_bits: i32) -> Result<Self, c_int> {
        unsafe {
            let fpb = bbs_fp_bytes(E::CURVE_ID);                                                   // step 1
            let api_id = [C::CIPHERSUITE_ID, b"H2G_HM2S_"].concat();                              // src/sign.rs:44
            let mut gens = vec![0u8; (l + 1) * 2 * fpb];
            let rc = bbs_create_generators(E::CURVE_ID, l + 1, api_id.as_ptr(), api_id.len(), gens.as_mut_ptr());   // step 2
            if rc != 0 { return Err(rc); }
            let mut ctx = std::ptr::null_mut();
            let rc = bbs_ctx_create(E::CURVE_ID, device, &mut ctx);                               // step 3
            if rc != 0 { return Err(rc); }
            bbs_ctx_set_window_bits(ctx, window_bits);                                             // step 4 (20: 52 GB of tables at l = 32)
            let rc = bbs_ctx_set_generators(ctx, gens.as_ptr(), l + 1, api_id.as_ptr(), api_id.len());   // step 5
            if rc != 0 { bbs_ctx_destroy(ctx); return Err(rc); }
            // every E::G1 inside a Proof<E, F> / Signature<E, F> is a checked subgroup member (ark-ec), so the shim can vouch
            bbs_ctx_set_points_in_subgroup(ctx, 1);                                                // step 6
            Ok(Self { ctx, api_id, fpb, l, _m: PhantomData })
        }
    }

    /// `sk` of `SecretKey::sign` (sign.rs:32); the library derives pk = sk * BP2 (key_gen.rs:83-90, sign.rs:81).
    pub fn set_secret_key(&mut self, sk: &SecretKey<F>) -> Result<(), c_int> {
        let mut b = Vec::new();
        put_fr(&sk.sk, &mut b);
        let rc = unsafe { bbs_ctx_set_secret_key(self.ctx, b.as_ptr()) };                          // step 8
        if rc != 0 { Err(rc) } else { Ok(()) }
    }

    /// `pk` of verify / proof_gen / proof_verify (key_gen.rs:12-15): x.c0 | x.c1 | y.c0 | y.c1, canonical LE.
    pub fn set_public_key(&mut self, pk: &PublicKey<E>) -> Result<(), c_int>
    where <E::G2Affine as AffineRepr>::BaseField: ark_ff::Field<BasePrimeField = <E::G1Affine as AffineRepr>::BaseField> {
        let a = pk.pk.into_affine();
        let rc = match a.xy() {
            None => unsafe { bbs_ctx_set_public_key(self.ctx, std::ptr::null(), 1) },
            Some((x, y)) => {
                let mut b = Vec::new();
                for c in x.to_base_prime_field_elements().chain(y.to_base_prime_field_elements()) {
                    b.extend(c.into_bigint().to_bytes_le());
                }
                unsafe { bbs_ctx_set_public_key(self.ctx, b.as_ptr(), 0) }                         // step 9 (get) / alt
            }
        };
        if rc != 0 { Err(rc) } else { Ok(()) }
    }

    /// msg_to_scalars (interface_utilities.rs:76-88) of every message of every item in one device call:
    /// flat scalars (32 B LE each) + per-item offsets in scalars.
    fn msg_to_scalars(&self, items: &[&[&[u8]]]) -> (Vec<u8>, Vec<u64>) {
        let (mut flat, mut off, mut item_off) = (Vec::new(), vec![0u64], vec![0u64]);
        for item in items {
            for m in *item { flat.extend_from_slice(m); off.push(flat.len() as u64); }
            item_off.push((off.len() - 1) as u64);
        }
        let dst = [self.api_id.as_slice(), b"MAP_MSG_TO_SCALAR_AS_HASH_"].concat();
        let mut sc = vec![0u8; 32 * (off.len() - 1)];
        let rc = unsafe { bbs_hash_to_scalar_batch(self.ctx, off.len() - 1, flat.as_ptr(), off.as_ptr(), dst.as_ptr(), dst.len(), sc.as_mut_ptr()) };   // step 10
        assert_eq!(rc, 0, "bbs_hash_to_scalar_batch: {rc}");
        (sc, item_off)
    }

    fn put_signatures(&self, sigs: &[Signature<E, F>]) -> Vec<u8> {
        let mut out = Vec::new();
        for s in sigs { put_g1::<E>(&s.a, self.fpb, &mut out); put_fr(&s.e, &mut out); }
        out
    }

    /// `SecretKey::sign` (src/sign.rs:32-60) for n items.
    pub fn sign(&self, messages: &[&[&[u8]]], headers: &[&[u8]]) -> Vec<Result<Signature<E, F>, SignatureError>> {
        let n = messages.len();
        let (ms, mo) = self.msg_to_scalars(messages);
        let (hb, ho) = ragged(headers);
        let rec = 2 * self.fpb + 32;
        let (mut out, mut st) = (vec![0u8; n * rec], vec![0i8; n]);
        let rc = unsafe { bbs_core_sign_batch(self.ctx, n, ms.as_ptr(), mo.as_ptr(), hb.as_ptr(), ho.as_ptr(), out.as_mut_ptr(), st.as_mut_ptr()) };   // step 11
        assert_eq!(rc, 0, "bbs_core_sign_batch: {rc}");
        (0..n).map(|i| if st[i] == 1 {
            let r = &out[i * rec..(i + 1) * rec];
            Ok(Signature { a: get_g1::<E>(r, self.fpb), e: get_fr::<F>(&r[2 * self.fpb..]) })
        } else { Err(signature_error(st[i])) }).collect()
    }

    /// `PublicKey::verify` (src/verify.rs:18-50) for n items.
    pub fn verify(&self, signatures: &[Signature<E, F>], headers: &[&[u8]], messages: &[&[&[u8]]]) -> Vec<Result<bool, SignatureError>> {
        let n = signatures.len();
        let sg = self.put_signatures(signatures);
        let (ms, mo) = self.msg_to_scalars(messages);
        let (hb, ho) = ragged(headers);
        let mut st = vec![0i8; n];
        let rc = unsafe { bbs_core_verify_batch(self.ctx, n, sg.as_ptr(), ms.as_ptr(), mo.as_ptr(), hb.as_ptr(), ho.as_ptr(), st.as_mut_ptr()) };   // step 12
        assert_eq!(rc, 0, "bbs_core_verify_batch: {rc}");
        st.into_iter().map(|s| match s { 1 => Ok(true), 0 => Ok(false), e => Err(signature_error(e)) }).collect()
    }

    /// `proof_gen` (src/proof_gen.rs:78-113) for n items.  The random scalars are drawn here exactly as the reference does
    /// (`calculate_random_scalars(5 + l - r)`, proof_gen.rs:145-149, with r the UN-deduplicated number of indexes) and
    /// passed in, so the engine stays deterministic.
    pub fn proof_gen(&self, signatures: &[Signature<E, F>], headers: &[&[u8]], phs: &[&[u8]], messages: &[&[&[u8]]],
                     disclosed_indexes: &[&[usize]]) -> Vec<Result<Proof<E, F>, ProofGenError>> {
        let n = signatures.len();
        let sg = self.put_signatures(signatures);
        let (ms, mo) = self.msg_to_scalars(messages);
        let (hb, ho) = ragged(headers);
        let (pb, po) = ragged(phs);
        let (mut di, mut dio, mut rs, mut ro) = (Vec::new(), vec![0u64], Vec::new(), vec![0u64]);
        for i in 0..n {
            di.extend(disclosed_indexes[i].iter().map(|&x| x as u64));
            dio.push(di.len() as u64);
            let (l, r) = (messages[i].len(), disclosed_indexes[i].len());
            // r > l is the reference's first error (proof_gen.rs:135-137): the engine reports it; draw nothing
            let count = if r <= l { 5 + l - r } else { 0 };
            for s in calculate_random_scalars::<48, F>(count) { put_fr(&s, &mut rs); }
            ro.push((rs.len() / 32) as u64);
        }
        let rec = 6 * self.fpb + 128;
        let total: usize = messages.iter().map(|m| m.len()).sum();
        let (mut pf, mut cm, mut cmo, mut st) = (vec![0u8; n * rec], vec![0u8; 32 * total.max(1)], vec![0u64; n + 1], vec![0i8; n]);
        let rc = unsafe {
            bbs_core_proof_gen_batch(self.ctx, n, sg.as_ptr(), ms.as_ptr(), mo.as_ptr(), di.as_ptr(), dio.as_ptr(), rs.as_ptr(), ro.as_ptr(),
                                     hb.as_ptr(), ho.as_ptr(), pb.as_ptr(), po.as_ptr(), pf.as_mut_ptr(), cm.as_mut_ptr(), cmo.as_mut_ptr(),
                                     st.as_mut_ptr())                                               // step 14
        };
        assert_eq!(rc, 0, "bbs_core_proof_gen_batch: {rc}");
        (0..n).map(|i| if st[i] == 1 {
            let r = &pf[i * rec..(i + 1) * rec];
            let f = self.fpb;
            let sc = |k: usize| get_fr::<F>(&r[6 * f + 32 * k..6 * f + 32 * (k + 1)]);
            Ok(Proof {
                a_bar: get_g1::<E>(&r[..2 * f], f), b_bar: get_g1::<E>(&r[2 * f..4 * f], f), d: get_g1::<E>(&r[4 * f..6 * f], f),
                e_cap: sc(0), r1_cap: sc(1), r3_cap: sc(2),
                commitments: (cmo[i] as usize..cmo[i + 1] as usize).map(|k| get_fr::<F>(&cm[32 * k..32 * (k + 1)])).collect(),
                challenge: Challenge { scalar: sc(3) },
            })
        } else { Err(proof_error(st[i])) }).collect()
    }

    /// `proof_verify` (src/proof_verify.rs:19-61) for n proofs: the asynchronous form, so that a serving loop keeps
    /// several batches in flight -- `submit` returns at once, `PendingVerify::wait` yields the n results.
    pub fn proof_verify_submit<'a>(&'a self, proofs: &[Proof<E, F>], headers: &[&[u8]], phs: &[&[u8]], disclosed_messages: &[&[&[u8]]],
                               disclosed_indexes: &[&[usize]]) -> PendingVerify<'a> {
        let n = proofs.len();
        let (dm, dmo) = self.msg_to_scalars(disclosed_messages);
        let (mut fixed, mut cm, mut cmo, mut di, mut dio) = (Vec::new(), Vec::new(), vec![0u64], Vec::new(), vec![0u64]);
        for (i, p) in proofs.iter().enumerate() {
            put_g1::<E>(&p.a_bar, self.fpb, &mut fixed); put_g1::<E>(&p.b_bar, self.fpb, &mut fixed); put_g1::<E>(&p.d, self.fpb, &mut fixed);
            put_fr(&p.e_cap, &mut fixed); put_fr(&p.r1_cap, &mut fixed); put_fr(&p.r3_cap, &mut fixed); put_fr(&p.challenge.scalar, &mut fixed);
            for c in &p.commitments { put_fr(c, &mut cm); }
            cmo.push((cm.len() / 32) as u64);
            di.extend(disclosed_indexes[i].iter().map(|&x| x as u64));
            dio.push(di.len() as u64);
        }
        let (hb, ho) = ragged(headers);
        let (pb, po) = ragged(phs);
        let mut status = vec![-128i8; n].into_boxed_slice();      // stays valid (heap) until wait()
        let mut job = std::ptr::null_mut();
        let rc = unsafe {
            bbs_core_proof_verify_submit(self.ctx, n, fixed.as_ptr(), cm.as_ptr(), cmo.as_ptr(), dm.as_ptr(), dmo.as_ptr(), di.as_ptr(),
                                         dio.as_ptr(), hb.as_ptr(), ho.as_ptr(), pb.as_ptr(), po.as_ptr(), status.as_mut_ptr(), &mut job)   // step 15
        };
        assert_eq!(rc, 0, "bbs_core_proof_verify_submit: {rc}");
        PendingVerify { job, status, _issuer: std::marker::PhantomData }   // the input buffers may be dropped: the library staged them
    }

    /// Synchronous convenience: one batch, wait.
    pub fn proof_verify(&self, proofs: &[Proof<E, F>], headers: &[&[u8]], phs: &[&[u8]], disclosed_messages: &[&[&[u8]]],
                        disclosed_indexes: &[&[usize]]) -> Vec<Result<bool, ProofGenError>> {
        self.proof_verify_submit(proofs, headers, phs, disclosed_messages, disclosed_indexes).wait()
    }

    pub fn messages_per_item(&self) -> usize { self.l }
}

/// A submitted proof_verify batch.  It borrows the issuer it was submitted to: the job's streams and buffers belong to that
/// issuer's context, so the borrow checker refuses to drop the issuer (bbs_ctx_destroy) while a batch is outstanding.
pub struct PendingVerify<'a> { job: *mut BbsJob, status: Box<[i8]>, _issuer: std::marker::PhantomData<&'a ()> }
impl<'a> PendingVerify<'a> {
    pub fn wait(mut self) -> Vec<Result<bool, ProofGenError>> {
        let rc = unsafe { bbs_job_wait(self.job) };                                                // step 16
        unsafe { bbs_job_free(self.job) };                                                         // step 17
        self.job = std::ptr::null_mut();
        assert_eq!(rc, 0, "bbs_job_wait: {rc} (-102: an item was left undecided; the library fails closed)");
        self.status.iter().map(|&s| match s { 1 => Ok(true), 0 => Ok(false), e => Err(proof_error(e)) }).collect()
    }
}
/// Completion-order retire for a serving loop with several batches in flight (bbs_jobs_wait_any): sleeps until ONE of the
/// pending batches has finished, removes it from `pending` and returns its position there together with its results.
/// Batches submitted together share the chip and do not finish in submission order; waiting for the oldest one first makes
/// the loop run in convoys.
pub fn wait_any<'a>(pending: &mut Vec<PendingVerify<'a>>) -> Option<(usize, Vec<Result<bool, ProofGenError>>)> {
    if pending.is_empty() { return None; }
    let jobs: Vec<*mut BbsJob> = pending.iter().map(|p| p.job).collect();
    let mut k = 0usize;
    let rc = unsafe { bbs_jobs_wait_any(jobs.as_ptr(), jobs.len(), &mut k) };
    assert_eq!(rc, 0, "bbs_jobs_wait_any: {rc} (-102: an item was left undecided; the library fails closed)");
    assert!(k < pending.len());
    let done = pending.remove(k);
    Some((k, done.wait()))                // the job has been delivered: wait() returns at once and frees it (steps 16, 17)
}
impl<'a> Drop for PendingVerify<'a> {
    fn drop(&mut self) { if !self.job.is_null() { unsafe { bbs_job_wait(self.job); bbs_job_free(self.job); } } }
}

impl<E: GpuCurve, F: PrimeField, C> Drop for GpuIssuer<E, F, C> {
    fn drop(&mut self) { unsafe { bbs_ctx_destroy(self.ctx) } }                                     // step 18
}
