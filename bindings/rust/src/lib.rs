//! Host shim: the reference's `proof_verify` / `PublicKey::verify` / `SecretKey::sign` / `proof_gen` call sites
//! (src/proof_verify.rs:19-61, src/verify.rs:18-50, src/sign.rs:32-60, src/proof_gen.rs:78-113) re-pointed at the
//! MI355X engine through the C ABI of include/bbs_sign_amd.h.
//!
//! SOURCE ONLY: the build image has no Rust toolchain; this file has never been compiled.  It documents the
//! binding a maintainer would add; the identical call sequence is what bbs_sign_amd/api.py does (tested).
//!
//! Shape: a `GpuIssuer<E>` owns one engine context = (curve, generators for L messages, issuer public key).
//! The reference recomputes `create_generators` on every call (sign.rs:49, verify.rs:35, proof_gen.rs:98,
//! proof_verify.rs:40-43); here it is computed once by the library (`bbs_create_generators`) when the context is
//! built.  Proofs are verified in batches: `verify_proofs` takes n proofs and returns n `Result<bool, _>`.
use ark_ec::{pairing::Pairing, AffineRepr, CurveGroup};
use ark_ff::{BigInteger, PrimeField};
use std::os::raw::c_int;

use bbs_plus::proof_gen::{Proof, ProofGenError};

#[repr(C)]
pub struct BbsCtx { _p: [u8; 0] }

extern "C" {
    fn bbs_fp_bytes(curve: c_int) -> usize;
    fn bbs_ctx_create(curve: c_int, device_id: c_int, out: *mut *mut BbsCtx) -> c_int;
    fn bbs_ctx_destroy(ctx: *mut BbsCtx);
    fn bbs_ctx_set_window_bits(ctx: *mut BbsCtx, bits: c_int) -> c_int;
    fn bbs_ctx_set_generators(ctx: *mut BbsCtx, gens: *const u8, count: usize, api_id: *const u8, api_id_len: usize) -> c_int;
    fn bbs_ctx_set_public_key(ctx: *mut BbsCtx, pk: *const u8, is_identity: c_int) -> c_int;
    fn bbs_ctx_set_batch_verification(ctx: *mut BbsCtx, enabled: c_int, seed32: *const u8) -> c_int;
    fn bbs_ctx_set_points_in_subgroup(ctx: *mut BbsCtx, vouched: c_int) -> c_int;
    fn bbs_create_generators(curve: c_int, count: usize, api_id: *const u8, api_id_len: usize, out_affine: *mut u8) -> c_int;
    fn bbs_hash_to_scalar_batch(ctx: *mut BbsCtx, n: usize, msgs: *const u8, msg_off: *const u64,
                                dst: *const u8, dst_len: usize, scalars_out: *mut u8) -> c_int;
    fn bbs_core_proof_verify_batch(ctx: *mut BbsCtx, n: usize, proofs_fixed: *const u8,
        commitments: *const u8, commit_off: *const u64, disclosed_msgs: *const u8, dmsg_off: *const u64,
        disclosed_idx: *const u64, didx_off: *const u64, headers: *const u8, hdr_off: *const u64,
        ph: *const u8, ph_off: *const u64, status: *mut i8) -> c_int;
    // bbs_core_sign_batch / bbs_core_verify_batch / bbs_core_proof_gen_batch, the job API (upload / run / wait /
    // fetch) and the octet codec follow the same pattern: see include/bbs_sign_amd.h
}

/// Curves the engine knows (BBS_CURVE_* of the header).
pub trait GpuCurve: Pairing {
    const CURVE_ID: c_int;
    const CIPHERSUITE_ID: &'static [u8];
}
impl GpuCurve for ark_bls12_381::Bls12_381 {
    const CURVE_ID: c_int = 0;
    const CIPHERSUITE_ID: &'static [u8] = b"BBS_BLS12381G1_XMD:SHA-256_SSWU_RO_";
}
impl GpuCurve for ark_bn254::Bn254 {
    const CURVE_ID: c_int = 1;
    const CIPHERSUITE_ID: &'static [u8] = b"BBS_QUUX-V01-CS02-with-BN254G1_XMD:SHA-256_SVDW_RO_";
}

fn put_fr<F: PrimeField>(x: &F, out: &mut Vec<u8>) { out.extend(x.into_bigint().to_bytes_le()); }      // 32 B canonical LE

fn put_g1<E: Pairing>(p: &E::G1, fpb: usize, out: &mut Vec<u8>)
where <E::G1Affine as AffineRepr>::BaseField: PrimeField {
    match p.into_affine().xy() {
        None => out.extend(std::iter::repeat(0u8).take(2 * fpb)),                                      // identity = all zero
        Some((x, y)) => { out.extend(x.into_bigint().to_bytes_le()); out.extend(y.into_bigint().to_bytes_le()); }
    }
}

/// Per-item status of the header -> the reference's Result (src/proof_gen.rs ProofGenError, proof_verify.rs:139-150).
fn status_to_result(st: i8) -> Result<bool, ProofGenError> {
    match st {
        1 => Ok(true),
        0 => Ok(false),
        -1 => Err(ProofGenError::InvalidDisclosedIndex),
        -2 => Err(ProofGenError::InvalidIndicesAndMessagesLength),
        -3 => Err(ProofGenError::InvalidMessageAndGeneratorsLength),
        // -20.. are the reference's panics (dst too long, index out of bounds): re-raise them as panics
        other => panic!("bbs_sign_amd status {other}"),
    }
}

pub struct GpuIssuer<E: GpuCurve> {
    ctx: *mut BbsCtx,
    api_id: Vec<u8>,
    fpb: usize,
    _e: std::marker::PhantomData<E>,
}

impl<E: GpuCurve> GpuIssuer<E>
where <E::G1Affine as AffineRepr>::BaseField: PrimeField, E::ScalarField: PrimeField {
    /// Context for proofs over `l` messages of issuer `pk_affine` (x.c0 | x.c1 | y.c0 | y.c1, canonical LE).
    pub fn new(device: i32, l: usize, pk_affine: &[u8], batch_verification: bool) -> Result<Self, c_int> {
        unsafe {
            let fpb = bbs_fp_bytes(E::CURVE_ID);
            let api_id = [E::CIPHERSUITE_ID, b"H2G_HM2S_"].concat();            // src/proof_verify.rs:35
            let mut gens = vec![0u8; (l + 1) * 2 * fpb];
            let rc = bbs_create_generators(E::CURVE_ID, l + 1, api_id.as_ptr(), api_id.len(), gens.as_mut_ptr());
            if rc != 0 { return Err(rc); }
            let mut ctx = std::ptr::null_mut();
            let rc = bbs_ctx_create(E::CURVE_ID, device, &mut ctx);
            if rc != 0 { return Err(rc); }
            bbs_ctx_set_window_bits(ctx, 20);   // 52 GB of tables for L = 32 on a 288 GB device; 16 -> 4 GB, a few % slower
            let rc = bbs_ctx_set_generators(ctx, gens.as_ptr(), l + 1, api_id.as_ptr(), api_id.len());
            if rc != 0 { bbs_ctx_destroy(ctx); return Err(rc); }
            let rc = bbs_ctx_set_public_key(ctx, pk_affine.as_ptr(), 0);
            if rc != 0 { bbs_ctx_destroy(ctx); return Err(rc); }
            // every E::G1Affine inside a Proof<E> / Signature<E> is a checked subgroup member (ark-ec), so the shim can vouch
            bbs_ctx_set_points_in_subgroup(ctx, 1);
            if batch_verification { bbs_ctx_set_batch_verification(ctx, 1, std::ptr::null()); }
            Ok(Self { ctx, api_id, fpb, _e: std::marker::PhantomData })
        }
    }

    /// `proof_verify` (src/proof_verify.rs:19-61) for n proofs at once: byte messages in, Result<bool> per item out.
    pub fn verify_proofs(&self, proofs: &[Proof<E>], headers: &[&[u8]], phs: &[&[u8]],
                         disclosed_msgs: &[&[&[u8]]], disclosed_idx: &[&[usize]]) -> Vec<Result<bool, ProofGenError>> {
        let n = proofs.len();
        // msg_to_scalars (interface_utilities.rs:76-88) for all disclosed messages of the batch in one device call
        let (mut flat, mut off) = (Vec::new(), vec![0u64]);
        for item in disclosed_msgs { for m in *item { flat.extend_from_slice(m); off.push(flat.len() as u64); } }
        let dst = [self.api_id.as_slice(), b"MAP_MSG_TO_SCALAR_AS_HASH_"].concat();
        let mut dm = vec![0u8; 32 * (off.len() - 1)];
        unsafe { bbs_hash_to_scalar_batch(self.ctx, off.len() - 1, flat.as_ptr(), off.as_ptr(), dst.as_ptr(), dst.len(), dm.as_mut_ptr()); }
        // records: a_bar | b_bar | d | e_cap | r1_cap | r3_cap | challenge, then ragged commitments / indexes / bytes
        let (mut fixed, mut cm, mut cm_off) = (Vec::new(), Vec::new(), vec![0u64]);
        let (mut dm_off, mut di, mut di_off) = (vec![0u64], Vec::new(), vec![0u64]);
        let (mut hb, mut h_off, mut pb, mut p_off) = (Vec::new(), vec![0u64], Vec::new(), vec![0u64]);
        for (i, p) in proofs.iter().enumerate() {
            put_g1::<E>(&p.a_bar, self.fpb, &mut fixed); put_g1::<E>(&p.b_bar, self.fpb, &mut fixed); put_g1::<E>(&p.d, self.fpb, &mut fixed);
            put_fr(&p.e_cap, &mut fixed); put_fr(&p.r1_cap, &mut fixed); put_fr(&p.r3_cap, &mut fixed); put_fr(&p.challenge, &mut fixed);
            for c in &p.commitments { put_fr(c, &mut cm); }
            cm_off.push((cm.len() / 32) as u64);
            dm_off.push(dm_off[i] + disclosed_msgs[i].len() as u64);
            di.extend(disclosed_idx[i].iter().map(|&x| x as u64)); di_off.push(di.len() as u64);
            hb.extend_from_slice(headers[i]); h_off.push(hb.len() as u64);
            pb.extend_from_slice(phs[i]); p_off.push(pb.len() as u64);
        }
        let mut status = vec![0i8; n];
        let rc = unsafe {
            bbs_core_proof_verify_batch(self.ctx, n, fixed.as_ptr(), cm.as_ptr(), cm_off.as_ptr(), dm.as_ptr(), dm_off.as_ptr(),
                                        di.as_ptr(), di_off.as_ptr(), hb.as_ptr(), h_off.as_ptr(), pb.as_ptr(), p_off.as_ptr(),
                                        status.as_mut_ptr())
        };
        assert_eq!(rc, 0, "bbs_core_proof_verify_batch: {rc}");
        status.into_iter().map(status_to_result).collect()
    }
}

impl<E: GpuCurve> Drop for GpuIssuer<E> {
    fn drop(&mut self) { unsafe { bbs_ctx_destroy(self.ctx) } }
}
