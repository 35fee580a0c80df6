/*
 * bbs_sign_amd -- C ABI of the MI355X-native batched BBS+ engine.
 *
 * This is the drop-in boundary for the hot path of hashcloak/bbs_sign: the `core_*` level of the
 * reference (the narrowest seam that isolates all heavy arithmetic and that the reference's own
 * tests call directly, src/tests/core_sign_tests.rs:53-64).  Each entry point names the reference
 * function it replaces.  Plain pointers and sizes only; no exceptions cross this boundary.
 *
 * Data formats (both curves):
 *   Fp / Fr element : canonical value < modulus, LITTLE-endian bytes (48 B for BLS12-381 Fp,
 *                     32 B otherwise) == ark-ff `into_bigint().to_bytes_le()`.
 *   G1 affine       : x || y (2 * fp_bytes); the identity is encoded as all-zero bytes.
 *   G2 affine       : x.c0 || x.c1 || y.c0 || y.c1 (4 * fp_bytes) + a separate identity flag.
 *   scalar          : 32 B LE, must be < r (else per-item status BBS_ST_NONCANONICAL).
 *   ragged arrays   : a flat buffer + an offsets array of n+1 uint64 (item i owns
 *                     [off[i], off[i+1]) in units of elements: scalars, indexes or bytes).
 *
 * Per-item status (int8): mirrors the reference's Result<bool|T, Error>:
 *     1  Ok(true) / Ok(value)        0  Ok(false)
 *   < 0  the Err variant, a reference panic, or an input the reference's types cannot hold.
 * -128 and -127 are internal "not decided yet" states: the library never returns them -- a fetch that finds one
 * fails with BBS_E_STATE instead (an item no kernel decided must not read as Ok(true)).
 * Function return: 0 on success, BBS_E_* on a batch-level failure (nothing was computed).
 */
#ifndef BBS_SIGN_AMD_H
#define BBS_SIGN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBS_CURVE_BLS12_381 0
#define BBS_CURVE_BN254 1

#define BBS_FR_BYTES 32

/* per-item status codes */
#define BBS_ST_TRUE 1
#define BBS_ST_FALSE 0
/* SignatureError (src/sign.rs:24-28) / ProofGenError (src/proof_gen.rs:59-75) */
#define BBS_ST_INVALID_MESSAGE_AND_GENERATORS_LENGTH (-1)
#define BBS_ST_INVALID_DISCLOSED_INDICES_LENGTH (-2)
#define BBS_ST_INVALID_DISCLOSED_INDEX (-3)
#define BBS_ST_INVALID_RANDOM_SCALARS_AND_UNDISCLOSED_INDICES_LENGTH (-4)
#define BBS_ST_INVALID_UNDISCLOSED_INDICES_LENGTH (-5)
#define BBS_ST_INVALID_INDICES_AND_MESSAGES_LENGTH (-6)
/* KeyGenError (src/key_gen.rs:34-42) */
#define BBS_ST_INVALID_KEY_MATERIAL_LENGTH (-7)
#define BBS_ST_INVALID_KEY_INFO_LENGTH (-8)
#define BBS_ST_INVALID_SECRET_KEY (-9)
/* reference panics */
#define BBS_ST_PANIC_SK_PLUS_E_ZERO (-20)     /* src/sign.rs:129 unwrap on inverse of 0 */
#define BBS_ST_PANIC_R2_ZERO (-21)            /* src/proof_gen.rs:346 unwrap on inverse of 0 */
#define BBS_ST_PANIC_INDEX_OUT_OF_BOUNDS (-22)/* src/proof_verify.rs:177-179 commitments[i] */
#define BBS_ST_PANIC_DST_TOO_LONG (-23)       /* src/utils/utilities_helper.rs:46-52 */
/* inputs arkworks' types cannot represent */
#define BBS_ST_NONCANONICAL (-40)             /* scalar >= r or coordinate >= p */
#define BBS_ST_NOT_ON_CURVE (-41)            /* also: not in the prime-order subgroup (codec) */
#define BBS_ST_INVALID_ENCODING (-42)        /* octet string of the wrong shape / forbidden identity or zero */
#define BBS_ST_NO_RESOURCES (-43)            /* bbs_issuer_* only: the context of this item's message count could not be set
                                              * up (device memory, or every resident context busy at the context limit); the
                                              * item was NOT computed -- the other items of the call were; retry later */

/* batch-level errors */
#define BBS_OK 0
#define BBS_E_ARG (-100)
#define BBS_E_HIP (-101)
#define BBS_E_STATE (-102)       /* generators / public key / secret key not set; or an item left undecided */
#define BBS_E_PUBLIC_KEY (-103)  /* public key not on the twist or not of order r */
#define BBS_E_NO_DEVICE (-104)
#define BBS_E_NOMEM (-105)
#define BBS_E_UNSUPPORTED (-106) /* reserved: operation not available for this curve */
#define BBS_E_NO_RESOURCES (-107)/* the hardware-queue budget of the device is spent (bbs_runtime_queue_budget): nothing was created */

typedef struct bbs_ctx bbs_ctx;
typedef struct bbs_job bbs_job;

/* ------------------------------------------------------------------------------------------
 * Context: one per (curve, GPU, issuer key, generator set).  Holds the device-resident
 * fixed-base window tables of {P1, Q1, H_1..H_L}, the domain-hash midstate and the Miller-loop
 * line tables of the issuer key.  Thread-compatible (one call at a time per context).
 * ------------------------------------------------------------------------------------------ */
size_t bbs_fp_bytes(int curve);
const char* bbs_version(void);
/* Hash of the sources (every file under bbs_sign_amd/csrc and this header) the library was built from; bbs_sign_amd/build.py rebuilds
 * when it differs from the tree's (a prebuilt library travels with the tree: staleness is judged by content). */
const char* bbs_source_hash(void);
/* GPU_MAX_HW_QUEUES as the process environment has it (0 = unset).  The library sets 20 when it is loaded unless the
 * variable is already set; the HIP runtime reads it at its own initialisation, so the setting only takes effect if
 * this library was loaded before the process's first HIP call ("Build / deployment"). */
int bbs_runtime_hw_queues(void);
/* For a process that cannot arrange that (its first HIP call -- any torch import that touches the GPU -- comes before this
 * library is loaded: the runtime's pool is then 4 hardware queues, several jobs share one, 1.30 M proof_verify/s instead
 * of 1.50 M): up to k job streams per device get a hardware queue OF THEIR OWN (streams created with an all-ones
 * compute-unit mask, which the runtime does not draw from the pool) -- 1.50 M/s whatever GPU_MAX_HW_QUEUES says
 * (profiles/r04_f_dedicated_queues.log).  k = 0: off (the default); 12 is a good value, 16 the most accepted.  k is a WISH:
 * every hardware queue reserves scratch memory for the largest kernel frame it has run, pool + dedicated queues x that frame
 * is a budget, and past it the runtime first collapses and then ABORTS the process -- so the library grants
 * min(k, bbs_runtime_queue_budget's dedicated_cap) and serves further streams from the pool (round 5; before that, 16 on top of
 * a pool of 14 could abort).
 * Call before the first context is created (streams are recycled); BBS_DEDICATED_QUEUES=k in the environment does the same
 * (1 means 12).  Caveat: such streams synchronise with the legacy default stream (the runtime offers no non-blocking flag
 * for them): a process that also runs its own kernels on stream 0 -- torch and RCCL work on the default stream -- serialises
 * them with the jobs. */
int bbs_runtime_set_dedicated_queues(int k);
/* The hardware-queue budget of a device, as the library enforces it: *scratch_bytes_per_lane = the largest kernel frame of
 * this library (every kernel is asked at start-up), *total = hardware queues (pooled + dedicated) that frame allows within
 * 8.5 % of the device's memory (queues x bytes per lane x 64 lanes x wave slots; 25 on MI355X at 1.8 KB), *pool =
 * GPU_MAX_HW_QUEUES as the environment has it (4 when unset), *dedicated_cap = max(0, total - pool).  If the POOL alone exceeds
 * the budget (an explicit GPU_MAX_HW_QUEUES=32) the library creates at most `total` streams: a job that gets none of its own
 * shares its context's stream and runs its side-stream stages in order on it, and bbs_ctx_create fails with
 * BBS_E_NO_RESOURCES once even the context's stream cannot be had -- slower or refused, never the runtime's abort.  Any
 * pointer may be NULL.  BBS_E_NO_DEVICE for a device that does not exist. */
int bbs_runtime_queue_budget(int device_id, int* total, int* pool, int* dedicated_cap, size_t* scratch_bytes_per_lane);
int bbs_device_count(void);
size_t bbs_device_free_bytes(int device_id);          /* device memory free right now (0: no such device) */

int bbs_ctx_create(int curve, int device_id, bbs_ctx** out);
void bbs_ctx_destroy(bbs_ctx* ctx);
/* device memory of the context's tables (fixed-base window tables, line tables, constants), in bytes */
size_t bbs_ctx_table_bytes(const bbs_ctx* ctx);

/* window width (bits) of the fixed-base tables, 4..22, or 0 (THE DEFAULT) = chosen at bbs_ctx_set_generators from the
 * memory free on the device: the widest of 20 / 16 / 12 / 8 whose tables fit 1/32 of it and 8 GiB (32 messages on an empty
 * MI355X: 16 bits, 2 GB; ask for 20 explicitly where one issuer may have 26 GB for +3.5 %) -- and, if that
 * allocation fails all the same (the free figure is a snapshot: other ranks or processes on the device, fragmentation),
 * the next narrower width, down to 8, before BBS_E_NOMEM; takes effect at the next bbs_ctx_set_generators, which changes
 * nothing of the context unless it succeeds.  Digits are SIGNED: table bytes = (count+1) * ceil(256/w) * 2^(w-1) * 2 * limb bytes of
 * an Fp element (56 on BLS12-381, 40 on BN254) -- for 32 messages 26 GB at w = 20 (1.45 M proof_verify/s, 8.0 M sign/s),
 * 2 GB at 16 (1.30 M, 7.3 M), 172 MB at 12 (1.30 M, 6.1 M), 16 MB at 8 (1.18 M, 4.4 M). */
int bbs_ctx_set_window_bits(bbs_ctx* ctx, int bits);

/* Subgroup vouching (off by default).  The reference's point types (ark-ec `Affine`, built by `deserialize_compressed`
 * or `Affine::new`, src/proof_gen.rs:29, src/sign.rs:18) can only hold members of the prime-order subgroup; this ABI
 * takes raw coordinates and by default computes exactly what the reference's double-and-add would for ANY on-curve
 * point -- with ONE exception, core_proof_gen: Abar e and Abar e~ (src/proof_gen.rs:255-258) are computed as the multiples
 * (r1 r2 e mod r) A and (r1 r2 e~ mod r) A of the signature's own point, which is the reference's value for every A in the
 * prime-order subgroup (all that the reference's types can hold) and for the identity; for an on-curve A OUTSIDE the
 * subgroup -- a signature that cannot verify -- the proof's Bbar, T1 and what follows from them differ from the
 * reference's (tests: check_proof_gen_unusual_points pins what is computed instead).  vouched = 1 promises that every G1 point later handed to core_verify / core_proof_verify / core_proof_gen (and the generators given to
 * bbs_ctx_set_generators) through this context is in the subgroup (true for everything that came out of bbs_*_from_octets*, which check it): on BLS12-381
 * the variable-base multiplications of those paths then use the GLV endomorphism split (half the doublings).  Results
 * are identical for such inputs; for on-curve points outside the subgroup they are unspecified.  BN254 has cofactor 1
 * (every on-curve point is in the subgroup), so there the split is always on and this setting changes nothing.
 * Takes effect for jobs uploaded afterwards. */
int bbs_ctx_set_points_in_subgroup(bbs_ctx* ctx, int vouched);

/* Latency form of a job.  A batch can be laid out for the least WORK (the throughput form: T1 = Bbar*c + Abar*e^ + D*r1^,
 * src/proof_verify.rs:163-164, as ONE joint windowed chain on one lane per item; both Miller loops of an item's pairing
 * product on one six-lane group with shared squarings) or for the shortest CRITICAL PATH (the latency form: the three
 * multiplications on three lanes, summed afterwards; the two Miller loops on separate wavefronts, multiplied before the
 * final exponentiation).  Same group elements and booleans; the latency form is ~25 % shorter for a batch that has the
 * chip to itself and costs ~15 % more instructions, so it loses once several batches are in flight.
 *   enabled = 0: never; 1: always; 2: AUTO (the default) -- a job gets the latency form iff at most one other job is alive
 *   ON THE DEVICE when it is created (upload / submit; jobs of every context of the process count: the contexts of one
 *   bbs_issuer, or a BLS12-381 and a BN254 context side by side, share the chip), i.e. a serving loop that keeps many
 *   batches in flight runs in the throughput form, a caller that verifies one batch at a time gets the short path
 *   without asking.
 * Takes effect for jobs created afterwards. */
int bbs_ctx_set_latency_mode(bbs_ctx* ctx, int enabled);
/* Fixed-base sums (the generators' multiples: B of sign / verify / proof_gen, the fixed part of T2 of proof_verify) as
 * ONE tree of affine additions per item with one shared inversion per level, instead of eight chains of mixed Jacobian
 * additions: fewer field multiplications, the same group element, bit-identical results.  Needs work memory of about
 * 90 bytes x table points (messages + 2) x windows per item of a job.  EXPERIMENTAL and off by default: in its present
 * form it is not faster (measured); today it covers proof_verify and bbs_g1_msm_batch.  Applies to jobs
 * created afterwards. */
int bbs_ctx_set_fixed_base_tree(bbs_ctx* ctx, int enabled);

/* Batch verification for core_proof_verify and core_verify (off by default).  When enabled, the n two-pairing
 * products of a batch (src/proof_verify.rs:112-115, src/verify.rs:88-92) are replaced by 16 products over random
 * linear combinations of the items' G1 arguments (sum rho_i * Abar_i, sum rho_i * Bbar_i; for verify
 * sum rho_i * A_i, sum rho_i * (e_i A_i - B_i)) with 16 independent 8-bit coefficients rho_i per item derived
 * from a secret seed (bucket-method multi-scalar multiplication on the device), taken over the items that passed
 * every earlier check (a proof_verify job in its latency form combines beside the challenge check instead of behind it:
 * over the structurally valid items); only if one of those combined checks fails are the items still pending checked
 * one by one.  The booleans
 * equal the reference's except with probability 2^-128 per batch.  seed32 = NULL draws the seed from the operating
 * system; a caller-supplied seed must be secret and fresh.  Takes effect for jobs uploaded afterwards. */
int bbs_ctx_set_batch_verification(bbs_ctx* ctx, int enabled, const uint8_t* seed32);

/* generators = [Q1, H_1 .. H_L] (count = L+1 affine G1 points) and the api_id they belong to:
 * the `generators: &[E::G1]` and `api_id: &[u8]` arguments of every reference core_* function
 * (src/sign.rs:63-69, src/verify.rs:53-60, src/proof_gen.rs:116-125, src/proof_verify.rs:64-73). */
int bbs_ctx_set_generators(bbs_ctx* ctx, const uint8_t* generators_affine, size_t count,
                           const uint8_t* api_id, size_t api_id_len);

/* issuer public key (`pk: PublicKey<E>`, src/key_gen.rs:12-15). */
int bbs_ctx_set_public_key(bbs_ctx* ctx, const uint8_t* pk_affine, int is_identity);
/* issuer secret key (`&self` of core_sign, src/sign.rs:63); also sets pk = sk * BP2
 * (src/key_gen.rs:83-90, src/sign.rs:81). */
int bbs_ctx_set_secret_key(bbs_ctx* ctx, const uint8_t* sk32);
int bbs_ctx_get_public_key(bbs_ctx* ctx, uint8_t* pk_affine_out, int* is_identity_out);
/* ark-serialize compressed public key as hashed into the domain (96 B BLS / 64 B BN254). */
int bbs_ctx_get_public_key_compressed(bbs_ctx* ctx, uint8_t* out, size_t cap, size_t* len_out);

/* ------------------------------------------------------------------------------------------
 * Batched core operations.  `*_upload` stages the batch in page-locked memory, copies it to HBM with one
 * asynchronous copy and enqueues the ingest kernel -- the reference's checks in the reference's order, range
 * checks and unpacking run on the device -- and returns a device-resident job; `bbs_job_run` enqueues the
 * kernels on the job's streams (asynchronous); `bbs_job_wait` blocks; `bbs_job_fetch_*` copies results back.
 * The one-shot `*_batch` functions do all of it; `*_submit` is the asynchronous one-shot form.
 * ------------------------------------------------------------------------------------------ */

/* core_proof_verify (src/proof_verify.rs:64-116 with proof_verify_init :119-188).
 * proofs_fixed: n records of  a_bar || b_bar || d (3 G1 affine) || e_cap || r1_cap || r3_cap ||
 * challenge (4 scalars)  == 6*fp_bytes + 128 bytes each (src/proof_gen.rs:29-39).
 * commitments / disclosed_msgs: scalars, ragged; disclosed_idx: uint64 indexes, ragged, caller
 * order is significant exactly as in the reference (src/proof_verify.rs:18). */
int bbs_core_proof_verify_upload(bbs_ctx* ctx, size_t n, const uint8_t* proofs_fixed,
                                 const uint8_t* commitments, const uint64_t* commit_off,
                                 const uint8_t* disclosed_msgs, const uint64_t* dmsg_off,
                                 const uint64_t* disclosed_idx, const uint64_t* didx_off,
                                 const uint8_t* headers, const uint64_t* hdr_off,
                                 const uint8_t* ph, const uint64_t* ph_off, bbs_job** job_out);
int bbs_core_proof_verify_batch(bbs_ctx* ctx, size_t n, const uint8_t* proofs_fixed,
                                const uint8_t* commitments, const uint64_t* commit_off,
                                const uint8_t* disclosed_msgs, const uint64_t* dmsg_off,
                                const uint64_t* disclosed_idx, const uint64_t* didx_off,
                                const uint8_t* headers, const uint64_t* hdr_off,
                                const uint8_t* ph, const uint64_t* ph_off, int8_t* status);

/* Asynchronous form of bbs_core_proof_verify_batch, for a stream of batches from ONE submitting thread per GPU: the
 * inputs are staged in page-locked memory (the caller's input buffers may be reused as soon as the call returns), then
 * ONE host-to-device copy, the validation / unpacking kernel (the reference's checks of src/proof_verify.rs:139-150 and
 * the range checks run on the device), the verification kernels and the copy of the statuses back are enqueued; the
 * call returns without waiting for the device, so several submitted batches are in flight together.
 * bbs_job_wait(job) blocks until `status` (n entries; must stay valid until then) has been written -- it returns
 * BBS_E_STATE, and writes nothing, if any item was left undecided (fail closed); then bbs_job_free(job). */
int bbs_core_proof_verify_submit(bbs_ctx* ctx, size_t n, const uint8_t* proofs_fixed,
                                 const uint8_t* commitments, const uint64_t* commit_off,
                                 const uint8_t* disclosed_msgs, const uint64_t* dmsg_off,
                                 const uint64_t* disclosed_idx, const uint64_t* didx_off,
                                 const uint8_t* headers, const uint64_t* hdr_off,
                                 const uint8_t* ph, const uint64_t* ph_off, int8_t* status, bbs_job** job_out);

/* core_proof_verify from the WIRE: the n proofs as octet strings (compress(Abar) || compress(Bbar) || compress(D) || e^ ||
 * r1^ || r3^ || m^_1 .. m^_U || c, scalars 32 bytes big-endian: the form of src/tests/test_vector.rs:199-260; ragged,
 * oct_off: n + 1 byte offsets) instead of decoded records.  Decompression (square roots), on-curve and prime-order-subgroup
 * checks and the scalar range checks run on the device in front of the same pipeline; the per-item status is what
 * bbs_proof_from_octets followed by bbs_core_proof_verify_batch would give: BBS_ST_INVALID_ENCODING (shape, identity
 * point), BBS_ST_NONCANONICAL, BBS_ST_NOT_ON_CURVE (also: outside the subgroup) before the reference's own checks.  The
 * disclosed messages are scalars, as for bbs_core_proof_verify_batch.  Because every point has been subgroup-checked, the
 * variable-base terms use the GLV split (BLS12-381) without bbs_ctx_set_points_in_subgroup. */
int bbs_proof_verify_octets_submit(bbs_ctx* ctx, size_t n, const uint8_t* proof_octets, const uint64_t* oct_off,
                                   const uint8_t* disclosed_msgs, const uint64_t* dmsg_off,
                                   const uint64_t* disclosed_idx, const uint64_t* didx_off,
                                   const uint8_t* headers, const uint64_t* hdr_off,
                                   const uint8_t* ph, const uint64_t* ph_off, int8_t* status, bbs_job** job_out);
int bbs_proof_verify_octets_batch(bbs_ctx* ctx, size_t n, const uint8_t* proof_octets, const uint64_t* oct_off,
                                  const uint8_t* disclosed_msgs, const uint64_t* dmsg_off,
                                  const uint64_t* disclosed_idx, const uint64_t* didx_off,
                                  const uint8_t* headers, const uint64_t* hdr_off,
                                  const uint8_t* ph, const uint64_t* ph_off, int8_t* status);
/* The reference's PUBLIC proof_verify (src/proof_verify.rs:19-61) for contexts of a fixed number of messages, in one
 * call: proof octet strings and the disclosed messages as RAW BYTES.  Message t of the batch is
 * msg_bytes[msg_byte_off[t] .. msg_byte_off[t + 1]) (t counts the disclosed messages of all items in order);
 * msg_item_off[i] .. msg_item_off[i + 1] are the messages of item i (n + 1 entries, in messages; as with every offset
 * array of this header they need not start at zero: a window into larger arrays is passed by pointing into them).  msg_to_scalars
 * (interface_utilities.rs:76-88, dst = api_id || "MAP_MSG_TO_SCALAR_AS_HASH_") runs on the device in front of the checks;
 * statuses as bbs_proof_verify_octets_* on the hashed messages (BBS_ST_PANIC_DST_TOO_LONG for an item with disclosed
 * messages when that dst exceeds 255 bytes, as the reference's expand_message panics). */
int bbs_proof_verify_wire_submit(bbs_ctx* ctx, size_t n, const uint8_t* proof_octets, const uint64_t* oct_off,
                                 const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                                 const uint64_t* disclosed_idx, const uint64_t* didx_off,
                                 const uint8_t* headers, const uint64_t* hdr_off,
                                 const uint8_t* ph, const uint64_t* ph_off, int8_t* status, bbs_job** job_out);
int bbs_proof_verify_wire_batch(bbs_ctx* ctx, size_t n, const uint8_t* proof_octets, const uint64_t* oct_off,
                                const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                                const uint64_t* disclosed_idx, const uint64_t* didx_off,
                                const uint8_t* headers, const uint64_t* hdr_off,
                                const uint8_t* ph, const uint64_t* ph_off, int8_t* status);

/* core_verify (src/verify.rs:53-93).  signatures: n records  A (G1 affine) || e (scalar). */
int bbs_core_verify_upload(bbs_ctx* ctx, size_t n, const uint8_t* signatures,
                           const uint8_t* messages, const uint64_t* msg_off,
                           const uint8_t* headers, const uint64_t* hdr_off, bbs_job** job_out);
int bbs_core_verify_batch(bbs_ctx* ctx, size_t n, const uint8_t* signatures,
                          const uint8_t* messages, const uint64_t* msg_off,
                          const uint8_t* headers, const uint64_t* hdr_off, int8_t* status);
/* asynchronous form, as bbs_core_proof_verify_submit: bbs_job_wait(job) delivers `status`, then bbs_job_free(job) */
int bbs_core_verify_submit(bbs_ctx* ctx, size_t n, const uint8_t* signatures,
                           const uint8_t* messages, const uint64_t* msg_off,
                           const uint8_t* headers, const uint64_t* hdr_off, int8_t* status, bbs_job** job_out);

/* verify from the wire: n signature octet strings of fp_bytes + 32 octets each, compress(A) || I2OSP(e, 32) (the byte
 * string of src/tests/test_vector.rs:188-191), decompressed, subgroup- and range-checked on the device in front of
 * core_verify.  status[i] = what bbs_signature_from_octets gives when it fails (malformed / not on the curve or in the
 * subgroup / identity / e = 0 or >= r), else what core_verify gives.  _submit as bbs_core_verify_submit. */
int bbs_verify_octets_submit(bbs_ctx* ctx, size_t n, const uint8_t* signature_octets,
                             const uint8_t* messages, const uint64_t* msg_off,
                             const uint8_t* headers, const uint64_t* hdr_off, int8_t* status, bbs_job** job_out);
int bbs_verify_octets_batch(bbs_ctx* ctx, size_t n, const uint8_t* signature_octets,
                            const uint8_t* messages, const uint64_t* msg_off,
                            const uint8_t* headers, const uint64_t* hdr_off, int8_t* status);

/* The reference's PUBLIC verify (src/verify.rs:18-50) for a context's number of messages in one call: signature octet
 * strings and the messages as RAW BYTES (layout of the message arrays as bbs_proof_verify_wire_submit); msg_to_scalars on
 * the device.  Statuses as bbs_verify_octets_* on the hashed messages. */
int bbs_verify_wire_submit(bbs_ctx* ctx, size_t n, const uint8_t* signature_octets,
                           const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                           const uint8_t* headers, const uint64_t* hdr_off, int8_t* status, bbs_job** job_out);
int bbs_verify_wire_batch(bbs_ctx* ctx, size_t n, const uint8_t* signature_octets,
                          const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                          const uint8_t* headers, const uint64_t* hdr_off, int8_t* status);
/* The reference's PUBLIC sign (src/sign.rs:32-60): raw messages in, signature octet strings out (as bbs_sign_octets_*). */
int bbs_sign_wire_submit(bbs_ctx* ctx, size_t n, const uint8_t* msg_bytes, const uint64_t* msg_byte_off,
                         const uint64_t* msg_item_off, const uint8_t* headers, const uint64_t* hdr_off,
                         uint8_t* signature_octets_out, int8_t* status, bbs_job** job_out);
int bbs_sign_wire_batch(bbs_ctx* ctx, size_t n, const uint8_t* msg_bytes, const uint64_t* msg_byte_off,
                        const uint64_t* msg_item_off, const uint8_t* headers, const uint64_t* hdr_off,
                        uint8_t* signature_octets_out, int8_t* status);

/* core_sign (src/sign.rs:63-133); needs bbs_ctx_set_secret_key.
 * signatures_out: n records A || e (status 1 where written). */
int bbs_core_sign_upload(bbs_ctx* ctx, size_t n, const uint8_t* messages, const uint64_t* msg_off,
                         const uint8_t* headers, const uint64_t* hdr_off, bbs_job** job_out);
int bbs_core_sign_batch(bbs_ctx* ctx, size_t n, const uint8_t* messages, const uint64_t* msg_off,
                        const uint8_t* headers, const uint64_t* hdr_off, uint8_t* signatures_out,
                        int8_t* status);
/* asynchronous form: returns with everything enqueued; bbs_job_wait(job) delivers `status` and writes the records into
 * signatures_out (may be NULL); both buffers must stay valid until then.  The inputs may be released on return. */
int bbs_core_sign_submit(bbs_ctx* ctx, size_t n, const uint8_t* messages, const uint64_t* msg_off,
                         const uint8_t* headers, const uint64_t* hdr_off, uint8_t* signatures_out,
                         int8_t* status, bbs_job** job_out);

/* sign to the wire: signature_octets_out receives n strings of fp_bytes + 32 octets, compress(A) || I2OSP(e, 32) (the
 * byte string of src/tests/test_vector.rs:188-191), all zero for an item whose status is not 1; compression on the device. */
int bbs_sign_octets_submit(bbs_ctx* ctx, size_t n, const uint8_t* messages, const uint64_t* msg_off,
                           const uint8_t* headers, const uint64_t* hdr_off, uint8_t* signature_octets_out,
                           int8_t* status, bbs_job** job_out);
int bbs_sign_octets_batch(bbs_ctx* ctx, size_t n, const uint8_t* messages, const uint64_t* msg_off,
                          const uint8_t* headers, const uint64_t* hdr_off, uint8_t* signature_octets_out,
                          int8_t* status);

/* core_proof_gen (src/proof_gen.rs:116-208: proof_init :211-269, proof_challenge_calculate
 * :272-328, proof_finalize :331-365).  random_scalars replaces the draw at :145-149 and must hold
 * 5 + L - R scalars per item (R = number of disclosed indexes BEFORE dedup, as the reference
 * sizes it), else BBS_E_ARG.  Outputs: proofs_fixed_out (same record as above), commitments_out
 * packed with commit_off_out (n+1 entries; capacity sum_i L_i scalars is always enough). */
int bbs_core_proof_gen_upload(bbs_ctx* ctx, size_t n, const uint8_t* signatures,
                              const uint8_t* messages, const uint64_t* msg_off,
                              const uint64_t* disclosed_idx, const uint64_t* didx_off,
                              const uint8_t* random_scalars, const uint64_t* rnd_off,
                              const uint8_t* headers, const uint64_t* hdr_off,
                              const uint8_t* ph, const uint64_t* ph_off, bbs_job** job_out);
int bbs_core_proof_gen_batch(bbs_ctx* ctx, size_t n, const uint8_t* signatures,
                             const uint8_t* messages, const uint64_t* msg_off,
                             const uint64_t* disclosed_idx, const uint64_t* didx_off,
                             const uint8_t* random_scalars, const uint64_t* rnd_off,
                             const uint8_t* headers, const uint64_t* hdr_off,
                             const uint8_t* ph, const uint64_t* ph_off,
                             uint8_t* proofs_fixed_out, uint8_t* commitments_out,
                             uint64_t* commit_off_out, int8_t* status);
/* proof_gen to the wire: the proofs as octet strings (src/tests/test_vector.rs:199-260: three compressed points, e^,
 * r1^, r3^, the m^ of the undisclosed messages, c; scalars 32 bytes big-endian) packed into octets_out, item i at
 * oct_off_out[i] .. oct_off_out[i + 1] (n + 1 byte offsets; an item whose status is not 1 has an empty string).
 * Capacity n x (3 fp_bytes + 32 (4 + L)) is always enough. */
int bbs_proof_gen_octets_submit(bbs_ctx* ctx, size_t n, const uint8_t* signatures,
                                const uint8_t* messages, const uint64_t* msg_off,
                                const uint64_t* disclosed_idx, const uint64_t* didx_off,
                                const uint8_t* random_scalars, const uint64_t* rnd_off,
                                const uint8_t* headers, const uint64_t* hdr_off,
                                const uint8_t* ph, const uint64_t* ph_off,
                                uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status, bbs_job** job_out);
int bbs_proof_gen_octets_batch(bbs_ctx* ctx, size_t n, const uint8_t* signatures,
                               const uint8_t* messages, const uint64_t* msg_off,
                               const uint64_t* disclosed_idx, const uint64_t* didx_off,
                               const uint8_t* random_scalars, const uint64_t* rnd_off,
                               const uint8_t* headers, const uint64_t* hdr_off,
                               const uint8_t* ph, const uint64_t* ph_off,
                               uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status);
/* The reference's PUBLIC proof_gen (src/proof_gen.rs:78-113) for a context's number of messages in one call: signature
 * octet strings and the messages as RAW BYTES in (layout as bbs_proof_verify_wire_submit), proof octet strings out (as
 * bbs_proof_gen_octets_*).  Statuses: what bbs_signature_from_octets gives when it fails, else msg_to_scalars' panic
 * (BBS_ST_PANIC_DST_TOO_LONG), else core_proof_gen's. */
int bbs_proof_gen_wire_submit(bbs_ctx* ctx, size_t n, const uint8_t* signature_octets,
                              const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                              const uint64_t* disclosed_idx, const uint64_t* didx_off,
                              const uint8_t* random_scalars, const uint64_t* rnd_off,
                              const uint8_t* headers, const uint64_t* hdr_off,
                              const uint8_t* ph, const uint64_t* ph_off,
                              uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status, bbs_job** job_out);
int bbs_proof_gen_wire_batch(bbs_ctx* ctx, size_t n, const uint8_t* signature_octets,
                             const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                             const uint64_t* disclosed_idx, const uint64_t* didx_off,
                             const uint8_t* random_scalars, const uint64_t* rnd_off,
                             const uint8_t* headers, const uint64_t* hdr_off,
                             const uint8_t* ph, const uint64_t* ph_off,
                             uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status);
/* asynchronous form, as bbs_core_sign_submit: bbs_job_wait(job) delivers `status` and the three outputs */
int bbs_core_proof_gen_submit(bbs_ctx* ctx, size_t n, const uint8_t* signatures,
                              const uint8_t* messages, const uint64_t* msg_off,
                              const uint64_t* disclosed_idx, const uint64_t* didx_off,
                              const uint8_t* random_scalars, const uint64_t* rnd_off,
                              const uint8_t* headers, const uint64_t* hdr_off,
                              const uint8_t* ph, const uint64_t* ph_off,
                              uint8_t* proofs_fixed_out, uint8_t* commitments_out,
                              uint64_t* commit_off_out, int8_t* status, bbs_job** job_out);

int bbs_job_run(bbs_job* job);                       /* asynchronous */
int bbs_job_wait(bbs_job* job);
/* Completion-order retire for a serving loop that keeps several batches in flight (the reference has no counterpart: its
 * calls are synchronous, src/proof_verify.rs:19-61).  Jobs submitted together do not finish in submission order -- they
 * share the chip.  What completion order buys is measured: it matters where jobs differ in length or belong to different
 * lists (pipelined lists of BASELINE configs[4]: 7.3 -> 4.9 ms per list, an issuer's groups); on a loop of UNIFORM jobs, the
 * headline loop, it changes nothing (1.42 - 1.44 M/s either way, profiles/r04_b_bench_{any,fifo}_*.json: the convoys are
 * made on the GPU).
 * bbs_jobs_wait_any sleeps until ONE of jobs[0 .. n) -- entries may be NULL; jobs that were never run are ignored -- has
 * finished everything enqueued for it, then does what bbs_job_wait does for that job (delivers statuses / records of the
 * submit forms, BBS_E_STATE if an item was left undecided) and stores its position in *index_out; if several have
 * finished, the one that finished first.  The caller then frees or re-runs that job and calls again with the rest.
 * Event-driven: a host function placed on the job's stream behind its last operation wakes the waiter; nothing polls the
 * device.  BBS_E_STATE if no job of the set has been run.  Jobs of different contexts, curves and devices may be mixed.
 * A job whose last bbs_job_run could not be enqueued completely counts as finished WITH AN ERROR: it is handed out at once
 * (*index_out set) and the call -- like bbs_job_wait on it -- returns BBS_E_HIP and delivers nothing (it never blocks a set).
 * bbs_job_poll: 1 if the job has finished everything enqueued for it (bbs_job_wait will not block), 0 if not. */
int bbs_jobs_wait_any(bbs_job* const* jobs, size_t n, size_t* index_out);
int bbs_job_poll(const bbs_job* job);
size_t bbs_job_size(const bbs_job* job);
/* device memory the job holds until bbs_job_free, in bytes (sizing: INTEGRATION.md); batch verification adds its own */
size_t bbs_job_device_bytes(const bbs_job* job);
int bbs_job_fetch_status(bbs_job* job, int8_t* status);
int bbs_job_fetch_signatures(bbs_job* job, uint8_t* signatures_out);
int bbs_job_fetch_proofs(bbs_job* job, uint8_t* proofs_fixed_out, uint8_t* commitments_out,
                         uint64_t* commit_off_out);
void bbs_job_free(bbs_job* job);

/* Run the job `reps` times back to back and time it with HIP events recorded on the context's
 * own stream: total_ms = whole pipeline; kernel_ms[k] = accumulated time of stage k
 * (n_stages_out stages, names via bbs_job_stage_name).  Used by bench.py for the roofline line. */
int bbs_job_run_timed(bbs_job* job, int reps, float* total_ms, float* kernel_ms, int kernel_cap,
                      int* n_stages_out);
const char* bbs_job_stage_name(const bbs_job* job, int stage);
/* Throughput form: `njobs` device-resident batches in flight, step k runs on jobs[k % njobs] (every
 * job owns a HIP stream pair, so independent batches overlap on the GPU).  total_ms = first event to
 * the latest last-stage event; kernel_ms[s] = sum over all steps of stage s's own duration. */
int bbs_jobs_run_timed(bbs_job** jobs, int njobs, int steps, float* total_ms, float* kernel_ms,
                       int kernel_cap, int* n_stages_out);

/* Stage timers for the submit / run forms: jobs created after bbs_ctx_set_stage_timing(ctx, 1) record a HIP event pair
 * around every stage of every run, each on the stream the stage is launched on.  bbs_job_stage_times (after
 * bbs_job_wait) reads the LAST run: total_ms = first stage start .. last stage stop, kernel_ms[k] = duration of stage
 * k (names: bbs_job_stage_name).  BBS_E_STATE for a job created without timing or not yet run. */
int bbs_ctx_set_stage_timing(bbs_ctx* ctx, int enabled);
int bbs_job_stage_times(bbs_job* job, float* total_ms, float* kernel_ms, int kernel_cap, int* n_stages_out);

/* ------------------------------------------------------------------------------------------
 * Unit-parity primitives (hash_to_scalar, G1 multi-scalar multiplication, pairing product).
 * ------------------------------------------------------------------------------------------ */
/* hash_to_scalar (src/utils/core_utilities.rs:11-21) of n ragged messages under one dst. */
int bbs_hash_to_scalar_batch(bbs_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off,
                             const uint8_t* dst, size_t dst_len, uint8_t* scalars_out);
/* out[i] = sum_k fixed_scalars[i][k] * G_k  +  sum_m var_scalars[i][m] * var_points[i][m]
 * with G = [P1, Q1, H_1..H_L] of the context (n_fixed <= L+2), affine output. */
int bbs_g1_msm_batch(bbs_ctx* ctx, size_t n, const uint8_t* fixed_scalars, size_t n_fixed,
                     const uint8_t* var_points, const uint8_t* var_scalars, size_t n_var,
                     uint8_t* out_affine, int8_t* status);
/* out = sum_i scalars[i] * points[i]: one large variable-base multi-scalar multiplication over n per-item
 * affine points by the bucket (Pippenger) method (the arkworks `VariableBaseMSM` shape; the reference itself
 * only ever sums <= 38 terms per item, src/proof_verify.rs:163-182).  status[i] = 1, or -40 / -41 for an item
 * that is not canonical / not on the curve (it then contributes nothing). */
int bbs_g1_msm_pippenger(bbs_ctx* ctx, size_t n, const uint8_t* points_affine, const uint8_t* scalars,
                         uint8_t* out_affine, int* out_is_identity, int8_t* status);
/* status[i] = ( e(Pa[i], pk) * e(Pb[i], BP2) == 1 ) */
int bbs_pairing_product2_is_one_batch(bbs_ctx* ctx, size_t n, const uint8_t* pa_affine,
                                      const uint8_t* pb_affine, int8_t* status);

/* ------------------------------------------------------------------------------------------
 * bbs_issuer: the reference's four PUBLIC functions over batches whose items differ in their number of messages.
 *
 * The reference chooses the generators by the item's own length on every call: create_generators(messages.len() + 1) in
 * sign / verify / proof_gen (src/sign.rs:44-49, src/verify.rs:30-35, src/proof_gen.rs:91-96) and
 * create_generators(proof.commitments.len() + disclosed_indexes.len() + 1) in proof_verify (src/proof_verify.rs:40-43).
 * A bbs_ctx holds the tables of one generator set; a bbs_issuer holds one context per message count it has seen (created
 * on first use: generators by hash-to-curve on the host, window and line tables on the device, window width by
 * bbs_ctx_set_window_bits(ctx, 0) unless bbs_issuer_set_limits says otherwise) and routes the items of a call: grouped by
 * message count, every group through the context's one-call wire form (bbs_*_wire_submit), all groups in flight together,
 * results scattered back into the caller's order.  Arguments as the bbs_*_wire_* functions of the same name.
 *
 * Statuses: exactly those of the wire form under the item's own context -- so BBS_ST_INVALID_MESSAGE_AND_GENERATORS_LENGTH
 * cannot occur (as in the reference's public functions), EXCEPT for an item with more than max_messages messages (default
 * 1024; bbs_issuer_set_limits), which gets that code and is not computed: a table set per length is device memory an
 * untrusted caller must not be able to allocate without bound.  A proof octet string of the wrong shape is
 * BBS_ST_INVALID_ENCODING (its length does not define a message count).
 *
 * Resident contexts are BOUNDED: at most max_contexts (default 64) and max_table_bytes of window tables (default: half of
 * the device memory free when the issuer builds its first context) stay on the device; when a new message count arrives
 * and a bound is reached, idle contexts leave least-recently-used first (a context is idle when no routed list is in
 * flight on it) and are rebuilt if their length comes back.  The message count of a proof is read from the length of an
 * untrusted octet string: the bounds are what keeps a caller who sends one proof of every length from filling the device.
 * If the context of a group cannot be set up -- the device is out of memory even for the narrowest tables, or every
 * resident context is busy at the limit -- the items of THAT group get BBS_ST_NO_RESOURCES and the rest of the call is
 * served.  A new context is built (hash-to-curve on the host, table build on the device) under that context's own lock:
 * calls for other message counts proceed meanwhile.
 *
 * Threads: bbs_issuer_* may be called on one issuer from several threads.  Submissions to the same context are
 * serialised inside (the bbs_ctx contract: one call at a time per context), every routed list owns its jobs.  The
 * configuration calls (set_public_key, set_secret_key, set_modes) are BBS_E_STATE while a routed list is in flight --
 * its jobs read the contexts' keys -- and otherwise take effect for every context, resident or future, at its next use.
 * ------------------------------------------------------------------------------------------ */
typedef struct bbs_issuer bbs_issuer;
int bbs_issuer_create(int curve, int device_id, const uint8_t* api_id, size_t api_id_len, bbs_issuer** out);
void bbs_issuer_destroy(bbs_issuer* issuer);
int bbs_issuer_set_public_key(bbs_issuer* issuer, const uint8_t* pk_affine, int is_identity);
int bbs_issuer_set_secret_key(bbs_issuer* issuer, const uint8_t* sk32);          /* also sets pk = sk * BP2 */
/* max_messages: longest item served; window_bits: 0 = by free device memory, else 4..22 -- for contexts created afterwards */
int bbs_issuer_set_limits(bbs_issuer* issuer, size_t max_messages, int window_bits);
/* resident contexts: at most max_contexts (>= 1) and max_table_bytes (0 = half of the free device memory at the first
 * context) of tables; idle contexts beyond the new bounds leave at once.  bbs_issuer_table_bytes: what is resident now. */
int bbs_issuer_set_budget(bbs_issuer* issuer, size_t max_contexts, size_t max_table_bytes);
size_t bbs_issuer_table_bytes(bbs_issuer* issuer);
/* bbs_ctx_set_latency_mode / _set_batch_verification (seed from the operating system) / _set_points_in_subgroup on every
 * context of the issuer, present and future */
int bbs_issuer_set_modes(bbs_issuer* issuer, int latency_mode, int batch_verification, int points_in_subgroup);
/* the context serving items of `message_count` messages (created if needed), e.g. to warm it up before traffic arrives, or
 * to drive it directly (one call at a time, and not while the issuer routes lists to it or a bbs_issuer_set_* call is made).
 * A context handed out this way is never evicted, and bbs_issuer_set_public_key / _set_secret_key / _set_modes bring it up to
 * the new configuration BEFORE they return (the other contexts pick it up at their next use). */
int bbs_issuer_context(bbs_issuer* issuer, size_t message_count, bbs_ctx** out);
size_t bbs_issuer_context_count(bbs_issuer* issuer);
/* proof_verify (src/proof_verify.rs:19-61); message count of item i = (its proof's commitments) + (its disclosed indexes) */
int bbs_issuer_proof_verify(bbs_issuer* issuer, size_t n, const uint8_t* proof_octets, const uint64_t* oct_off,
                            const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                            const uint64_t* disclosed_idx, const uint64_t* didx_off,
                            const uint8_t* headers, const uint64_t* hdr_off,
                            const uint8_t* ph, const uint64_t* ph_off, int8_t* status);
/* verify (src/verify.rs:18-50), sign (src/sign.rs:32-60), proof_gen (src/proof_gen.rs:78-113); message count of item i =
 * its number of messages.  sign: fp_bytes + 32 octets per item (zeros where status != 1).  proof_gen: octet strings packed
 * in the caller's order, oct_off_out n + 1 byte offsets, octets_out needs sum_i (3 fp_bytes + 32 (4 + messages_i)) bytes. */
int bbs_issuer_verify(bbs_issuer* issuer, size_t n, const uint8_t* signature_octets,
                      const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                      const uint8_t* headers, const uint64_t* hdr_off, int8_t* status);
int bbs_issuer_sign(bbs_issuer* issuer, size_t n, const uint8_t* msg_bytes, const uint64_t* msg_byte_off,
                    const uint64_t* msg_item_off, const uint8_t* headers, const uint64_t* hdr_off,
                    uint8_t* signature_octets_out, int8_t* status);
int bbs_issuer_proof_gen(bbs_issuer* issuer, size_t n, const uint8_t* signature_octets,
                         const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                         const uint64_t* disclosed_idx, const uint64_t* didx_off,
                         const uint8_t* random_scalars, const uint64_t* rnd_off,
                         const uint8_t* headers, const uint64_t* hdr_off,
                         const uint8_t* ph, const uint64_t* ph_off,
                         uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status);

/* The asynchronous forms, for a serving loop that keeps several lists in flight: *_submit packs the groups, submits them and
 * returns; the inputs may be released at once, `status` and the output buffers must stay valid until bbs_issuer_job_wait,
 * which waits for every group and scatters the results into the caller's order (BBS_E_STATE if an item was left undecided).
 * bbs_issuer_job_free releases the job (after waiting for it if that has not happened); every job of an issuer must be freed
 * before bbs_issuer_destroy (it holds the issuer's contexts).  Arguments as the synchronous calls. */
typedef struct bbs_issuer_job bbs_issuer_job;
int bbs_issuer_proof_verify_submit(bbs_issuer* issuer, size_t n, const uint8_t* proof_octets, const uint64_t* oct_off,
                                   const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                                   const uint64_t* disclosed_idx, const uint64_t* didx_off,
                                   const uint8_t* headers, const uint64_t* hdr_off,
                                   const uint8_t* ph, const uint64_t* ph_off, int8_t* status, bbs_issuer_job** job_out);
int bbs_issuer_verify_submit(bbs_issuer* issuer, size_t n, const uint8_t* signature_octets,
                             const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                             const uint8_t* headers, const uint64_t* hdr_off, int8_t* status, bbs_issuer_job** job_out);
int bbs_issuer_sign_submit(bbs_issuer* issuer, size_t n, const uint8_t* msg_bytes, const uint64_t* msg_byte_off,
                           const uint64_t* msg_item_off, const uint8_t* headers, const uint64_t* hdr_off,
                           uint8_t* signature_octets_out, int8_t* status, bbs_issuer_job** job_out);
int bbs_issuer_proof_gen_submit(bbs_issuer* issuer, size_t n, const uint8_t* signature_octets,
                                const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                                const uint64_t* disclosed_idx, const uint64_t* didx_off,
                                const uint8_t* random_scalars, const uint64_t* rnd_off,
                                const uint8_t* headers, const uint64_t* hdr_off,
                                const uint8_t* ph, const uint64_t* ph_off,
                                uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status, bbs_issuer_job** job_out);
int bbs_issuer_job_wait(bbs_issuer_job* job);
void bbs_issuer_job_free(bbs_issuer_job* job);

/* ------------------------------------------------------------------------------------------
 * Host-side setup helpers: once per ciphersuite / key, no GPU involved.
 * ------------------------------------------------------------------------------------------ */
/* create_generators (src/utils/interface_utilities.rs:47-73) with the curve's hash-to-curve backend
 * (:24-44: BLS12-381 simplified SWU + 11-isogeny; BN254 Shallue-van de Woestijne, pinned by P1 of
 * src/constants.rs:39-51); out = count affine G1 points.  The reference recomputes this on
 * every sign / verify / proof_gen / proof_verify call; callers cache it per (api_id, count). */
int bbs_create_generators(int curve, size_t count, const uint8_t* api_id, size_t api_id_len,
                          uint8_t* out_affine);
/* HashToG1::hash_to_g1 (interface_utilities.rs:17-44), BLS12-381. */
int bbs_hash_to_g1(int curve, const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len,
                   uint8_t* out_affine);
/* FromOkm (src/utils/utilities_helper.rs:15-40): 48 big-endian bytes reduced mod r, 32 B LE out -- what
 * calculate_random_scalars (src/utils/core_utilities.rs:70-81) applies to 48 random bytes. */
int bbs_scalar_from_okm(int curve, const uint8_t* okm48, uint8_t* scalar_out);
/* SecretKey::key_gen (src/key_gen.rs:46-81): returns 0 or the KeyGenError code. */
int bbs_key_gen(int curve, const uint8_t* key_material, size_t key_material_len, const uint8_t* key_info,
                size_t key_info_len, const uint8_t* key_dst, size_t key_dst_len, uint8_t* sk32_out);

/* ------------------------------------------------------------------------------------------
 * Wire codec (host side).  Octet strings as in the reference's vectors (src/tests/test_vector.rs:
 * 163-260): signature = compress(A) || e, proof = compress(Abar) || compress(Bbar) || compress(D) ||
 * e^ || r1^ || r3^ || m^_1 .. m^_U || c, public key = compress(pk); scalars 32 B big-endian.
 * `*_from_octets` validate: canonical encodings, on curve, prime-order subgroup, and reject an
 * identity point / zero e where the BBS draft does.  Return 0 or a BBS_ST_* code.
 * (The reference derives ark-serialize's CanonicalSerialize/Deserialize for these types --
 * src/sign.rs:18, src/proof_gen.rs:29, src/key_gen.rs:12 -- without ever exercising them.)
 * ------------------------------------------------------------------------------------------ */
int bbs_signature_to_octets(int curve, const uint8_t* sig_record, uint8_t* out /* fp_bytes + 32 */);
int bbs_signature_from_octets(int curve, const uint8_t* octets, uint8_t* sig_record_out);
int bbs_proof_to_octets(int curve, const uint8_t* proof_fixed, const uint8_t* commitments, size_t n_commitments,
                        uint8_t* out /* 3 * fp_bytes + 32 * (4 + n_commitments) */);
int bbs_proof_from_octets(int curve, const uint8_t* octets, size_t len, uint8_t* proof_fixed_out,
                          uint8_t* commitments_out, size_t commitments_cap, size_t* n_commitments_out);
/* n compressed G1 points (fp_bytes each, the ciphersuite's encoding) -> affine records on the device: square root,
 * on-curve and prime-order-subgroup checks.  code[i]: 0 ok, 1 ok and the identity (record = zeros), BBS_ST_NONCANONICAL
 * malformed / not canonical, BBS_ST_NOT_ON_CURVE not on the curve or outside the subgroup. */
int bbs_g1_decompress_batch(bbs_ctx* ctx, size_t n, const uint8_t* compressed, uint8_t* out_affine, int8_t* code);
/* bbs_signature_from_octets for n signatures (fp_bytes + 32 bytes each) at once, the point work on the device;
 * status[i] = 1 or the code bbs_signature_from_octets returns for item i; records zeroed where status != 1. */
int bbs_signatures_from_octets_batch(bbs_ctx* ctx, size_t n, const uint8_t* octets, uint8_t* sig_records_out, int8_t* status);
/* bbs_proof_to_octets for n proofs at once (host; no field arithmetic: byte order and the sign flag only).  out needs
 * sum_i (3 * fp_bytes + 32 * (4 + U_i)) bytes; out_off: n + 1 byte offsets; status[i] = 1 or BBS_ST_NONCANONICAL. */
int bbs_proofs_to_octets_batch(int curve, size_t n, const uint8_t* proofs_fixed, const uint8_t* commitments,
                               const uint64_t* commit_off, uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status);
/* bbs_proof_from_octets for n proofs at once: octets ragged (oct_off: n + 1 byte offsets); the 3 n points are
 * decompressed and checked (on curve, prime-order subgroup) on the device, scalars on the host.  status[i] = 1 or the
 * code bbs_proof_from_octets returns for item i; proofs_fixed_out: n records (zeros where status != 1);
 * commitments_out / commit_off_out (n + 1 entries): the commitments of every well-formed item, packed. */
int bbs_proofs_from_octets_batch(bbs_ctx* ctx, size_t n, const uint8_t* octets, const uint64_t* oct_off,
                                 uint8_t* proofs_fixed_out, uint8_t* commitments_out, uint64_t* commit_off_out,
                                 int8_t* status);
int bbs_public_key_to_octets(int curve, const uint8_t* pk_affine, int is_identity, uint8_t* out /* 2 * fp_bytes */);
int bbs_public_key_from_octets(int curve, const uint8_t* octets, uint8_t* pk_affine_out, int* is_identity_out);

/* GPU self-test: one Fp12 operation (12 Fp values a, b, canonical LE, tower order) computed by the
 * one-lane code and by the six-lane wavefront-cooperative code; the caller compares the outputs.
 * op: 0 mul, 1..3 Frobenius^k, 4 inverse, 5 conjugate, 6 line multiplication, 7 final
 * exponentiation, 10 cyclotomic square, 11 power by the curve parameter. */
int bbs_selftest_f12(bbs_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint8_t* out_single,
                     uint8_t* out_dist);
/* Host arithmetic self-test (no GPU): out = sum_k w_k * a_k * b_k in Fp2 computed by the lazily reduced column
 * accumulators the pairing kernel uses (one Montgomery reduction pair per dot product).  a, b: n_terms records
 * c0 || c1 (canonical LE); weights[k] = 1, 2, or 0 for "b_k is its c0 in Fp"; total weight <= 6. */
/* Host arithmetic self-test (no GPU): x^-1 in the base field (scalar_field = 0, fp_bytes LE) or the scalar field
 * (1, 32 bytes LE) by the safegcd inversion the kernels use and by the Fermat power x^(p-2). */
int bbs_selftest_inv(int curve, int scalar_field, const uint8_t* x, uint8_t* out_safegcd, uint8_t* out_fermat);
/* Host arithmetic self-test (no GPU): 3 x0 + 2 x1 (plus = 1) or 3 x0 - 2 x1 (plus = 0) in the base field (fp_bytes LE,
 * canonical in and out) by the single reduction chain with a run-time sign that ends the cyclotomic square. */
int bbs_selftest_lin_pm(int curve, int plus, const uint8_t* x0, const uint8_t* x1, uint8_t* out);
/* Host arithmetic self-test (no GPU): the GLV split of a canonical scalar k (32 bytes LE) as the kernels compute it:
 * k = (+-k1) + (+-k2) * lambda mod r, k1 and k2 below 2^128 (16 bytes LE each), *neg = 1 for a negative half.
 * BLS12-381: lambda = x^2 - 1, k2 = floor(k / lambda), both halves non-negative; BN254: rounding against a short
 * lattice basis (lambda = the cube root of unity whose words are GLV_LAMBDA in params_gen.hpp). */
int bbs_selftest_glv_split(int curve, const uint8_t* k32, uint8_t* k1_16, uint8_t* k2_16, int* neg1, int* neg2);
/* Host arithmetic self-test (no GPU): k0 P0 + k1 P1 + k2 P2 by the joint windowed chain proof_verify uses for T1
 * (src/proof_verify.rs:163-164), plain (glv = 0) or with the GLV split (glv = 1; BLS12-381 points must be in the subgroup).  points: 3 affine records, scalars: 3 x 32 bytes LE canonical. */
int bbs_selftest_mul3(int curve, int glv, const uint8_t* points, const uint8_t* scalars, uint8_t* out_affine);
/* Host arithmetic self-test (no GPU): one half of an Fp4 square as the pairing kernel computes it (four limb-column
 * products, one reduction pair): hi = 0: a^2 + xi b^2, hi = 1: 2 a b, for a, b in Fp2 (c0 || c1, canonical LE);
 * hi | 2 (BLS12-381): the same through the two-pass column accumulators. */
int bbs_selftest_fp4sqr(int curve, int hi, const uint8_t* a, const uint8_t* b, uint8_t* out);
int bbs_selftest_f2dot(int curve, size_t n_terms, const uint8_t* a, const uint8_t* b, const uint8_t* weights,
                       uint8_t* out);
/* The same dot product by the two-pass column accumulators (low columns -> Montgomery quotients -> high columns; half the
 * accumulator registers): must equal bbs_selftest_f2dot limb for limb. */
int bbs_selftest_f2dot2(int curve, size_t n_terms, const uint8_t* a, const uint8_t* b, const uint8_t* weights,
                        uint8_t* out);

/* ------------------------------------------------------------------------------------------
 * Pool: core_proof_verify over a LIST of proofs, fanned out over SEVERAL GPUs behind this ABI ("multi-GPU
 * fan-out is internal"; 8(e): split by curve, contiguous ranges per GPU, tables replicated, no data-path collective).
 * The reference verifies one proof per call (src/proof_verify.rs:19-61, core form :64-116); a caller with a list loops over
 * it.  A pool owns one context per (curve, member device) with the same generators and issuer key on every member; a list
 * is handed over as one section per curve in the layout of bbs_core_proof_verify_batch.  Every section is cut into
 * contiguous shares of ceil(n / members) items, every share into jobs of at most max_batch items, and the jobs of a member
 * run from ONE submitting thread per member (owned by the pool) through bbs_core_proof_verify_submit (completion-order
 * retire, curves alternating).  The statuses land in the caller's array in the caller's order: one process drives every GPU, so no
 * exchange between processes exists on this path (the one-process-per-GPU launcher of bench.py exchanges them with one
 * all_gather; both partition by the same rule).  Statuses of every item are exactly those of bbs_core_proof_verify_batch
 * on the item's own curve.
 * ------------------------------------------------------------------------------------------ */
typedef struct bbs_pool bbs_pool;
typedef struct bbs_pv_list {      /* the items of ONE curve of a list, as for bbs_core_proof_verify_batch */
    int curve;
    size_t n;
    const uint8_t* proofs_fixed;
    const uint8_t* commitments;    const uint64_t* commit_off;
    const uint8_t* disclosed_msgs; const uint64_t* dmsg_off;
    const uint64_t* disclosed_idx; const uint64_t* didx_off;
    const uint8_t* headers;        const uint64_t* hdr_off;     /* hdr_off / ph_off may be NULL: all empty */
    const uint8_t* ph;             const uint64_t* ph_off;
    const uint64_t* global_index;  /* NULL: status[k] is item k of this section.  Else: item k is item global_index[k] of
                                    * the caller's whole (mixed) list and its status is written to status[global_index[k]] */
    int8_t* status;
} bbs_pv_list;
/* device_ids: the member devices, in share order; an id may repeat (two context sets on one GPU -- what a test on a
 * one-GPU box uses).  BBS_E_NO_DEVICE if an id does not exist. */
int bbs_pool_create(const int* device_ids, size_t n_devices, bbs_pool** out);
void bbs_pool_destroy(bbs_pool* pool);
size_t bbs_pool_device_count(const bbs_pool* pool);
/* the same configuration on every member's context of `curve` (contexts are created at the first call naming the curve) */
int bbs_pool_set_window_bits(bbs_pool* pool, int curve, int bits);
int bbs_pool_set_generators(bbs_pool* pool, int curve, const uint8_t* generators, size_t count, const uint8_t* api_id,
                            size_t api_id_len);
int bbs_pool_set_public_key(bbs_pool* pool, int curve, const uint8_t* pk_affine, int is_identity);
/* jobs outstanding per member (default 6, the plateau of the single-GPU serving loop) */
int bbs_pool_set_inflight(bbs_pool* pool, int jobs_per_member);
/* member `member`'s context of `curve`, e.g. to run any other operation of this ABI on that device */
int bbs_pool_context(bbs_pool* pool, int curve, size_t member, bbs_ctx** out);
/* Verify a list: the sections are cut into jobs and queued to the members' submitting threads (one per member, inside the
 * library); the call returns at once.  bbs_pool_job_wait blocks until every status of THIS list has been written and returns
 * the first failure of any member (BBS_OK otherwise); then bbs_pool_job_free.  Several lists may be in flight: a member does
 * not drain between lists -- the jobs of the next one go in while the last jobs of this one finish -- and lists complete in
 * submission order per member.  The caller's input buffers AND the bbs_pv_list array's contents must stay valid until
 * bbs_pool_job_wait has returned (unlike bbs_core_proof_verify_submit, the jobs are staged later, by the members' threads).
 * max_batch = 0: 4096.  BBS_E_STATE if a section's curve has no generators / public key yet; BBS_E_ARG before anything is
 * queued if a section is malformed.  The bbs_pool_set_* calls are BBS_E_STATE while a list is in flight.  Free every job
 * before bbs_pool_destroy. */
typedef struct bbs_pool_job bbs_pool_job;
int bbs_pool_proof_verify_submit(bbs_pool* pool, const bbs_pv_list* lists, size_t n_lists, size_t max_batch,
                                 bbs_pool_job** job_out);
int bbs_pool_job_wait(bbs_pool_job* job);
void bbs_pool_job_free(bbs_pool_job* job);
/* submit + wait + free */
int bbs_pool_proof_verify(bbs_pool* pool, const bbs_pv_list* lists, size_t n_lists, size_t max_batch);

#ifdef __cplusplus
}
#endif
#endif /* BBS_SIGN_AMD_H */
