/*
 * halo2_mi355x.h -- C ABI of libhalo2_mi355x.so: the MI355X (gfx950) backend for the two
 * arithmetic kernels that dominate halo2_proofs::plonk::create_proof under KZG on BN256.
 *
 * What it replaces.  The reference (summa-dev/halo2-experiments) reaches this path only through
 * full_prover (/root/reference/src/circuits/utils.rs:22-70: ParamsKZG::setup :28, keygen_vk :31,
 * keygen_pk :35, create_proof :40-48), which runs the free functions of the halo2_proofs crate
 * pinned at /root/reference/Cargo.toml:10 (git tag v2023_02_02; source not vendored):
 *
 *     pub fn best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve
 *     pub fn best_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32)
 *
 * instantiated with C = bn256::G1Affine and G = bn256::Fr.  Those crates have no FFI; the
 * boundary below is what a `[patch]`-ed halo2_proofs::arithmetic would bind (INTEGRATION.md
 * shows the Rust side).  Each entry point names the upstream function it stands in for.
 *
 * Memory formats (halo2curves bn256) -- exactly the bytes Rust holds:
 *   Fr / Fq   4 x u64 little-endian limbs, Montgomery form (v * 2^256 mod m), fully reduced
 *   G1Affine  { x: Fq, y: Fq } = 8 x u64; the identity is (0, 0)
 *   G1        Jacobian { x, y, z: Fq } = 12 x u64; the identity has z = 0
 *
 * Conventions: every function returns HM_OK (0) or a negative HM_ERR_* code and never aborts or
 * throws across the boundary; hm_last_error() returns the message of the calling thread's last
 * failure.  Pointers are borrowed for the duration of the call only.  Calls are thread-safe (one
 * lock per device).  There is no CPU fallback inside this library: without a usable gfx950
 * device every compute entry point fails with HM_ERR_NO_DEVICE.
 */
#ifndef HALO2_MI355X_H
#define HALO2_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HM_OK 0
#define HM_ERR_BAD_ARG (-1)
#define HM_ERR_NO_DEVICE (-2)
#define HM_ERR_HIP (-3)
#define HM_ERR_NOT_FOUND (-4)
#define HM_ERR_INTERNAL (-5)
/* an in-place host-pointer call failed AFTER it began writing its result: the array is undefined (every other error
 * leaves the caller's arrays untouched, so a caller may fall back to its CPU body on them -- not after this one) */
#define HM_ERR_PARTIAL_OUTPUT (-6)

/* ---- lifecycle ---------------------------------------------------------------------------- */

/* Select the HIP device this thread's subsequent calls use (default: the current HIP device).
 * One process per GPU is the intended deployment; a single process may also drive several. */
int hm_set_device(int device);
/* Number of visible HIP devices (0 when there is none); never fails. */
int hm_device_count(void);
/* Free every device buffer, twiddle table and registered base set of the current device. */
int hm_shutdown(void);
/* Message of the last failure on the calling thread ("" if none). */
const char* hm_last_error(void);
/* Library version string. */
const char* hm_version(void);

/* ---- MSM: stands in for halo2_proofs::arithmetic::best_multiexp::<bn256::G1Affine> --------- */

/* sum_i scalars[i] * bases[i].  scalars: n x 4 u64 (Fr), bases: n x 8 u64 (G1Affine), both host
 * memory.  Result as an affine point (out_xy, Montgomery) plus an identity flag, which is the
 * canonical form of the G1 value best_multiexp returns; n == 0 yields the identity.
 * The scalars cross PCIe in every call.  The converted bases of the previous call are reused only when a
 * digest of the WHOLE base array (every word, 256 bits) and its length match -- never by pointer: a buffer
 * mutated at any index, or re-allocated at the same address with other contents, is uploaded again
 * (hm_set_host_base_cache(0) turns the reuse off altogether).  A caller that owns a long-lived base array --
 * ParamsKZG::g / g_lagrange -- still does best to register it once (below): that skips the digest too.
 * Largest call (every MSM form, per device): n * windows < 2^31 (point, window) item slots -- 2^27 points (12 windows
 * with the fixed-base table, 15 without); above that HM_ERR_BAD_ARG with a message, nothing computed.  A larger sum is
 * the hm_g1_sum of several calls over index ranges, or a split over devices (hm_set_msm_devices).  BASELINE's largest
 * configuration is 2^26. */
int hm_msm_bn256_g1(const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t out_xy[8], int* out_is_identity);

/* Same, Jacobian output (x, y, 1) / (0, 0, 0): the `C::Curve` value itself. */
int hm_msm_bn256_g1_jacobian(const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t out_xyz[12]);

/* Device-resident base sets (ParamsKZG::g / g_lagrange live for the whole proof):
 * upload + convert once, then run MSMs against bases[offset .. offset + n).
 * hm_release_bases never waits for the device: a set that an un-awaited hm_msm_submit_dev ticket still
 * reads is freed when that ticket is awaited, and the buffers of a released set are recycled by the next
 * registration of the same size. */
int hm_register_bases(const uint64_t* bases, size_t n, uint64_t* out_handle);
int hm_release_bases(uint64_t handle);
int hm_msm_bn256_g1_h(uint64_t handle, size_t offset, const uint64_t* scalars, size_t n, uint64_t out_xy[8],
                      int* out_is_identity);

/* Fixed-base variant for long-lived SRS sets (in halo2 the SRS is fixed for the life of the process:
 * /root/reference/src/circuits/utils.rs:28): additionally stores 2^(offset of window j) * P_i for every window j
 * (W x the memory: 12 GiB for 2^24 points -- sized for 288 GB of HBM; built once, ~0.22 s at 2^24), so that all
 * windows share ONE bucket set, the window grows from 17 to 22 bits and a fifth of the mixed additions disappears
 * (2^24 points: 19.5 ms against 21.5).  MSMs that cover the whole set (offset 0, n = set size) use the table; other
 * slices run the plain path on the same handle.  Results are identical either way. */
int hm_register_bases_precomp(const uint64_t* bases, size_t n, uint64_t* out_handle);
int hm_register_bases_precomp_dev(const void* d_bases, size_t n, void* stream, uint64_t* out_handle);
/* hm_register_bases / hm_register_bases_dev build that table BY THEMSELVES for sets of at least 2^log2_n points
 * (default 17: measured gains 4 % at 2^17, 13 % per dense commitment of a phase at 2^18, 3 % at 2^20 .. 2^22, 5 % at
 * 2^23, 9-15 % at 2^24, 15 % at 2^26; 0 = never, the plain layout only -- one copy of the points).  A phase of
 * commitments still sends its SPARSE columns (few surviving 256-row blocks) through the plain copy's five-launch plan.
 * Process-wide; read at registration time.  The environment variable HALO2_MI355X_FIXED_BASE_FROM_LOG sets the
 * initial value. */
int hm_set_fixed_base_threshold(uint32_t log2_n);
/* The PLAIN layout whatever the threshold says: one copy of the points, never a table.  For TRANSIENT sets -- a caller
 * that registers, runs one MSM and releases (the tensor form of the Python mirror's best_multiexp; a verifier's
 * commitments): building the table costs about ten MSMs of the same size (0.22 s and 12 GiB at 2^24 points). */
int hm_register_bases_plain(const uint64_t* bases, size_t n, uint64_t* out_handle);
int hm_register_bases_plain_dev(const void* d_bases, size_t n, void* stream, uint64_t* out_handle);

/* What a registration ended up with.  hm_register_bases / _dev fall back to the plain layout when the W copies of the
 * table cannot be allocated (after giving the parked buffers of released sets back to the allocator and trying once
 * more): the call still succeeds, table_windows == 0 tells, and default_tables_dropped counts such registrations of
 * the process.  A multi-device handle (hm_set_msm_devices) reports sums over its parts. */
typedef struct hm_bases_info {
  uint64_t n;                      /* points of the set */
  uint64_t device_bytes;           /* HBM the set holds (all devices) */
  uint64_t parked_bytes;           /* buffers of RELEASED sets kept for reuse on the set's device(s) (at most 2 GiB each) */
  uint64_t default_tables_dropped; /* process-wide: default registrations that fell back to the plain layout */
  uint32_t table_windows;          /* 0: plain layout; else the W of the fixed-base table */
  uint32_t table_window_bits;      /* the table's widest window */
  uint32_t devices;                /* 1, or the parts of a multi-device handle */
  uint32_t sliced;                 /* multi-device: 1 = index-range slices, 0 = replicas */
} hm_bases_info;
int hm_get_bases_info(uint64_t handle, hm_bases_info* out);

/* Device-pointer forms (inputs already in HBM; `stream` is a hipStream_t or NULL).  The result is
 * written to host memory, so the call synchronises `stream` before returning. */
int hm_register_bases_dev(const void* d_bases, size_t n, void* stream, uint64_t* out_handle);
int hm_msm_bn256_g1_dev(uint64_t handle, size_t offset, const void* d_scalars, size_t n, void* stream,
                        uint64_t out_xyz[12]);

/* Asynchronous form: enqueue the whole MSM on `stream` and return a ticket at once (up to 8 MSMs in
 * flight per device, each with its own workspace); hm_msm_wait blocks on that MSM only, folds and
 * returns its result.  Independent commitments issued on different streams overlap: one MSM's
 * latency-bound phases (sort, bucket reduction, host fold) hide behind another's accumulation. */
int hm_msm_submit_dev(uint64_t handle, size_t offset, const void* d_scalars, size_t n, void* stream, uint64_t* out_ticket);
int hm_msm_wait(uint64_t ticket, uint64_t out_xyz[12]);
/* The commitments of one prover phase in one call (create_proof commits its advice / lookup / permutation columns in
 * loops of independent commit_lagrange calls): `count` scalar arrays of n elements each (d_scalars: host array of device
 * pointers, produced on `stream`) against bases[offset .. offset + n); out_xyz: count x 12 u64, in call order.  The library
 * keeps eight MSMs in flight on streams of its own; the call returns when all results are on the host. */
int hm_msm_batch_bn256_g1_dev(uint64_t handle, size_t offset, const void* const* d_scalars, size_t n, size_t count, void* stream,
                              uint64_t* out_xyz);
/* The same with the scalar arrays in HOST memory (scalars: host array of `count` host pointers, n x 4 u64 each) -- what a
 * prover that still keeps its polynomials in host vectors calls per phase: every chain's upload runs on its own stream,
 * so the PCIe time of one commitment hides behind the kernels of the others. */
int hm_msm_batch_bn256_g1_h(uint64_t handle, size_t offset, const uint64_t* const* scalars, size_t n, size_t count, uint64_t* out_xyz);

/* Single-process multi-GPU best_multiexp (the form a Rust prover, one process for the whole node,
 * binds): after hm_set_msm_devices(devices, count >= 2) every hm_msm_bn256_g1 /
 * hm_msm_bn256_g1_jacobian call splits [0, n) into `count` contiguous index ranges, runs one range per
 * listed device from its own host thread (each device caches its slice of the SRS exactly as the
 * one-device path does) and folds the 96-byte partial results on the host -- the data path has no
 * inter-GPU exchange.  Inputs below 2^14 points per device go whole to devices[0].  count == 0
 * restores the default (the calling thread's current device).  Replaces the same call sites as
 * hm_msm_bn256_g1 (halo2_proofs::arithmetic::best_multiexp). */
int hm_set_msm_devices(const int* devices, int count);

/* Sum of `count` G1 values (12 x u64 each, z = 0 for the identity), normalised to (x, y, 1): the
 * fold of per-GPU partial results of a sharded best_multiexp after the RCCL all-gather.  Pure host
 * arithmetic on a handful of points (the exchange is ~96 B per rank); needs no device. */
int hm_g1_sum(const uint64_t* points_xyz, size_t count, uint64_t out_xyz[12]);

/* enable != 0 (default): hm_msm_bn256_g1* may reuse the previous call's converted bases when the full-content digest
 * matches; 0: always upload and convert. */
int hm_set_host_base_cache(int enable);

/* ---- device memory for callers without a HIP binding of their own -------------------------------------------------
 * Everything a Rust (or C) prover needs to keep its polynomials in HBM between the steps of create_proof and use the *_dev
 * entry points with nothing but this library (rust/halo2_proofs-patch/src/mi355x_dev.rs: DevicePoly).  All on the calling
 * thread's current device (hm_set_device).
 *   hm_device_malloc / hm_device_free: hipMalloc / hipFree (free waits for the device: keep buffers for the life of a proof);
 *   hm_copy_to_device / hm_copy_to_host: synchronous, through the same copy policy as the host-pointer forms (hm_set_host_copies).
 *     Ordered behind everything queued on the device's default stream and on blocking streams (the call waits for them, whichever
 *     way the bytes travel); work on NON-blocking streams of the caller's own needs the caller's synchronisation first
 *     (hm_device_synchronize() waits for all streams).  stream == NULL in the *_dev entry points is the device's default stream. */
int hm_device_malloc(size_t bytes, void** d_out);
int hm_device_free(void* d_ptr);
int hm_copy_to_device(void* d_dst, const void* src, size_t bytes);
int hm_copy_to_host(void* dst, const void* d_src, size_t bytes);
/* `count` arrays in ONE call: srcs[i] -> d_dsts[i] (resp. d_srcs[i] -> dsts[i]), bytes[i] each.  The copy lanes treat them as one
 * transfer -- the columns of a proof (48 x 8 MiB at k = 18) move at the rate of one 384 MiB array instead of 48 small ones, each with
 * its own helper threads, pipeline fill and drain.  Same ordering and error convention as the single forms; a failure of the
 * _to_host form may leave any of the destinations partly written (HM_ERR_PARTIAL_OUTPUT). */
int hm_copy_many_to_device(void* const* d_dsts, const void* const* srcs, const size_t* bytes, size_t count);
int hm_copy_many_to_host(void* const* dsts, const void* const* d_srcs, const size_t* bytes, size_t count);
int hm_device_synchronize(void);

/* How the host-pointer forms move their arrays (csrc/xfer.hip) -- a rule on the host RANGE, never on a timing.
 * 0 = auto (the default; HALO2_MI355X_HOST_COPIES sets the start value): a range the caller has declared long-lived with
 *     hm_host_register goes straight to hipMemcpy (a DMA from registered memory); every other range of 256 KiB or more goes through the
 *     library's own pinned staging lanes -- nothing of the caller's memory is handed to the driver, so the cost does not depend on what
 *     pinning a page costs on the box (0.1 us or 9 us: both exist in one pool) nor on what the caller maps and unmaps around the calls;
 * 1 = lanes always (registered ranges too); 2 = direct always (hipMemcpy pins the caller's pages on the fly: 0.2 ms per 64 MiB
 *     faster where pinning is cheap, 2.3 ms per MiB slower where it is not -- for boxes known to pin fast).  Process-wide. */
int hm_set_host_copies(int mode);
/* Declare [p, p + bytes) long-lived host memory (an SRS kept in memory, a staging buffer reused for every proof): pinned ONCE here
 * (hipHostRegister, portable across devices), after which the host-pointer forms and hm_copy_to_* DMA from / into it directly.  The
 * range must stay mapped until hm_host_unregister(p) (p = the start of a registered range); ranges must not overlap.  Needs a
 * device (the registration is the runtime's).  Not for arrays that live for one call: registering costs what the lanes avoid. */
int hm_host_register(const void* p, size_t bytes);
int hm_host_unregister(const void* p);

/* Window-size override for experiments (0 = automatic). */
int hm_msm_set_window(int c);
/* Per-phase timing of an MSM (digits / sort / accumulate / reduce and the accumulate kernel alone, reported by
 * hm_get_msm_stats) costs seven event records per call.  mode -1 (default): the general pipeline (n >= 2^19 and
 * precomputed sets) records them, the five-launch plan of prover sizes records only the total; 0: never; 1: always. */
int hm_msm_set_phase_timing(int mode);

/* ---- NTT: stands in for halo2_proofs::arithmetic::best_fft::<bn256::Fr> --------------------- */

/* In place on host memory: a[j] <- sum_i a[i] * omega^(i*j), natural order in and out, unscaled.
 * a: 2^log_n x 4 u64 (Fr); omega: 4 u64 (Fr), a 2^log_n-th root of unity; log_n <= 28.
 * `a` is written only by the final copy from the device: every error code but HM_ERR_PARTIAL_OUTPUT (that copy itself
 * failed) leaves it exactly as it was. */
int hm_ntt_bn256_fr(uint64_t* a, const uint64_t omega[4], uint32_t log_n);

/* Device-pointer form, in place, asynchronous on `stream`.  Every *_dev NTT entry point may be called
 * concurrently on different streams (and from different threads): the ping-pong buffer of a multi-pass
 * transform belongs to the stream in use (a small pool, handed over behind an event), fused constants
 * travel to the kernels by value, and twiddle tables built on one stream are published to the others
 * behind an event. */
int hm_ntt_bn256_fr_dev(void* d_a, const uint64_t omega[4], uint32_t log_n, void* stream);

/* `batch` back-to-back arrays of 2^log_n elements transformed by ONE set of launches (the ~48
 * same-size transforms of a proof).  scale (4 words) and coset (12 words) are
 * optional (NULL): the fused EvaluationDomain::ifft divisor and the coeff_to_extended coset shift. */
int hm_ntt_batch_bn256_fr_dev(void* d_a, size_t batch, const uint64_t omega[4], uint32_t log_n, const uint64_t* scale,
                              const uint64_t* coset, void* stream);

/* ---- next rows: the EvaluationDomain steps either side of best_fft ----------- */

/* EvaluationDomain::ifft: best_fft(a, omega_inv, log_n) then a[i] *= divisor, with the scaling
 * fused into the last NTT pass.  Device pointers, in place. */
int hm_ifft_bn256_fr_dev(void* d_a, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4], void* stream);
/* EvaluationDomain::coeff_to_extended's arithmetic: a[i] *= coset[i % 3] (distribute_powers_zeta
 * with coset = {1, g, g^2}) fused into the first pass of best_fft(a, omega_ext, log_n).
 * The caller has already zero-padded a to 2^log_n. */
int hm_coset_ntt_bn256_fr_dev(void* d_a, const uint64_t omega[4], uint32_t log_n, const uint64_t coset[12], void* stream);
/* EvaluationDomain::coeff_to_extended in one call (halo2_proofs poly/domain.rs: resize to the extended
 * domain with zeros, distribute_powers_zeta, best_fft with extended_omega): `batch` compact coefficient
 * arrays of 2^log_n x 32 B at d_coeffs -> `batch` arrays of 2^log_ext evaluations at d_ext (out of
 * place, d_ext need not be initialised).  coset = {1, zeta, zeta^2} as for hm_coset_ntt_bn256_fr_dev,
 * or NULL.  The zero-padded part is never materialised: the first pass reads only the coefficients, and
 * the log_ext - log_n butterfly stages that would pair them with zeros cost nothing. */
int hm_coeff_to_extended_bn256_fr_dev(const void* d_coeffs, void* d_ext, size_t batch, const uint64_t extended_omega[4],
                                      uint32_t log_n, uint32_t log_ext, const uint64_t* coset, void* stream);
/* The extended domain one coset of the n-th roots of unity at a time.  The 2^log_ext points zeta * extended_omega^m that
 * coeff_to_extended evaluates on fall into E = 2^(log_ext - log_n) cosets shift_j * <omega>, shift_j = zeta *
 * extended_omega^j, omega = extended_omega^E: row m = E t + j of the extended array is point t of coset j.  A rotation of
 * the circuit (X -> omega^rot X) stays inside a coset, and 1 / (X^n - 1) is ONE constant on it, so evaluate_h runs on each
 * coset by itself (hm_graph_evaluate_dev over 2^log_n rows, rotations unscaled) -- E independent n-point problems that can
 * go to E different GPUs, each needing only the coefficient arrays (later versions of halo2_proofs take the same route
 * for memory: coeff_to_extended_part).
 *   hm_coeff_to_coset:  d_out_b[t] = sum_i d_coeffs_b[i] (shift omega^t)^i for `batch` back-to-back arrays of 2^log_n Fr
 *     (= row E t + j of hm_coeff_to_extended's output, bit for bit); d_out may be d_coeffs itself.  columns_internal != 0:
 *     outputs multiplied by 32, the form HM_GRAPH_COLUMNS_INTERNAL loads.  The powers of `shift` are kept on the device
 *     per (shift, log_n).
 *   hm_coset_to_coeff:  in place, a_b <- divisor * (inverse transform with omega_inv) of a_b, then a_b[i] *= shift_inv^i:
 *     with divisor = 1/n this is d_j[i] = sum_q h[i + q n] zeta^(n q) w^(j q) (w = extended_omega^n, a primitive E-th
 *     root of unity) for the coefficients h of the polynomial whose values on coset j went in; the E of them recombine as
 *     h[i + q n] = zeta^(-n q) / E * sum_j w^(-j q) d_j[i] (hm_fr_linear_combination_dev, E terms per q).
 * Asynchronous on `stream`. */
int hm_coeff_to_coset_bn256_fr_dev(const void* d_coeffs, void* d_out, size_t batch, const uint64_t omega[4], uint32_t log_n,
                                   const uint64_t shift[4], int columns_internal, void* stream);
int hm_coset_to_coeff_bn256_fr_dev(void* d_a, size_t batch, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4],
                                   const uint64_t shift_inv[4], void* stream);
/* Several cosets in one launch chain (at most 16 per call).
 *   hm_coeff_to_cosets: d_out holds batch * count arrays of 2^log_n Fr; array b * count + c = coefficient array b evaluated
 *     on the coset shifts[c] * <omega> -- per input the `count` cosets lie one after the other, which is the column layout
 *     hm_graph_evaluate_segments_dev reads (segment c = coset c).  shifts: host, count x 4 u64.  d_out must not overlap d_coeffs.
 *   hm_cosets_to_coeff: in place on `count` back-to-back arrays (the values of ONE polynomial on `count` cosets): array c
 *     <- divisor * inverse transform, then element i *= shift_invs[c]^i.
 * Only j - 1 of the E cosets are needed for the quotient of a satisfied circuit (it has fewer than n (j - 1) coefficients):
 * partial_c[i] = sum_t h[i + t n] u_c^t with u_c = shifts[c]^n is a (j - 1) x (j - 1) Vandermonde system per i, whose
 * inverse matrix -- times 1 / (u_c - 1), the vanishing polynomial's inverse on coset c, if the numerator was evaluated
 * undivided -- gives every piece of h as ONE hm_fr_linear_combination_dev of the partials. */
int hm_coeff_to_cosets_bn256_fr_dev(const void* d_coeffs, void* d_out, size_t batch, const uint64_t omega[4], uint32_t log_n,
                                    const uint64_t* shifts, size_t count, int columns_internal, void* stream);
int hm_cosets_to_coeff_bn256_fr_dev(void* d_a, size_t count, const uint64_t omega_inv[4], uint32_t log_n, const uint64_t divisor[4],
                                    const uint64_t* shift_invs, void* stream);

/* EvaluationDomain::extended_to_coeff's arithmetic in one call (halo2_proofs poly/domain.rs: ifft over the
 * extended domain, then distribute_powers_zeta with the INVERSE coset powers): `batch` arrays of 2^log_ext
 * evaluations, in place; best_fft(a, extended_omega_inv, log_ext), then a[i] *= divisor * coset_inv[i % 3]
 * with coset_inv = {1, zeta^-1, zeta^-2} -- divisor and pattern are folded into three constants that the
 * LAST NTT pass multiplies in, so no separate sweep over the 2^log_ext elements remains.  The caller
 * truncates to n * (j - 1) coefficients. */
int hm_extended_to_coeff_bn256_fr_dev(void* d_a, size_t batch, const uint64_t extended_omega_inv[4], uint32_t log_ext,
                                      const uint64_t divisor[4], const uint64_t coset_inv[12], void* stream);

/* Host-pointer forms of the two steps above, for a prover whose polynomials stay in host memory (the drop-in patch of
 * rust/halo2_proofs-patch: EvaluationDomain::coeff_to_extended / extended_to_coeff, halo2_proofs poly/domain.rs).  Only what
 * upstream's arrays really hold crosses PCIe:
 *   hm_coeff_to_extended_bn256_fr: coeffs = 2^log_n x 4 u64 in (the zero padding upstream's `resize` appends is never uploaded),
 *     ext = 2^log_ext x 4 u64 out; coset = {1, zeta, zeta^2} (12 words) or NULL.  `ext` is written by the final copy alone and
 *     may be the allocation `coeffs` lives in (upstream resizes its Vec in place): every error code but HM_ERR_PARTIAL_OUTPUT
 *     leaves both arrays as they were.
 *   hm_extended_to_coeff_bn256_fr: a = 2^log_ext x 4 u64 evaluations in; the first `keep` coefficients (keep <= 2^log_ext:
 *     upstream truncates to n * (j - 1)) come back into a[0 .. keep), the rest of `a` keeps its old contents -- the caller
 *     truncates.  Same error contract.
 * Synchronous; run on the calling thread's current device (the library's own stream). */
int hm_coeff_to_extended_bn256_fr(const uint64_t* coeffs, uint64_t* ext, const uint64_t extended_omega[4], uint32_t log_n,
                                  uint32_t log_ext, const uint64_t* coset);
int hm_extended_to_coeff_bn256_fr(uint64_t* a, size_t keep, const uint64_t extended_omega_inv[4], uint32_t log_ext,
                                  const uint64_t divisor[4], const uint64_t coset_inv[12]);

/* halo2_proofs::arithmetic::eval_polynomial (the Horner evaluations create_proof makes of every committed
 * polynomial at x * omega^rot): out[q] = sum_i poly_q[i] * points[q]^i for q < count, where poly_q is the
 * coefficient array number poly_index[q] (or q when poly_index is NULL) of the back-to-back arrays of n Fr at
 * d_polys.  points and out are host memory (count x 4 u64); the call synchronises `stream`. */
int hm_eval_polynomial_bn256_fr_dev(const void* d_polys, size_t n, const uint32_t* poly_index, const uint64_t* points, size_t count,
                                    uint64_t* out, void* stream);

/* halo2_proofs::arithmetic::kate_division(a, z): the quotient of a(X) - a(z) by X - z, which multiopen builds for
 * every opening (upstream poly/kzg/multiopen; reached from /root/reference/src/circuits/utils.rs:40-48).  d_poly: n
 * coefficients, d_quotient: n - 1 coefficients (q[i] = a[i+1] + z q[i+1]); the two must not overlap.  n < 2 writes
 * nothing.  Asynchronous on `stream`. */
int hm_kate_division_bn256_fr_dev(const void* d_poly, size_t n, const uint64_t z[4], void* d_quotient, void* stream);

/* `count` such divisions in one launch chain (the quotients one multiopen round builds are independent of each other):
 * d_polys / d_quotients: host arrays of `count` device pointers (n resp. n - 1 coefficients each), z: host, count x 4
 * u64, one point per polynomial.  No quotient may overlap any polynomial of the call.  Same results as `count` calls of
 * the form above.  Asynchronous on `stream`. */
int hm_kate_division_batch_bn256_fr_dev(const void* const* d_polys, size_t n, const uint64_t* z, void* const* d_quotients, size_t count,
                                        void* stream);

/* The running products of the permutation and lookup arguments (upstream plonk/permutation/prover.rs and
 * plonk/lookup/prover.rs: z[0] = start, z[row] = z[row - 1] * factors[row - 1]): d_out[i] = start * prod_{j < i}
 * d_factors[j] for i < n.  d_out may be d_factors itself.  Asynchronous on `stream`. */
int hm_fr_grand_product_dev(const void* d_factors, size_t n, const uint64_t start[4], void* d_out, void* stream);

/* The z columns of one argument in one launch chain.  d_factors / d_out: host arrays of `count` device pointers (n
 * elements each; d_out[j] may be d_factors[j] itself, and must not overlap any other column of the call).
 *   chain_row >= n (HM_NO_CHAIN): every column starts from `start` -- the lookup arguments' products (upstream
 *     plonk/lookup/prover.rs, one z per lookup, each from 1);
 *   chain_row <  n: column 0 starts from `start` and column j + 1 from d_out[j][chain_row] -- upstream's `last_z`
 *     (plonk/permutation/prover.rs: the z of a column set starts where the set before stood at the last usable row,
 *     chain_row = n - (blinding_factors + 1)); nothing is read back between the columns.
 * Same results as `count` calls of the form above made in order.  Asynchronous on `stream`. */
#define HM_NO_CHAIN ((size_t)-1)
int hm_fr_grand_product_batch_dev(const void* const* d_factors, size_t n, const uint64_t start[4], size_t chain_row, void* const* d_out,
                                  size_t count, void* stream);

/* ff::BatchInvert::batch_invert on n device-resident elements, in place: every non-zero element is replaced by its
 * inverse, zero stays zero (the denominators of the grand products above).  Asynchronous on `stream`. */
int hm_fr_batch_invert_dev(void* d_values, size_t n, void* stream);

/* d_out[i] = sum_{j < count} coeffs[j] * d_polys[j][i] for i < n: `Polynomial * scalar` and `+` of upstream
 * poly.rs in one pass (the random linear combinations of multiopen, the pieces of h(X)).  d_polys: host array of
 * `count` device pointers (n x 4 u64 each); coeffs: host, count x 4 u64.  d_out may be one of the inputs (the same
 * pointer, not a shifted view).  count = 0 zeroes d_out.  Asynchronous on `stream`. */
int hm_fr_linear_combination_dev(const void* const* d_polys, const uint64_t* coeffs, size_t count, size_t n, void* d_out,
                                 void* stream);

/* The permuted columns of one lookup argument (upstream plonk/lookup/prover.rs: permute_expression_pair): from the first
 * `rows` (= usable rows) entries of the compressed input and table columns, d_permuted_input[0 .. rows) = the input values
 * sorted by their canonical integers, d_permuted_table[0 .. rows) = the table values arranged so that every row where the
 * sorted input changes holds that input value and the leftover table values fill the repeated rows (ascending values from
 * the last such row backwards, exactly as upstream walks them).  The blinding rows beyond `rows` are the caller's.  The
 * outputs must not overlap the inputs.  Returns HM_ERR_NOT_FOUND when an input value does not occur in the table
 * (upstream: Error::ConstraintSystemFailure).  Synchronises `stream`. */
int hm_lookup_permute_bn256_fr_dev(const void* d_input, const void* d_table, size_t rows, void* d_permuted_input,
                                   void* d_permuted_table, void* stream);
/* The same for the `count` lookup arguments of a circuit at once (create_proof maps commit_permuted over them): one launch
 * chain carries up to eight lookups, so their 256-bit sorts run side by side.  Host arrays of `count` device pointers;
 * missing (optional, count ints) is set to 1 for every lookup whose input holds a value its table lacks. */
int hm_lookup_permute_batch_bn256_fr_dev(const void* const* d_inputs, const void* const* d_tables, size_t count, size_t rows,
                                         void* const* d_permuted_inputs, void* const* d_permuted_tables, int* missing, void* stream);

/* d_a[i] *= pattern[i mod period] for i < n, in place: EvaluationDomain::divide_by_vanishing_poly (upstream poly/domain.rs:
 * on the extended coset 1 / (X^n - 1) takes only 2^(extended_k - k) values, t_evaluations).  pattern: host, period x 4 u64;
 * period a power of two <= 64.  Asynchronous on `stream`. */
int hm_fr_mul_periodic_dev(void* d_a, size_t n, const uint64_t* pattern, uint32_t period, void* stream);

/* out[i] = x^i for i < n (device pointer, n x 4 u64): the ladder 1, s, s^2, ... of ParamsKZG::setup, whose
 * fixed-base multiples are g, and whose scaled inverse NTT gives the Lagrange-basis scalars of g_lagrange. */
int hm_fr_powers_dev(void* d_out, size_t n, const uint64_t x[4], void* stream);

/* evaluate_h's gate arithmetic (halo2_proofs::plonk::evaluation::GraphEvaluator): a circuit's gate / lookup
 * expressions, flattened into a straight-line program, run once per row of the extended domain with every column
 * resident in HBM.  A calculation is five words { op, a, b, c, target }:
 *   op      0 Add  1 Sub  2 Mul  3 Square  4 Double  5 Negate  6 Store  7 MulAdd (a * b + c: one Horner step)
 *   a b c   value sources: kind << 30 | rotation_index << 20 | index, with kind 0 Constant(index), 1 Intermediate(index),
 *           2 Column(index & 0x3fff) at row (idx + rotations[rotation_index]) mod 2^log_size, 3 PreviousValue (d_values[idx] on entry)
 *   target  the intermediate this calculation defines (each is written exactly once, as upstream's are)
 * Upstream's Fixed / Advice / Instance queries are entries of one column table.  Constants [0, n_const) belong to the
 * program (the circuit's); constants [n_const, n_const + n_dynamic) are given with every call -- upstream's Challenge /
 * Beta / Gamma / Theta / Y sources, which change with every proof (n_dynamic <= 16).  Horner(start, parts, factor) is a
 * chain of MulAdd.  rotations are in ROWS (upstream's rot * 2^(extended_k - k)).  The value of the program is its last
 * calculation's (upstream GraphEvaluator::evaluate); it is written to d_values[idx] (2^log_size x 4 u64, external words).
 * A Column source may name a column SHORTER than the domain: index = column | log2(rows) << 14 reads it at the row index
 * modulo its 2^log2(rows) rows (log2(rows) = 0: a full-size column; the table holds at most 256 columns) -- the inverse
 * vanishing-polynomial pattern of divide_by_vanishing_poly (period 2^(extended_k - k)) is such a column, so the division is
 * the program's last multiplication.
 * hm_graph_create validates the program, assigns intermediates to a minimal number of slots by liveness and uploads it;
 * evaluation is asynchronous on `stream` and may run concurrently on different streams. */
int hm_graph_create(const uint32_t* calcs, size_t n_calc, const uint64_t* constants, size_t n_const, size_t n_dynamic,
                    const int32_t* rotations, size_t n_rot, size_t n_columns, uint32_t n_intermediates, uint64_t* out_handle);
int hm_graph_evaluate_dev(uint64_t handle, const void* const* d_columns, size_t n_columns, const uint64_t* dynamic_constants,
                          size_t n_dynamic, uint32_t log_size, void* d_values, void* stream);
/* The quotient h(X) of a proof in ONE call, from COEFFICIENT arrays -- what upstream's evaluate_h, divide_by_vanishing_poly and
 * extended_to_coeff make of the extended arrays (halo2_proofs plonk/evaluation.rs, poly/domain.rs; reached from
 * /root/reference/src/circuits/utils.rs:40-48).
 *   program: an hm_graph_create handle of the UNDIVIDED numerator (custom gates, permutation and lookup terms combined by y)
 *     with rotations as on the n-row domain (unscaled) and every entry of its column table a polynomial of n coefficients;
 *   d_coeff_columns: n_columns device pointers, 2^log_n coefficients each (fixed, then advice, then instance -- the
 *     identity polynomial X where the program reads the point itself);
 *   d_on_cosets: NULL, or n_columns entries of which the non-NULL ones name columns whose values on the `count` cosets exist
 *     already -- count x 2^log_n words as hm_coeff_to_cosets_bn256_fr_dev(..., columns_internal = 1) wrote them for the same
 *     shifts (the fixed columns, permutation polynomials and l_0 / l_last / l_active of a proving key: transformed once, not
 *     once per proof); d_coeff_columns[i] is then not read;
 *   shifts: host, count x 4 u64 -- the cosets shift_c * <omega> to evaluate on, shift_c = zeta * extended_omega^(j_c), distinct
 *     and outside the n-th roots; `count` >= `pieces`, and pieces = (max degree - 1) cosets determine the quotient of a satisfied
 *     circuit (it has fewer than pieces * n coefficients): 5 of the 8 cosets of the extended domain for the reference's circuits;
 *   d_h: pieces x 2^log_n coefficients of h, piece t = coefficients [t n, (t + 1) n).
 * With all E cosets the result equals hm_extended_to_coeff of the divided whole-array evaluation word for word for ANY
 * columns; with fewer it is the polynomial of degree < count * n through the numerator / (X^n - 1) on those cosets -- the same
 * h exactly when the circuit is satisfied.  Asynchronous on `stream`; the workspace (columns x (count + 1) + count arrays of n)
 * belongs to the stream's slot. */
int hm_quotient_by_cosets_bn256_fr_dev(uint64_t program, const void* const* d_coeff_columns, const void* const* d_on_cosets, size_t n_columns,
                                       const uint64_t* dynamic_constants, size_t n_dynamic, uint32_t log_n, const uint64_t omega[4],
                                       const uint64_t* shifts, size_t count, size_t pieces, void* d_h, void* stream);
/* The same in two steps, for a proof split over several GPUs: every device runs hm_quotient_partials on ITS cosets (same
 * arguments; d_partials: count x 2^log_n words, the partial of shifts[c] at c * 2^log_n), the partials travel to one device
 * (2^log_n x 32 B per coset), and hm_quotient_combine turns ALL of them -- d_partials: host array of `count` device pointers in the
 * order of `shifts`, count <= 64 -- into the pieces of h.  hm_quotient_by_cosets is the two on one device. */
int hm_quotient_partials_bn256_fr_dev(uint64_t program, const void* const* d_coeff_columns, const void* const* d_on_cosets, size_t n_columns,
                                      const uint64_t* dynamic_constants, size_t n_dynamic, uint32_t log_n, const uint64_t omega[4],
                                      const uint64_t* shifts, size_t count, void* d_partials, void* stream);
int hm_quotient_combine_bn256_fr_dev(const void* const* d_partials, const uint64_t* shifts, size_t count, uint32_t log_n, size_t pieces, void* d_h,
                                     void* stream);
int hm_graph_destroy(uint64_t handle);
/* The same evaluation with options.  HM_GRAPH_COLUMNS_INTERNAL: every column of the table (short ones included) holds
 * 32 * value mod r instead of value -- the library's internal Montgomery radix is 2^261, so such words need no conversion
 * product when they are loaded (a third of the multiplications of the MerkleSumTree circuit's evaluate_h program are
 * conversions of column loads).  A prover gets its extended-domain columns in that form for nothing: the coset constants
 * of hm_coeff_to_extended_bn256_fr_dev are multiplied into the first NTT pass anyway, so it passes {32, 32 zeta, 32 zeta^2}
 * there (fixed and permutation columns of the proving key: once, at keygen).  PreviousValue on entry, the program's
 * constants and the result written to d_values stay ordinary Fr words.  flags = 0 is hm_graph_evaluate_dev. */
#define HM_GRAPH_COLUMNS_INTERNAL 1
int hm_graph_evaluate_flags_dev(uint64_t handle, const void* const* d_columns, size_t n_columns, const uint64_t* dynamic_constants,
                                size_t n_dynamic, uint32_t log_size, void* d_values, uint32_t flags, void* stream);
/* The same program over `segments` back-to-back blocks of 2^log_segment rows (every column holds segments << log_segment rows;
 * short columns stay periodic in the row index): a rotation wraps INSIDE its block.  One launch for several cosets of the
 * extended domain laid out one after the other per column (hm_coeff_to_cosets_bn256_fr_dev): block c = coset c, rotations
 * unscaled.  segments = 1 is hm_graph_evaluate_flags_dev. */
int hm_graph_evaluate_segments_dev(uint64_t handle, const void* const* d_columns, size_t n_columns, const uint64_t* dynamic_constants,
                                   size_t n_dynamic, uint32_t log_segment, uint32_t segments, void* d_values, uint32_t flags, void* stream);

/* Inputs and known answer of the benchmark of SURVEY.md §8d, without leaving the device:
 *   hm_fr_random_dev           out[i] uniform in [0, r) (Fr::random): one xoshiro256** stream per element, seeded by
 *                              splitmix64 from (seed, i), 254-bit candidates rejected until below r
 *   hm_fr_affine_sequence_dev  out[i] = a + i * b -- the scalars t_i of the bases P_i = [a + i b]G
 *   hm_fr_dot_bn256_dev        out = sum_i a[i] * b[i] (host, 4 u64; synchronises `stream`) -- sum_i s_i t_i, whose
 *                              multiple of G is the MSM's expected result */
int hm_fr_random_dev(void* d_out, size_t n, uint64_t seed, void* stream);
int hm_fr_affine_sequence_dev(void* d_out, size_t n, const uint64_t a[4], const uint64_t b[4], void* stream);
int hm_fr_dot_bn256_dev(const void* d_a, const void* d_b, size_t n, uint64_t out[4], void* stream);

/* a[i] *= c element-wise (device pointer, in place). */
int hm_fr_scale_dev(void* d_a, size_t n, const uint64_t c[4], void* stream);
/* EvaluationDomain::distribute_powers_zeta on its own: a[i] *= c3[i % 3] (device pointer, in place). */
int hm_fr_distribute_powers_dev(void* d_a, size_t n, const uint64_t c3[12], void* stream);

/* ParamsKZG::setup's G1 work: out[i] = [scalars[i]] * base (fixed-base), affine output.
 * scalars: n x 4 u64 device; out: n x 8 u64 device. */
int hm_g1_fixed_base_mul_dev(const void* d_scalars, size_t n, const uint64_t base_xy[8], void* d_out_xy, void* stream);

/* ---- introspection --------------------------------------------------------------------------- */

typedef struct hm_msm_stats {
  double digits_ms, sort_ms, accumulate_ms, reduce_ms, total_ms; /* hipEvent times of the last MSM */
  double accumulate_kernel_ms;                                   /* the bucket-accumulation launch alone */
  uint64_t pairs, tasks;                                         /* non-zero digits, accumulation tasks */
  uint32_t window_bits, windows;
} hm_msm_stats;
int hm_get_msm_stats(hm_msm_stats* out);

/* Per-call counters of the current device since start / hm_reset_stats: what a build of the Rust shim
 *  reads after create_proof to get the MEASURED call trace -- how many best_multiexp /
 * best_fft calls of which size, and where their time went. */
typedef struct hm_stats {
  uint64_t msm_calls, msm_points;         /* best_multiexp-equivalent calls (every form) and the points they covered */
  uint64_t ntt_calls, ntt_elements;       /* best_fft-equivalent transforms (a batched call of b arrays counts b) */
  uint64_t msm_calls_by_log2[32];         /* histogram over floor(log2 n) */
  uint64_t ntt_calls_by_log2[32];         /* histogram over log_n */
  double msm_h2d_us, msm_device_us, msm_host_us;  /* host-pointer uploads; SUM of the hipEvent spans of the launch chains (a grouped
                                                     chain counts once; chains in flight overlap: not a wall time); host fold */
  double ntt_h2d_us, ntt_device_us, ntt_d2h_us;   /* host-pointer form only (device-pointer calls are not waited for) */
  uint64_t h2d_bytes, d2h_bytes;          /* bytes the host-pointer forms moved over PCIe */
  /* the entry points beyond the two functions, by HM_STAT_* kind: calls (queries / lookups where a call carries several)
   * and the field elements they covered */
  uint64_t vector_calls[8], vector_elements[8];
  /* HBM the library holds between calls for the coset transforms' power tables (2^log_n x 32 B per (shift, log_n, form)): a
   * state, not a counter -- hm_reset_stats leaves it.  LRU, at most 48 tables and 2 GiB; given back on an allocation failure. */
  uint64_t coset_table_bytes, coset_tables;
  /* ... and for the twiddle tables of the transforms, one set per (omega, log_n): stage tables, and up to 2^21 a direct inter-pass
   * table of 2^log_n x 36 B (75 MB at 2^21).  LRU, at most 64 sets and 1 GiB. */
  uint64_t ntt_table_bytes, ntt_tables;
  /* The host-pointer forms' copies (csrc/xfer.hip), counted on this device since the library was loaded (hm_reset_stats leaves
   * them): copies handed to hipMemcpy on the caller's pointers (registered ranges, anything below 256 KiB, policy "direct"), copies
   * moved through the library's pinned staging lanes, and the ranges registered now (hm_host_register; process-wide). */
  uint64_t host_copies_direct, host_copies_staged, host_ranges_registered;
} hm_stats;
#define HM_STAT_EVAL_POLYNOMIAL 0
#define HM_STAT_GRAPH_EVALUATE 1
#define HM_STAT_KATE_DIVISION 2
#define HM_STAT_GRAND_PRODUCT 3
#define HM_STAT_BATCH_INVERT 4
#define HM_STAT_LINEAR_COMBINATION 5
#define HM_STAT_LOOKUP_PERMUTE 6
int hm_get_stats(hm_stats* out);
int hm_reset_stats(void);

#ifdef __cplusplus
}
#endif
#endif /* HALO2_MI355X_H */
