/* libzkmi355x -- C-ABI of the MI355X-native Groth16 / Pinocchio prove path for zukelang.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  Every entry
 * point names the reference interface it replaces (paths relative to the zukelang tree).  The
 * OCaml ctypes stubs a maintainer would add are shown in INTEGRATION.md.
 *
 * Byte formats (what the OCaml side already holds):
 *   Fr      32 B little-endian canonical integer < r            (Bls12_381.Fr.to_bytes)
 *   G1      96 B uncompressed ZCash big-endian x || y           (G1.to_bytes, curve.ml:161)
 *   G2     192 B uncompressed x1 || x0 || y1 || y0              (G2.to_bytes)
 *   infinity: first byte 0x40, rest zero.
 *   Compressed outputs (48 B / 96 B) carry the 0x80 / 0x40 / 0x20 flag bits of
 *   to_compressed_bytes (curve.ml:199,208) -- the JSON form of proofs (groth16.ml:110-114).
 *
 * Ownership: all buffers are caller-owned; the library copies host->device and retains no host
 * pointer after return.  Long-lived device state sits behind uint64_t handles with explicit free.
 * Threading: calls are synchronous and blocking (the reference is single-threaded OCaml) and are made from ONE host thread.  One process drives
 * one GPU by default; with a device list of several entries (zk_set_devices) the SAME single-threaded calls drive all of them: a key uploaded
 * through zk_groth16_pk_upload is then sharded over the list behind one handle (see "multi-device keys" below).
 * Errors: 0 = ok, negative = the codes below (the reference raises exceptions; the shim maps
 * ZK_ERR_APPLY_POWERS -> Invalid_argument "apply_powers", ZK_ERR_REMAINDER / ZK_ERR_DOMAIN ->
 * Assert_failure, the rest -> Failure (zk_strerror)).
 */
#ifndef ZKMI355X_H
#define ZKMI355X_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define ZK_OK 0
#define ZK_ERR_ARG (-1)            /* bad length / null pointer / unsupported size */
#define ZK_ERR_NOT_ON_CURVE (-2)   /* a base point fails y^2 = x^3 + b */
#define ZK_ERR_SCALAR_RANGE (-3)   /* an Fr input is >= r */
#define ZK_ERR_REMAINDER (-4)      /* p mod Z != 0: `assert (Polynomial.is_zero rem)`, src/lib/zk/QAP.ml:134 */
#define ZK_ERR_HIP (-5)            /* HIP runtime error (no GPU, OOM, launch failure) */
#define ZK_ERR_APPLY_POWERS (-6)   /* fewer points than coefficients: invalid_arg "apply_powers", src/lib/zk/curve.ml:116 */
#define ZK_ERR_HANDLE (-7)         /* unknown or freed handle */
#define ZK_ERR_DOMAIN (-8)         /* key sets differ: `assert false` in G.dot, src/lib/zk/curve.ml:96-100 */

const char* zk_strerror(int code);
const char* zk_last_error(void);       /* detail of the most recent failure in this process */
int zk_device_count(void);             /* number of visible HIP devices (0 without a GPU) */
int zk_init(int device);               /* bind this process to one GPU (a device list of one entry); idempotent */
int zk_shutdown(void);
/* ---- multi-device keys: N GPUs of one node behind the handle `Groth16.Make(C).prove` holds (SURVEY.md 8b "zk_set_devices(mask)", 8e) -------------
 * zk_set_devices(mask): bit d selects HIP device d; the library's device list becomes the selected devices in ascending order.
 * zk_set_device_list: the general form -- `count` device indices in the order given.  An index may appear MORE THAN ONCE ("virtual devices": several
 * shards of a key on one card, each with its own streams, tables and slots); that is how a one-GPU box exercises the multi-device path, it is never
 * faster than the plain list.  Both may be called before anything else or whenever no key handle is alive (ZK_ERR_ARG otherwise); zk_init(d) afterwards
 * succeeds iff d is the list's first entry.
 * With a list of N > 1 entries:
 *   - zk_groth16_pk_upload / _upload_lagrange shard the key over the list (slice g of both pools on entry g, cut by zk_groth16_shard_range for equal
 *     work) and return ONE handle; zk_groth16_prove / _prove_async / _prove_wait / _set_witness / _reserve_slots / _pk_derive_lagrange / _qap_eval /
 *     _pool_points / _pk_free work on it exactly as on a single-GPU key and produce the SAME bytes (groth16.ml:123-161: the proof does not depend on how
 *     the sums are cut).  Per proof: the Fr stage (QAP.eval, QAP.ml:120-135) runs once, on the slot's owner device (slot mod N); every device copies
 *     its slices of the three scalar vectors out of the owner's memory (hipMemcpyPeerAsync over xGMI), runs its part of the three multi-scalar
 *     products and sends 768 bytes of partial sums to the first device, which adds them and emits the proof.  Nothing blocks before _prove_wait.
 *   - zk_groth16_pk_derive_lagrange derives one of the three independent sets per device (devices 0, 1, 2 of the list) and installs every shard.
 *   - zk_pinocchio_pk_upload likewise cuts every one of the key's eight pools in N slices behind ONE handle (round 5); zk_pinocchio_prove / _prove_async /
 *     _prove_wait / _set_witness / _reserve_slots / _pk_derive_lagrange / _pool_points / _pk_free work on it as on a single-GPU key, same bytes: the Fr stage
 *     and the eight scalar vectors once on the slot's owner device, every device its slices of the eight products (pinocchio.ml:438-505), 1 920 bytes of
 *     partial sums to the first device, which adds them.  The consistency check behind the compact h pool and the derivation of the h bases run on the
 *     first device.
 *   - everything else (zk_msm_*, zk_fr_*, keygen helpers, verify, the explicit one-process-per-GPU shard API below) runs on the list's FIRST
 *     device; the shard API (zk_groth16_pk_upload_sharded, _prove_partial*, _scalars_async ...) refuses multi-device handles with ZK_ERR_ARG.
 * The one-process-per-GPU path (torch.distributed / RCCL, bench.py --gpus N) does not use the device list: every rank keeps its one-entry list. */
int zk_set_devices(uint64_t mask);
/* Configuration by call instead of by environment.  The reference configures by functor application (no flags, no environment), and an
 * OCaml host cannot change the process environment after the library was loaded in a way the C side is sure to see; the tuning knobs INTEGRATION.md 5
 * lists as environment variables are therefore also settable here: name = the variable's name, with or without the "ZK_" prefix, in either case
 * ("msm_window", "ZK_MSM_WINDOW"); value = the string the variable would hold; value == NULL hands the knob back to the environment.  An option set
 * here wins over the environment.  Knobs are read when a key is uploaded or at the first proof and cached: set options BEFORE the first key upload.
 * Unknown names -> ZK_ERR_ARG.  None changes a result (the one that changes a precondition, key_subgroup_check = 0, also switches off the folded
 * windows that rely on it). */
int zk_set_option(const char* name, const char* value);
int zk_set_device_list(const int32_t* devices, uint32_t count);
int zk_get_device_list(int32_t* devices /* may be NULL */, uint32_t capacity, uint32_t* count);

/* ---- Fr stage ---------------------------------------------------------------------------
 * FFT.Make(F).fft / ifft over Bls12_381.Fr -- src/lib/zk/FFT.ml:29-86,222-233:
 * out[k] = sum_j a_j w_N^(jk), w_N = w^(2^32/N), w = 5^((r-1)/2^32); natural order in and out;
 * inverse uses w^-1 and divides by N.  inout: 2^log_n Fr elements. */
int zk_fr_ntt(uint8_t* inout, uint32_t log_n, int inverse);

/* FFT.Make(F).polynomial_mul -- src/lib/zk/FFT.ml:98-105 (== Polynomial.mul, polynomial.ml:124-131):
 * out receives na+nb-1 coefficients (low -> high), *nout the normalized length. */
int zk_fr_poly_mul(const uint8_t* a, size_t na, const uint8_t* b, size_t nb, uint8_t* out, size_t* nout);

/* ---- curve plugin seam (Curve.S.G, src/lib/zk/curve.mli:3-31) ------------------------------
 * G.apply_powers cs xis / G.dot m c -- src/lib/zk/curve.ml:94-118: sum_i scalars[i] * bases[i]
 * over the first nscalars bases.  nscalars > nbases -> ZK_ERR_APPLY_POWERS.  window_bits 0 = auto. */
int zk_msm_g1(const uint8_t* bases, size_t nbases, const uint8_t* scalars, size_t nscalars,
              uint32_t window_bits, uint8_t out[96]);
int zk_msm_g2(const uint8_t* bases, size_t nbases, const uint8_t* scalars, size_t nscalars,
              uint32_t window_bits, uint8_t out[192]);

/* G.of_Fr mapped over a vector (G1.one * s_i) -- src/lib/zk/curve.ml:180; the engine of
 * G.powers (curve.ml:106-109) and of keygen (groth16.ml:70-90, pinocchio.ml:104-156). */
int zk_g1_of_fr(const uint8_t* scalars, size_t n, uint8_t* out /* n*96 */);
int zk_g2_of_fr(const uint8_t* scalars, size_t n, uint8_t* out /* n*192 */);
/* G.powers d s = [ g^(s^i) | i = 0..d ] -- src/lib/zk/curve.ml:106-109 (d+1 points). */
int zk_g1_powers(uint32_t d, const uint8_t s[32], uint8_t* out /* (d+1)*96 */);
int zk_g2_powers(uint32_t d, const uint8_t s[32], uint8_t* out /* (d+1)*192 */);

/* to_compressed_bytes (curve.ml:199,208) of one uncompressed point. */
int zk_g1_compress(const uint8_t in[96], uint8_t out[48]);
/* of_compressed_bytes_exn (curve.ml:199-212): square root, sign from the flag, curve + subgroup checks (host code) */
int zk_g1_decompress(const uint8_t in[48], uint8_t out[96]);
int zk_g2_decompress(const uint8_t in[96], uint8_t out[192]);
int zk_g2_compress(const uint8_t in[192], uint8_t out[96]);
/* The same of_compressed_bytes_exn over a whole list ON THE GPU (round 5): the reference's JSON holds every key point compressed (groth16.ml:24-34,
 * pinocchio.ml:37-60 through curve.ml:199-219), and a 2^20-constraint key is five million square roots and subgroup checks -- half an hour of one
 * host core through the calls above, about a second here.  n points in, n uncompressed points out; ZK_ERR_ARG when some point's compression flag is
 * missing or a coordinate is >= p, ZK_ERR_NOT_ON_CURVE when some abscissa is not on the curve or some point lies outside the prime-order subgroup
 * (the verdicts of the one-point calls; with several bad points in a list, the encoding error is reported first). */
int zk_g1_decompress_batch(const uint8_t* in /* n*48 */, size_t n, uint8_t* out /* n*96 */);
int zk_g2_decompress_batch(const uint8_t* in /* n*96 */, size_t n, uint8_t* out /* n*192 */);

/* ---- protocol seam: Groth16.Make(C).prove (src/groth16/groth16.ml:235-237) -------------------
 * The circuit enters as sparse R1CS rows instead of the dense QAP.t (which is 3*m*n field
 * elements, QAP.ml:11-16): three CSR matrices L (`v`, left operand), R (`w`, right operand),
 * O (`y`, the gate's lhs) with n rows = gates in Gate.Set order (QAP.ml:22) and m columns =
 * variables in Var.compare order; `lhs = l * r` (src/lib/zk/circuit.ml:73-75). */
typedef struct {
    const uint32_t* row_ptr;   /* n + 1 */
    const uint32_t* col;       /* nnz   */
    const uint8_t* val;        /* nnz * 32, Fr */
} zk_csr;

/* One sparse matrix times one vector over Fr: y[g] = sum_e val[e] * x[col[e]] over row g.  With M = one R1CS matrix and x = the solution it gives the
 * VALUES at X = g of `eval' vps` (QAP.ml:121-131: sum_k sol_k poly_k evaluated at the gate's point); with M transposed and x = the Lagrange basis at tau
 * it gives every u_k(tau) of a keygen at once (groth16.ml:59-68, pinocchio.ml:104-109 -- `Poly.apply u_k s` per variable in the reference).
 * ZK_ERR_ARG for a malformed matrix (row_ptr not monotone, column >= cols), ZK_ERR_SCALAR_RANGE for a value >= r. */
int zk_fr_spmv(uint32_t rows, uint32_t cols, const zk_csr* M, const uint8_t* x /* cols * 32 */, uint8_t* y /* rows * 32 */);


/* Proving key of groth16.ml:24-34, fields in declaration order, plus the circuit.
 *   g1: a | d1 | b1 | ti1[n+2] | tiztd[n-1] | ltd_mid[n_mid]   (ltd_mid in Var.Map key order)
 *   g2: b2 | d2 | ti2[n+2]
 *   mid[k] != 0  <=>  variable k is in Dom(ltd_mid) (= circuit.mids, groth16.ml:74-79).
 * Uploads once, precomputes the per-n tables of the Fr stage, returns a handle.
 * KEY points (here, in the sharded / Lagrange-form uploads and in the Pinocchio upload) are checked the way the reference checks them on the way in
 * (of_bytes_exn / of_compressed_bytes_exn, curve.ml:199-212): canonical encoding (ZK_ERR_ARG), curve equation and membership of the prime-order
 * subgroup, [r] P = O on the device (both ZK_ERR_NOT_ON_CURVE; 0.3 s of a 2^20-constraint upload; ZK_KEY_SUBGROUP_CHECK=0 in the environment skips the
 * subgroup part for keys that were checked before).  The per-call bases of zk_msm_g1/g2 are checked for encoding and curve equation only and otherwise
 * TRUSTED to lie in G1 / G2, as every point an OCaml host obtained from Bls12_381.G1/G2 does.
 * The entry points that take points from outside -- zk_g1/g2_decompress, zk_pairing_*, zk_*_verify -- check the subgroup on the host. */
int zk_groth16_pk_upload(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O,
                         const uint8_t* mid /* m */, const uint8_t* pk_g1, size_t pk_g1_points,
                         const uint8_t* pk_g2, size_t pk_g2_points, uint64_t* handle);
/* Lagrange-form proving key (scope row f4 -- an EXTENSION of the key, only a keygen that knows tau can emit it;
 * keys in the reference's format use zk_groth16_pk_upload).  The tau-power lists are replaced by
 *   g1 = a | d1 | b1 | [l_i(tau)]_1 (n) | [lambda_t(tau) Z(tau)/delta]_1 (n-1) | ltd_mid        g2 = b2 | d2 | [l_i(tau)]_2 (n)
 * with l_i the Lagrange basis of the QAP's points 0..n-1 (QAP.ml:84,92) and lambda_t that of n..2n-2.  The same
 * group elements come out (proof bytes identical), but the prover needs only VALUES of v, w, h: three convolutions
 * instead of the O(n log^2 n) basis conversion.  All prove entry points work on the handle (zk_groth16_qap_eval still runs the
 * basis conversion: it is asked for coefficient vectors). */
int zk_groth16_pk_upload_lagrange(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O,
                                  const uint8_t* mid, const uint8_t* pk_g1, size_t pk_g1_points,
                                  const uint8_t* pk_g2, size_t pk_g2_points, uint64_t* handle);
/* Turns an uploaded key in the REFERENCE's format (tau powers, zk_groth16_pk_upload) into the Lagrange form above ON THE DEVICE and without
 * tau: [l_i(tau)] = V^-T [tau^k] is the transposed interpolation map applied to the key's own points (NTTs "in the exponent" over the
 * subproduct tree of the QAP's points, QAP.ml:84,92; csrc/lagrange_derive.hip).  O(n log^2 n) scalar multiplications ONCE per key
 * (seconds at 2^16 constraints, minutes at 2^20); afterwards every prove entry point runs the three-convolution Fr stage.  Proof bytes do
 * not change.  Single-GPU keys only; no proof may be in flight. */
int zk_groth16_pk_derive_lagrange(uint64_t handle);
/* The same in two halves, so that the N ranks of a node share ONE derivation instead of running it N times: the three derived sets -- bit 0:
 * [l_i(tau)]_1, bit 1: [l_i(tau)]_2, bit 2: the h bases [lambda_t(tau) Z(tau)/delta]_1 -- are independent of one another.  Every rank uploads the key
 * whole (zk_groth16_pk_upload), derives the sets of its `sets` mask into caller-owned DEVICE buffers holding the Lagrange-form pools
 * (zk_groth16_lagrange_pool_sizes points of 96 / 192 B; the copied parts a | d1 | b1, ltd_mid, b2 | d2 are always written, a set that is not
 * selected leaves its region untouched), the host framework broadcasts each set from its owner (RCCL broadcast on device memory: set 0 is
 * g1[3, 3+n), set 1 is g2[2, 2+n), set 2 is g1[3+n, 3+n+n-1)), and zk_groth16_pk_install_lagrange builds this rank's slice of the window tables
 * from the complete pools (world = 1: the whole key; same slicing rule as zk_groth16_pk_shard) and flips the key to the three-convolution
 * Fr stage.  zk_groth16_pk_derive_lagrange == _sets(handle, 7, ...) + _install(..., 0, 1).  The key is unchanged until _install has succeeded. */
int zk_groth16_lagrange_pool_sizes(uint64_t handle, uint64_t* g1_points, uint64_t* g2_points);
int zk_groth16_pk_derive_lagrange_sets(uint64_t handle, uint32_t sets, void* d_g1_out, void* d_g2_out);
int zk_groth16_pk_install_lagrange(uint64_t handle, const void* d_g1, const void* d_g2, uint32_t rank, uint32_t world);
/* Turns a key uploaded WHOLE (zk_groth16_pk_upload[_lagrange], possibly after zk_groth16_pk_derive_lagrange) into rank `rank`'s shard of a
 * point-sharded multi-GPU prover: the rank keeps its contiguous slice of both pools (same slicing rule as zk_groth16_pk_upload_sharded) and
 * from then on answers zk_groth16_prove_partial*.  How a derived Lagrange-form key reaches N GPUs: every rank uploads, derives, shards. */
int zk_groth16_pk_shard(uint64_t handle, uint32_t rank, uint32_t world);
/* The key's resident base pool (group 1 or 2) as uncompressed points, in pool order: what was uploaded, or the derived Lagrange-form pool.
 * out == NULL: only *count. */
int zk_groth16_pool_points(uint64_t handle, int group, uint8_t* out, size_t capacity_points, size_t* count);
int zk_groth16_pk_free(uint64_t handle);

/* Groth16.prove rng qap pkey sol with r, s supplied by the caller in the order the reference
 * draws them (groth16.ml:124-125; Fr.gen is an external function).
 * sol: m Fr values in Var.compare order (the witness incl. ONE).
 * proof: a (G1 96 B) | b (G2 192 B) | c (G1 96 B) uncompressed; zk_g*_compress gives the JSON form.
 * Returns ZK_ERR_REMAINDER when the witness does not satisfy the gates (QAP.ml:134). */
int zk_groth16_prove(uint64_t handle, const uint8_t* sol, const uint8_t r[32], const uint8_t s[32],
                     uint8_t proof[384]);
/* Pipelined form: up to 15 proofs in flight on one key, each on its own `slot` (own scratch and
 * HIP streams).  _async enqueues and returns; _wait blocks for that slot and delivers the proof.
 * The single-wave tails of one proof (bucket reduction, affine conversion) then run under the
 * bulk kernels of the next.  zk_groth16_prove == _async + _wait on slot 0. */
int zk_groth16_reserve_slots(uint64_t handle, uint32_t count);   /* allocate slots 0..count-1 now instead of at first use */
int zk_groth16_prove_async(uint64_t handle, const uint8_t* sol, const uint8_t r[32], const uint8_t s[32], uint32_t slot);
int zk_groth16_prove_wait(uint64_t handle, uint32_t slot, uint8_t proof[384]);
/* Keeps a witness resident in HBM; a later zk_groth16_prove / _prove_partial / _qap_eval called with
 * sol = NULL uses it (no host -> device copy inside the call). */
int zk_groth16_set_witness(uint64_t handle, const uint8_t* sol);

/* QAP.eval (src/lib/zk/QAP.ml:120-135) on the uploaded circuit: coefficient vectors of
 * v = sum_k sol_k v_k, w, and h = (v*w - y)/Z, each padded with zeros to n (h: n-1). Any out
 * pointer may be NULL. */
int zk_groth16_qap_eval(uint64_t handle, const uint8_t* sol, uint8_t* v_out, uint8_t* w_out, uint8_t* h_out);

/* ---- point-sharded multi-GPU prove --------------------------------------------
 * One process per GPU.  Rank `rank` of `world` uploads the same key but keeps only its contiguous
 * slice of the two base pools (g1: a | d1 | b1 | ti1 | tiztd | ltd_mid, g2: b2 | d2 | ti2); the Fr
 * stage is replicated.  zk_groth16_prove_partial returns the rank's partial sums of the three
 * multi-scalar products as raw XYZZ Montgomery limbs: A | C (G1, 4*48 B each) | B (G2, 4*96 B).
 * EC addition is not a reduction operator of RCCL, so the host all-gathers the `world` blocks as
 * bytes (torch.distributed / ncclAllGather over xGMI) and zk_groth16_combine adds them on the
 * GPU and emits the proof.  The sum is exact: the proof bytes do not depend on `world`.
 * The slices are cut for equal WORK, not equal length: the first 3 + (n+2 | n) points of the G1 pool
 * (a | d1 | b1 | the tau basis) carry two products of a proof (A and C: groth16.ml:128-134, :147-160),
 * the others one, so rank g holds [cut(g), cut(g+1)) with cut at equal shares of points + heavy prefix;
 * the G2 pool is cut uniformly.  zk_groth16_shard_range is that rule (pure host arithmetic, heavy_prefix = 0
 * for the uniform cut); zk_groth16_pool_layout reports the slices a key actually holds. */
#define ZK_GROTH16_PARTIAL_BYTES 768
int zk_groth16_shard_range(uint64_t points, uint64_t heavy_prefix, uint32_t rank, uint32_t world, uint64_t* lo, uint64_t* hi);
int zk_groth16_pk_upload_sharded(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O,
                                 const uint8_t* mid, const uint8_t* pk_g1, size_t pk_g1_points,
                                 const uint8_t* pk_g2, size_t pk_g2_points, uint32_t rank, uint32_t world,
                                 uint64_t* handle);
int zk_groth16_prove_partial(uint64_t handle, const uint8_t* sol, const uint8_t r[32], const uint8_t s[32],
                             uint8_t partial[ZK_GROTH16_PARTIAL_BYTES]);
/* pipelined form (slots as in zk_groth16_prove_async): the exchange and combine of one proof run on
 * the host while the GPU is already working on the partial sums of the next ones */
int zk_groth16_prove_partial_async(uint64_t handle, const uint8_t* sol, const uint8_t r[32], const uint8_t s[32], uint32_t slot);
int zk_groth16_prove_partial_wait(uint64_t handle, uint32_t slot, uint8_t partial[ZK_GROTH16_PARTIAL_BYTES]);
/* Distributed Fr stage (the form bench.py uses at N > 1).  Proofs are handled in groups of N: rank j runs the
 * Fr stage of the j-th proof of the group ONCE (instead of every rank running it for every proof) and leaves the
 * three scalar vectors over the FULL pools (p1, p1 and p2 canonical Fr elements, see zk_groth16_pool_layout) in
 * caller-owned device buffers; the host framework exchanges the slices (one all-to-all over RCCL / xGMI: rank g
 * receives [lo_g, hi_g) of every proof of the group); zk_groth16_msm_partial_async then runs the three MSMs of
 * one proof over this rank's slice from such a device buffer, and zk_groth16_prove_partial_wait returns its
 * 768-byte partial sums for the all-gather + zk_groth16_combine as above.  Replaces the same reference code
 * (groth16.ml:123-161); QAP.eval (QAP.ml:120-135) is the first half, the apply_powers folds the second.
 * The device pointers are plain HBM addresses (hipMalloc / a framework tensor's data pointer). */
int zk_groth16_pool_layout(uint64_t handle, uint64_t* p1, uint64_t* p2, uint64_t* lo1, uint64_t* hi1, uint64_t* lo2, uint64_t* hi2);
int zk_groth16_scalars_async(uint64_t handle, const uint8_t* sol, const uint8_t r[32], const uint8_t s[32], uint32_t slot,
                             void* d_scal_a /* p1 * 32 B */, void* d_scal_c /* p1 * 32 B */, void* d_scal_b /* p2 * 32 B */);
int zk_groth16_scalars_wait(uint64_t handle, uint32_t slot);    /* ZK_ERR_REMAINDER / ZK_ERR_SCALAR_RANGE as zk_groth16_prove */
int zk_groth16_msm_partial_async(uint64_t handle, uint32_t slot, const void* d_scal_a_slice /* (hi1-lo1) * 32 B */,
                                 const void* d_scal_c_slice, const void* d_scal_b_slice /* (hi2-lo2) * 32 B */);
/* plain device memory for callers without a framework allocator */
int zk_device_malloc(size_t bytes, void** dptr);
int zk_device_free(void* dptr);
int zk_device_memcpy(void* dst, const void* src, size_t bytes);   /* any direction, synchronous */
int zk_groth16_combine(const uint8_t* partials /* world * 768 */, uint32_t world, uint8_t proof[384]);
/* Device-resident form of the exchange (the path RCCL takes: slot buffer -> all-gather on a DEVICE tensor -> combine, no PCIe hop in between):
 * _wait_device waits like zk_groth16_prove_partial_wait and leaves the 768-byte block at the DEVICE address d_partial (same return codes);
 * _combine_device reads `world` blocks from DEVICE memory, block j at d_partials + j * stride_bytes (stride >= 768: the all-gather of a whole
 * round of proofs lands as [rank][proof][768]), adds them on the GPU and returns the proof.  Same group elements, same bytes
 * (groth16.ml:123-161: the proof does not depend on how the sums were cut). */
int zk_groth16_prove_partial_wait_device(uint64_t handle, uint32_t slot, void* d_partial /* device, 768 B */);
int zk_groth16_combine_device(const void* d_partials /* device */, size_t stride_bytes, uint32_t world, uint8_t proof[384]);

/* ---- protocol seam: Pinocchio.Make(C).{NonZK,ZK}.prove (src/pinocchio/pinocchio.ml:536-538,559-561) ----
 * Evaluation key of pinocchio.ml:37-60 flattened per group (maps in Var.Map key order over I_mid, resp.
 * over all m variables for v_all / w_all; lists as stored):
 *   g1: vv | yy | vav | yay | bvwy  (n_mid each) | si (n+1) | v_all (m) | w_all (m) | vt | yt | vavt | yayt | vbt | wbt | ybt
 *   g2: ww | waw (n_mid each) | si2 (n+1, not used by the prover) | wt | wawt
 * zk_pinocchio_prove = ZKCompute.f (:427-514) with dv, dw, dy supplied in the order the reference draws
 * them (:428-430); all three zero gives Compute.f (:210-248), i.e. NonZK.prove.
 * proof: vv (G1) | ww (G2) | yy | h | vavv | waww (G2) | yayy | bvwy = 960 B uncompressed, the field order
 * of Compute.proof (:195-208).  ZK_ERR_REMAINDER as for Groth16.
 * The h product's points v_all | w_all carry the blinding terms sum_k (dw c_k) [v_k(s)] + sum_k (dv c_k) [w_k(s)] (:481-486) = dw [v(s)] + dv [w(s)]
 * with v = sum_k c_k v_k, w = sum_k c_k w_k -- the polynomials QAP.eval builds anyway.  At upload the library checks v_all / w_all against the
 * key's own powers (<v_all, rho> = <si, coefficients of sum_k rho_k v_k> for a pseudo-random rho, likewise w_all: true of every key
 * KeyGen.generate makes, :104-109,140-147) and then lets the two terms ride on si: the COMPACT h pool, n + 1 points instead of n + 1 + 2 m, same
 * proof bytes.  A key that fails the check, or any key under zk_set_option("ZK_PIN_COMPACT_H", "0"), keeps the full pool and is used point by
 * point as ZKCompute.f uses it.  Products that carry the SAME scalar vector -- vv|vt and vav|vavt, yy|yt and yay|yayt, ww|wt and waw|wawt (:438-447,
 * 489-498) -- share one counting sort of it when their base sets have the same identity pattern (true of generated keys; decided per key at upload;
 * "ZK_PIN_SHARED_SORT" = "0" switches it off). */
int zk_pinocchio_pk_upload(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O,
                           const uint8_t* mid, const uint8_t* pk_g1, size_t pk_g1_points,
                           const uint8_t* pk_g2, size_t pk_g2_points, uint64_t* handle);
/* As zk_groth16_pk_derive_lagrange, for the evaluation key of pinocchio.ml:37-60: the powers si are turned into the Lagrange basis of the points
 * n .. 2n-2 in the exponent (plus [Z(s)] = <si, Z>, once), so that h enters its multi-scalar product through VALUES; v(s), w(s) never needed
 * coefficient vectors (they come from the per-variable pools).  Once per key; proofs byte-identical.  With the compact h pool the blinding terms
 * follow: v = kappa_v X^(n-1) + (degree <= n-2), so dw [v(s)] + dv [w(s)] ride on [lambda_t(s)] through the values v(n+t), w(n+t) the prover
 * already extrapolates, plus ONE more base [s^(n-1)] = si[n-1] for the leading coefficients. */
int zk_pinocchio_pk_derive_lagrange(uint64_t handle);
/* A resident base pool of the key as uncompressed points, in pool order (out == NULL: only *count).  Pools 0..5 are the G1 products
 * vv|vt, yy|yt, vav|vavt, yay|yayt, bvwy|vbt|wbt|ybt and the h pool; 6..7 the G2 products ww|wt and waw|wawt (pinocchio.ml:37-60).
 * The h pool by key form:  compact (default): si (n + 1 points); after zk_pinocchio_pk_derive_lagrange [lambda_t(s)] (n-1) | [Z(s)] | [1] | [s^(n-1)];
 *                          full:              si | v_all | w_all;   after the derivation [lambda_t(s)] | [Z(s)] | [1] | v_all | w_all. */
int zk_pinocchio_pool_points(uint64_t handle, int pool, uint8_t* out, size_t capacity_points, size_t* count);
int zk_pinocchio_pk_free(uint64_t handle);
int zk_pinocchio_prove(uint64_t handle, const uint8_t* sol, const uint8_t dv[32], const uint8_t dw[32],
                       const uint8_t dy[32], uint8_t proof[960]);
/* Pipelined form, as for Groth16: up to 15 proofs in flight on one key, each on its own `slot`;
 * sol == NULL uses the witness made resident by zk_pinocchio_set_witness.  zk_pinocchio_prove == _async + _wait on slot 0. */
int zk_pinocchio_reserve_slots(uint64_t handle, uint32_t count);
int zk_pinocchio_set_witness(uint64_t handle, const uint8_t* sol);
int zk_pinocchio_prove_async(uint64_t handle, const uint8_t* sol, const uint8_t dv[32], const uint8_t dw[32],
                             const uint8_t dy[32], uint32_t slot);
int zk_pinocchio_prove_wait(uint64_t handle, uint32_t slot, uint8_t proof[960]);

/* ---- verify surface (scope row f1): Curve.S.Pairing.pairing (src/lib/zk/curve.mli:46-54) ---------------
 * Host code (a handful of pairings per proof, as in the reference, where they are calls into its external
 * library): Groth16 verify = 3 pairings (groth16.ml:163-173), Pinocchio Verify.f = 13 (pinocchio.ml:254-420).
 * zk_pairing_product: gt_out = prod_i e(P_i, Q_i) with one final exponentiation; points uncompressed as
 * everywhere else, checked for curve and subgroup membership.  GT encoding: the 12 Fp coefficients of the tower
 * Fp12 = Fp6[w]/(w^2 - v), Fp6 = Fp2[v]/(v^3 - (1+u)) in the order c0.c0.a, c0.c0.b, c0.c1.a, ..., c1.c2.b,
 * 48 B big-endian each (the reference's GT bytes are defined by its external library: parity unpinned there,
 * compared as field elements here).  zk_pairing_check: *is_one = (product == 1). */
int zk_pairing_product(const uint8_t* g1_points, const uint8_t* g2_points, size_t n, uint8_t gt_out[576]);
int zk_pairing_check(const uint8_t* g1_points, const uint8_t* g2_points, size_t n, int* is_one);
/* Groth16.verify (groth16.ml:163-173): *ok = [ e(A,B) == ab * e(sum_k io_k * ltgm_io_k, gm) * e(C, d) ];
 * ab = e(alpha, beta) in the GT encoding above, io_scalars the public coefficients in the key's variable order. */
int zk_groth16_verify(const uint8_t ab[576], const uint8_t* ltgm_io /* n_io * 96 */, const uint8_t* io_scalars /* n_io * 32 */,
                      size_t n_io, const uint8_t gm[192], const uint8_t d[192], const uint8_t proof[384], int* ok);
/* Pinocchio Verify.f (pinocchio.ml:254-420), verification key (pinocchio.ml:62-75) flattened:
 *   vk_g1 = one | aw | bgm | vv_io[n_io] | yy_io[n_io]      vk_g2 = one2 | av | ay | gm2 | bgm2 | yt | ww_io[n_io] */
int zk_pinocchio_verify(const uint8_t* vk_g1, const uint8_t* vk_g2, const uint8_t* io_scalars, size_t n_io,
                        const uint8_t proof[960], int* ok);

/* ---- measurement hooks (bench.py) ----------------------------------------------------------------
 * With profiling on, kernel families are bracketed by HIP events on the stream they run on;
 * zk_profile_get returns the summed milliseconds and launch count since the last reset. */
int zk_profile_enable(int level);   /* 0 off | 1 the MSM accumulate kernels only (cheap: safe inside a timed region) | 2 every family */
int zk_profile_reset(void);
int zk_profile_get(const char* family, double* total_ms, uint64_t* launches);
int zk_profile_names(char* buf, size_t buflen);   /* comma-separated family names */
/* Work counters gathered at level 2 (one proof at a time) since the last reset, e.g. "msm_accumulate_g1:entries" (sorted (point, window) entries),
 * ":copies" (first entries of a chunk or run: no field product), ":second_steps" (6-product additions of two affine points), ":full_additions". */
int zk_profile_counter(const char* name, uint64_t* value);
int zk_sync(void);                                 /* waits for everything the library has enqueued on every device of its list, the per-slot
                                                      streams of proofs in flight included (hipDeviceSynchronize per device) */
/* Throughput of the register-resident Montgomery multiplier (kind 0 = Fr, 1 = Fp): ALU ceiling.
 * kind | 4: one wave on the whole chip (dependent-chain latency).  kind | 8: the best rate over 2 / 4 / 6 / 8 waves per SIMD. */
int zk_bench_field_mul(int kind, uint32_t iters, double* gmul_per_s);
/* Field-layer self-test hook (tests/test_gpu_field.py): for each of n (even) operand pairs a_i, b_i < p (48-byte little-endian
 * integers) the device evaluates 23 base-field expressions through the lazy-reduction code paths of the group law and
 * returns them fully reduced (23 x 48 B per pair, little-endian); pairs (2k, 2k+1) also act as one Fp2 operand pair. */
int zk_selftest_fp(const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif
