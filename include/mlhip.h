/* mlhip.h -- C ABI of libmlhip.so: the MI355X (gfx950) backend for the data-parallel hot path of
 * IBM/mathlib (batched G1/G2 multi-scalar multiplication, Miller loop, final exponentiation).
 *
 * This is the boundary a Go `driver/hip` package binds through cgo (INTEGRATION.md shows the stub).
 * Every entry point replaces one call the reference's drivers make into gnark-crypto / kilic:
 *
 *   mlhip_msm_g1          driver/gurvy/bls12381/bls12-381.go:766-783 (G1Jac.MultiExp + FromJacobian),
 *                         driver/gurvy/bn254.go:232-245, driver/gurvy/bls12-377.go:229-242,
 *                         driver/kilic/bls12-381.go:247-254
 *   mlhip_msm_g2          additive (no G2 MSM in driver/math.go); semantics = sum of G2.Mul + Add,
 *                         driver/gurvy/bls12381/bls12-381.go:342-358
 *   mlhip_miller_loop     Pairing / Pairing2: bls12-381.go:448-464, bn254.go:247-263, bls12-377.go:244-260
 *   mlhip_final_exp       FExp: bls12-381.go:466-468, bn254.go:265-267, bls12-377.go:262-264
 *   mlhip_pairing_batch   FExp(Pairing(g2[i], g1[i])) element-wise (kilic's Pairing is this composition:
 *                         driver/kilic/bls12-381.go:260-267)
 *
 * Memory layout (in and out) is gnark-crypto's in-memory layout, so Go passes unsafe.Pointer(&slice[0])
 * with no conversion: Fp = k little-endian uint64 limbs in Montgomery form (k = 6 for BLS12-381/377,
 * 4 for BN254; evidence: driver/gurvy/custom.go:24-40, driver/kilic/custom.go:24-29); G1 affine = {X,Y};
 * G2 affine = {X.A0,X.A1,Y.A0,Y.A1}; infinity = all-zero; Gt = E12 {C0{B0{A0,A1},B1,B2},C1{..}}
 * (12 Fp); scalar = 4 uint64 limbs, either fr.Element (Montgomery, scalars_mont = 1) or a plain
 * little-endian integer (scalars_mont = 0; any 256-bit value, reduced mod r on the device the way
 * fr.Element.SetBigInt does for the reference's BaseZr scalars, driver/gurvy/bn254.go:239).
 *
 * Conventions: every function returns 0 on success and a negative MLHIP_E* code otherwise;
 * mlhip_last_error() gives the message for the calling thread.  The library is re-entrant
 * (perf_test.go:382-404 calls Curve methods from many goroutines); it keeps no caller pointer after
 * return.  There is NO CPU fallback: without a usable HIP device every compute entry point fails
 * with MLHIP_ENODEVICE (the Go shim panics, matching the reference's driver convention,
 * driver/gurvy/bn254.go:249-251).
 */
#ifndef MLHIP_H
#define MLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the functions declared here with MLHIP_API are its whole dynamic
 * symbol table (tests/test_abi.py compares `nm -D --defined-only` with this header). */
#define MLHIP_API __attribute__((visibility("default")))

#define MLHIP_CURVE_BN254 0
#define MLHIP_CURVE_BLS12_381 1
#define MLHIP_CURVE_BLS12_377 2

#define MLHIP_OK 0
#define MLHIP_EINVAL (-1)    /* bad argument (unknown curve, window out of range, null pointer) */
#define MLHIP_ENODEVICE (-2) /* no usable HIP device */
#define MLHIP_EHIP (-3)      /* a HIP runtime call failed; see mlhip_last_error() */
#define MLHIP_ENOMEM (-4)

#define MLHIP_GROUP_G1 1
#define MLHIP_GROUP_G2 2

/* ---- library / device ---------------------------------------------------------------------- */
MLHIP_API int mlhip_version(void);
MLHIP_API const char* mlhip_last_error(void);
MLHIP_API int mlhip_device_count(int* count);
/* Pins the calling thread's subsequent calls to one device (-1 undoes it: the thread follows the process's device
 * list again, whose first entry -- device 0 when there is no list -- serves calls that are not spread).  Nothing
 * touches the GPU before the first compute call (the reference computes GenGt at package init, math.go:142-255:
 * importing the backend must not need a GPU). */
MLHIP_API int mlhip_set_device(int device);

/* ---- several devices in one process (one host thread per device, the ABI takes a device list) ------
 * mlhip_init sets the process's device list (n_devices = 0: every visible device; the environment variable
 * MLHIP_DEVICES="0,1,2,3" | "all" does the same without a call).  With two or more devices listed, the host-buffer
 * entry points a Go caller reaches through math.Curve.MultiScalarMul (math.go:960-969,
 * driver/gurvy/bls12381/bls12-381.go:766-783) -- mlhip_msm_g1 / mlhip_msm_g2 / mlhip_bases_create + mlhip_bases_msm
 * from MLHIP_MULTI_MIN pairs (default 2^21), and mlhip_miller_loop / mlhip_final_exp / mlhip_pairing_batch from
 * MLHIP_MULTI_MIN_PAIRINGS elements (default 2^17) -- cut the call into contiguous shards, one per listed device, each
 * run by its own host thread through the whole single-device pipeline (own pooled plan, own PCIe link); the
 * per-device partial sums (already in host memory, where each shard's host tail leaves them) are added on the host.
 * No device-side collective is involved: the caller wants the result in host memory.  (The process-per-GPU form,
 * mathlib_amd/dist.py, exchanges the partials with one RCCL all-gather because there every rank wants the total.)
 * Threads pinned with mlhip_set_device are never spread.  A device may be listed more than once.  Device indices are
 * 0 .. 63 and below the device count; anything else is MLHIP_EINVAL (mlhip_init, mlhip_set_device, the explicit
 * lists below), and a MLHIP_DEVICES value that does not parse makes every compute call and mlhip_get_devices fail
 * with MLHIP_EINVAL instead of silently running on device 0.
 * STATUS: spreading is opt-in (nothing is spread unless the caller lists two or more devices) and has so far run only
 * on lists that repeat device 0 of a one-GPU box (tests/test_multi_device.py); see DESIGN.md section 5. */
MLHIP_API int mlhip_init(const int* devices, int n_devices);
MLHIP_API int mlhip_get_devices(int* devices, int cap); /* returns the length of the list, or MLHIP_EINVAL (malformed MLHIP_DEVICES) */
/* frees every cached plan / arena and forgets the device list; the next call reads MLHIP_DEVICES again */
MLHIP_API int mlhip_shutdown(void);
/* The same MSM with an explicit device list, whatever its size (group: MLHIP_GROUP_G1 / _G2). */
MLHIP_API int mlhip_msm_multi(int curve, int group, const int* devices, int n_devices, const void* points, const void* scalars,
                    int scalars_mont, size_t n, int window_c, void* out_affine);

/* sizes in bytes for a curve: Fp element, G1 affine, G2 affine, Gt, scalar (always 32) */
MLHIP_API int mlhip_sizes(int curve, size_t* fp, size_t* g1, size_t* g2, size_t* gt);

/* ---- host-buffer entry points (what the cgo shim binds) ------------------------------------- */
/* out = sum_i [scalars[i]] points[i].  window_c = 0 picks a window from n; BASELINE config 2 uses 16.
 * n = 0 gives the point at infinity (the reference's MultiExp on empty slices).  Plans and input buffers are pooled
 * between calls (mlhip_release_cache below).  G1 calls of 2^19 pairs and more (G2: 2^18) are streamed over PCIe in
 * segments of 2^18 (2^17) pairs: upload of one segment under the kernels of the one before;
 * MLHIP_STREAM_SEGMENTS=K fixes the count, 0 = one pass.  Smaller calls upload the points beside the sort of the
 * scalars. */
MLHIP_API int mlhip_msm_g1(int curve, const void* points, const void* scalars, int scalars_mont, size_t n, int window_c,
                 void* out_affine);
MLHIP_API int mlhip_msm_g2(int curve, const void* points, const void* scalars, int scalars_mont, size_t n, int window_c,
                 void* out_affine);
/* out_g1 = sum_i [scalars[i]] points_g1[i] and out_g2 = sum_i [scalars[i]] points_g2[i] over ONE scalar vector
 * (additive: BASELINE configs[3]; in the reference's terms MultiScalarMul(a1, b), driver/gurvy/bls12381/bls12-381.go:766-783,
 * and the G2 sum of g2.Mul(b[i]), :342-358, over the same b): the scalars travel and are sorted once, both groups
 * accumulate from the same entry lists.  Same results as mlhip_msm_g1 + mlhip_msm_g2. */
MLHIP_API int mlhip_msm_g1g2(int curve, const void* points_g1, const void* points_g2, const void* scalars, int scalars_mont, size_t n,
                   int window_c, void* out_g1, void* out_g2);

/* out[k] = prod_{j < pairs_per_product} MillerLoop(g1[k*ppp + j], g2[k*ppp + j]); Pairing = 1, Pairing2 = 2
 * (pairs_per_product <= 4).  Pairs holding an infinity contribute 1.  NOT final-exponentiated: like gurvy's
 * Pairing the value is only meaningful after mlhip_final_exp. */
MLHIP_API int mlhip_miller_loop(int curve, const void* g1, const void* g2, size_t pairs_per_product, size_t n_products,
                      void* out_gt);
/* out[i] = in[i]^(k (p^12-1)/r), k = 3 (BLS12 curves) or 2x(6x^2+3x+1) (BN254): gnark's and kilic's value */
MLHIP_API int mlhip_final_exp(int curve, const void* in_gt, size_t n, void* out_gt);
/* out[i] = FExp(Pairing(g2[i], g1[i])) */
MLHIP_API int mlhip_pairing_batch(int curve, const void* g1, const void* g2, size_t n, void* out_gt);
/* out[i] = a[i] * b[i] in Gt (Gt.Mul, driver/gurvy/bls12381/bls12-381.go:417-419), element-wise over n */
MLHIP_API int mlhip_gt_mul(int curve, const void* a_gt, const void* b_gt, size_t n, void* out_gt);

/* out[i] = in[i]^(scalars[i]) in Gt (Gt.Exp, driver/gurvy/bls12381/bls12-381.go:399-407), element-wise over n;
 * valid for any Gt value.  Scalars as for the MSM entry points. */
MLHIP_API int mlhip_gt_exp(int curve, const void* in_gt, const void* scalars, int scalars_mont, size_t n, void* out_gt);
/* out = FExp( prod_i MillerLoop(g1[i], g2[i]) ): a multi-pairing product with ONE shared final exponentiation
 * (what a verifier computes before IsUnity: perf_test.go:254-259).  n Miller loops run one per lane, the
 * product is a log-depth tree of Gt multiplications on the device. */
MLHIP_API int mlhip_pairing_product(int curve, const void* g1, const void* g2, size_t n, void* out_gt);

/* ---- device-resident entry points (points / scalars already in HBM; resident SRS) ----------- */
typedef struct mlhip_msm_plan mlhip_msm_plan;
/* Workspace for MSMs of up to max_n points on the calling thread's device. */
MLHIP_API int mlhip_msm_plan_create(int curve, int group, size_t max_n, int window_c, mlhip_msm_plan** plan);
MLHIP_API int mlhip_msm_plan_destroy(mlhip_msm_plan* plan);
/* d_points / d_scalars are device pointers; stream is a hipStream_t (NULL = default stream).
 * out_affine is HOST memory; the call returns after the result is there.  When out_xyzz is non-NULL
 * the un-normalised partial sum (X,Y,ZZ,ZZZ) is written there too (multi-GPU combine). */
MLHIP_API int mlhip_msm_run(mlhip_msm_plan* plan, const void* d_points, const void* d_scalars, int scalars_mont, size_t n,
                  void* stream, void* out_affine, void* out_xyzz);
/* The same in two halves, so consecutive MSMs pipeline: mlhip_msm_launch enqueues the kernels and the
 * D2H of the window sums on `stream` and returns; mlhip_msm_finish waits for them and runs the O(1) host
 * tail.  With two plans on two streams the sort of MSM k+1 overlaps the bucket accumulation of MSM k and
 * the host tail of MSM k overlaps GPU work (a Groth16 prover issues 4-5 MSMs back to back). */
MLHIP_API int mlhip_msm_launch(mlhip_msm_plan* plan, const void* d_points, const void* d_scalars, int scalars_mont, size_t n,
                     void* stream);
MLHIP_API int mlhip_msm_finish(mlhip_msm_plan* plan, void* out_affine, void* out_xyzz);
/* The G1 MSM and the G2 MSM of ONE scalar vector (additive: BASELINE configs[3] "G1 + G2 MSM, shared scalars"; in the
 * reference's terms MultiScalarMul(a1, b) -- driver/gurvy/bls12381/bls12-381.go:766-783 -- and the G2 sum of
 * g2.Mul(b[i]) -- :342-358 -- over the same b).  Both plans: same curve, device and window width; the (window, bucket)
 * entry lists depend on the scalars only, so they are sorted once and both groups accumulate from them.  Finish each
 * plan with mlhip_msm_finish.  Plans that cannot share (different widths, a second-implementation path) run one after
 * the other with the same results. */
MLHIP_API int mlhip_msm_launch_shared(mlhip_msm_plan* g1_plan, mlhip_msm_plan* g2_plan, const void* d_points_g1,
                            const void* d_points_g2, const void* d_scalars, int scalars_mont, size_t n, void* stream);
/* Phase timings of the last run with profiling on (HIP events on the plan's stream), milliseconds:
 * [0] digits [1] sort (histogram scan + scatter) [2] bucket accumulation [3] bucket reduction
 * [4] device total [5] host tail [6] the number of tiles the accumulation ran in (device-resident inputs from 2^22 / 2^23
 * points on are accumulated tile by tile; [1] and [2] are then sums over the tiles' launches).  A streamed host-buffer
 * MSM (mlhip_msm_g1 and friends on a pooled plan) records no phase events: [0..5] are then 0.  [7] and [8] are not times:
 * the window width c the plan runs with (the library's pick when it was created with window_c = 0) and its number of
 * windows W; [9] is 1 when the last launch summed its buckets in twisted Edwards coordinates (a subgroup-trusted
 * BLS12-377 G1 plan or table with the SRS promise, mlhip_msm_plan_assume_srs), else 0; [10] is 1 when the plan reads the
 * shifted-base tables of a mlhip_bases handle ([8] is then the number of digits a scalar is cut into).  Returns the number of
 * values written (at most `cap`, at most 11). */
MLHIP_API int mlhip_msm_plan_set_profiling(mlhip_msm_plan* plan, int on);
/* The caller declares (on != 0) that this plan's points are a fixed SRS: (1) the point buffer at a given device address
 * holds the same points at every launch until the promise is taken back (the plan keeps its converted copy of them and
 * converts nothing on later launches), and (2) every point lies in the prime-order subgroup or is the point at infinity --
 * what gnark's SetBytes checks on the way in, and what an SRS is by construction.  The reference's MultiScalarMul
 * (driver/gurvy/bls12-377.go:229-242, gnark MultiExp) accepts any curve points in fresh slices, and so does a plan without
 * this promise.  With it, a BLS12-377 G1 plan sums its buckets in twisted Edwards coordinates (7 field products per
 * addition instead of 10; that addition law is complete on the subgroup only): same result bytes, less time.  On the other
 * curves and for G2 only (1) matters.  mlhip_bases_create makes the same promise for its own table after checking (2) on
 * the device.  A promise that does not hold gives an undefined RESULT, never a fault. */
MLHIP_API int mlhip_msm_plan_assume_srs(mlhip_msm_plan* plan, int on);
MLHIP_API int mlhip_msm_plan_timings(mlhip_msm_plan* plan, float* ms, int cap);

MLHIP_API int mlhip_miller_loop_device(int curve, const void* d_g1, const void* d_g2, size_t pairs_per_product,
                             size_t n_products, void* d_out_gt, void* stream);
MLHIP_API int mlhip_final_exp_device(int curve, const void* d_in_gt, size_t n, void* d_out_gt, void* stream);
MLHIP_API int mlhip_pairing_batch_device(int curve, const void* d_g1, const void* d_g2, size_t n, void* d_out_gt,
                               void* stream);
MLHIP_API int mlhip_gt_mul_device(int curve, const void* d_a_gt, const void* d_b_gt, size_t n, void* d_out_gt, void* stream);
MLHIP_API int mlhip_gt_exp_device(int curve, const void* d_in_gt, const void* d_scalars, int scalars_mont, size_t n,
                        void* d_out_gt, void* stream);

/* out[i] = [scalars[i]] points[i * point_stride]: batched single-scalar multiplication (G1.Mul / G2.Mul,
 * driver/gurvy/bls12381/bls12-381.go:238-247, :342-351).  point_stride = 0 multiplies one base point by
 * every scalar (how the synthetic benchmark inputs [k_i]G are produced); from 2^12 scalars on that case builds a table
 * of [m 2^(wj)]P on the device (w = 12; MLHIP_FB_WINDOW overrides) and spends ceil(256 / w) mixed
 * additions per scalar instead of 256 doublings + 64 additions (MLHIP_FIXED_BASE_MIN=n moves the threshold, 0 = never).
 * The table stays on the device: the next call with the same curve, group and base point skips the build (the comparison
 * runs on the device, the call stays asynchronous; MLHIP_FB_CACHE=0 builds it every time).  Device pointers. */
MLHIP_API int mlhip_scalar_mul_device(int curve, int group, const void* d_points, size_t point_stride, const void* d_scalars,
                            int scalars_mont, size_t n, void* d_out_affine, void* stream);
/* host-buffer form */
MLHIP_API int mlhip_scalar_mul(int curve, int group, const void* points, size_t point_stride, const void* scalars,
                     int scalars_mont, size_t n, void* out_affine);

/* ---- resident bases (upload-once point table, only the scalars travel per call) ----------
 * For callers that cannot hold device pointers themselves (the Go shim): the points are uploaded once, every
 * mlhip_bases_msm() uploads n x 32 bytes of scalars and returns sum_i [s_i] P_i over the first n bases (n <= the
 * count given at creation).  Calls on one handle from several threads are serialized inside the library (one MSM at a
 * time per handle); handles are independent of each other. */
typedef struct mlhip_bases mlhip_bases;
/* window_c = 0 leaves the geometry to the library, and lets it keep SHIFTED-BASE TABLES for the handle (tables of at least
 * 2^10 bases that fit a quarter of the free device memory; MLHIP_BASES_TABLES=0 never, =1 always): besides P_i the device
 * holds 2^off(j) P_i for every digit position j (first bit off(j)) of a signed-digit scalar (13 rows a base from 2^16 bases
 * on -- 20-bit digits, 1.5 GB for 2^20 BLS12-381 G1 bases, built once in ~60 ms --, 16 to 20 rows below), so that all digits
 * of all scalars add into ONE set of 2^(c-1) buckets: a wider digit at the same reduction cost (19 % fewer bucket additions)
 * and a host tail of at most 19 doublings instead of 256.  Same result bytes.
 * An explicit window_c asks for that Pippenger geometry over the plain table (what BASELINE's "c = 16" names).  The
 * reference has no counterpart: its MultiScalarMul takes fresh slices (driver/gurvy/bls12381/bls12-381.go:766-783). */
MLHIP_API int mlhip_bases_create(int curve, int group, const void* points, size_t n, int window_c, mlhip_bases** bases);
/* the same handle from n affine points that are already in the current device's memory (a copy is taken; the caller's
 * buffer may be reused at once): always one device, never spread */
MLHIP_API int mlhip_bases_create_device(int curve, int group, const void* d_points, size_t n, int window_c, mlhip_bases** bases);
/* the table cut into contiguous shards over an explicit device list (mlhip_bases_create does this by itself with the
 * process's list from MLHIP_MULTI_MIN bases on): every mlhip_bases_msm then moves each device's scalars over its own
 * PCIe link and adds the per-device partial sums on the host */
MLHIP_API int mlhip_bases_create_multi(int curve, int group, const int* devices, int n_devices, const void* points, size_t n,
                             int window_c, mlhip_bases** bases);
MLHIP_API int mlhip_bases_msm(mlhip_bases* bases, const void* scalars, int scalars_mont, size_t n, void* out_affine);
/* the same MSM with the n scalars already in device memory (a prover whose witness was computed on the device); `stream`
 * as in mlhip_msm_run; single-device handles only (MLHIP_EINVAL for a table spread over several devices) */
MLHIP_API int mlhip_bases_msm_device(mlhip_bases* bases, const void* d_scalars, int scalars_mont, size_t n, void* stream,
                           void* out_affine);
/* the plan a single-device handle runs its MSMs on (NULL otherwise) -- for mlhip_msm_plan_set_profiling /
 * mlhip_msm_plan_timings only; it stays owned by the handle */
MLHIP_API mlhip_msm_plan* mlhip_bases_plan(mlhip_bases* bases);
/* 1 when mlhip_bases_create verified, on the device, that every point of the table is on the curve and in the prime-order
 * subgroup (or the point at infinity) -- done for BLS12-377 G1 tables, whose MSMs then sum their buckets in twisted
 * Edwards coordinates (see mlhip_msm_plan_assume_srs); 0 otherwise (other curves, G2, a table with a point outside the
 * subgroup, MLHIP_EDWARDS=0): such tables take the Weierstrass kernels and give the reference's result for any input. */
MLHIP_API int mlhip_bases_checked_subgroup(mlhip_bases* bases);
MLHIP_API int mlhip_bases_destroy(mlhip_bases* bases);

/* The host-buffer MSM entry points above keep up to 16 plans + input buffers (at most 32 GB) alive between calls (creating and
 * destroying them costs as much as a 2^20-point MSM); this frees the idle ones.  MLHIP_NO_PLAN_CACHE=1 disables the pool. */
MLHIP_API int mlhip_release_cache(void);

/* ---- wire format (bulk NewG1FromBytes / NewG1FromCompressed and G1.Bytes / G1.Compressed,
 * driver/gurvy/bls12381/bls12-381.go:531-569, :286-296) --------------------------------------------------
 * Decode n points of gnark's wire format (BLS12 curves: zcash 3-bit header; BN254: 2-bit header), all of the
 * same form: compressed (fp bytes each) or uncompressed (2 x fp bytes).  Every point gets a status byte:
 * 0 ok | 1 malformed (flags, coordinate >= p, bad infinity) | 2 not on the curve | 3 not in the r-torsion
 * subgroup; out_affine[i] is (0,0) unless status[i] == 0.  subgroup_check: 0 skips the test (gnark's SetBytes always
 * does it), 1 runs the fastest exact test (BLS12 curves: phi(P) = [-x^2]P for G1, psi(Q) = [x]Q for G2; BN254 G2:
 * [x+1]Q + psi([x]Q) + psi^2([x]Q) = psi^3([2x]Q) -- 64-bit ladders, the criteria gnark uses), 2 forces the plain [r]P ladder (kept for cross-checking). */
MLHIP_API int mlhip_g1_from_bytes(int curve, const void* wire, size_t n, int compressed, int subgroup_check, void* out_affine,
                        unsigned char* status);
MLHIP_API int mlhip_g1_to_bytes(int curve, const void* affine, size_t n, int compressed, void* wire);
MLHIP_API int mlhip_g1_from_bytes_device(int curve, const void* d_wire, size_t n, int compressed, int subgroup_check,
                               void* d_out_affine, unsigned char* d_status, void* stream);
MLHIP_API int mlhip_g1_to_bytes_device(int curve, const void* d_affine, size_t n, int compressed, void* d_wire, void* stream);
/* G2 (NewG2FromBytes / NewG2FromCompressed, bls12-381.go:541-569): 2 / 4 fp-sized big-endian values per point in
 * the order X.A1, X.A0 [, Y.A1, Y.A0]; y recovered by a square root in Fp2; subgroup_check as for G1. */
MLHIP_API int mlhip_g2_from_bytes(int curve, const void* wire, size_t n, int compressed, int subgroup_check, void* out_affine,
                        unsigned char* status);
MLHIP_API int mlhip_g2_to_bytes(int curve, const void* affine, size_t n, int compressed, void* wire);
MLHIP_API int mlhip_g2_from_bytes_device(int curve, const void* d_wire, size_t n, int compressed, int subgroup_check,
                               void* d_out_affine, unsigned char* d_status, void* stream);
MLHIP_API int mlhip_g2_to_bytes_device(int curve, const void* d_affine, size_t n, int compressed, void* d_wire, void* stream);

/* ---- group helpers (host, O(n) tiny): combine per-GPU partial results after the RCCL all-gather */
MLHIP_API int mlhip_g1_sum(int curve, const void* affine_points, size_t n, void* out_affine);
MLHIP_API int mlhip_g2_sum(int curve, const void* affine_points, size_t n, void* out_affine);

/* ---- field kernel (parity / roofline probe): out[i] = a[i] * b[i] (Montgomery), device pointers */
MLHIP_API int mlhip_fp_mul_device(int curve, const void* d_a, const void* d_b, size_t n, int repeat, void* d_out,
                        void* stream);

/* ---- environment switches (all of them; none is needed for normal use) ---------------------------------------------------
 * Read by the library at the point named.  "2nd impl" = selects a slower second implementation of the same result that the
 * parity tests run against the default one; it never changes a result byte.
 *
 *   devices and threads (read once per process)
 *     MLHIP_DEVICES="0,1,.." | "all"   the process's device list (see mlhip_init)
 *     MLHIP_MULTI_MIN, MLHIP_MULTI_MIN_PAIRINGS   smallest host-buffer MSM / pairing batch that is spread over the list (2^21, 2^17)
 *     MLHIP_HOST_THREADS=N             threads of the host-tail pool per call incl. the caller (default: min(8, cores of the
 *                                      affinity mask / LOCAL_WORLD_SIZE, halved when ranks share the host); 1 = none)
 *     MLHIP_NO_PLAN_CACHE=1            host-buffer MSMs create and destroy their plan per call
 *   MSM geometry and schedules (read per plan or per launch)
 *     MLHIP_STREAM_SEGMENTS=K          host-buffer MSMs in K equal segments (0 / 1 = one upload, one pass)
 *     MLHIP_STREAM_SCHEDULE="w0,w1,.." ... in segments of these relative lengths (default for resident bases: "3,13")
 *     MLHIP_TILE_LOG2=t                device-resident MSMs in tiles of 2^t pairs (0 = never; default 2^21 G1 / 2^20 G2 from 2^22 / 2^23)
 *     MLHIP_SORT_AHEAD=0, MLHIP_SORT_AHEAD_PRIO=0   tiles: sort of tile s+1 in line / on a default-priority stream
 *     MLHIP_SORT_TILE, MLHIP_CHUNK_LOG2, MLHIP_ACC_BLOCK, MLHIP_RED_BLOCK   kernel tile / chunk / workgroup sizes (sweeps)
 *     MLHIP_FIXED_BASE_MIN=n, MLHIP_FB_WINDOW=w, MLHIP_FB_CACHE=0   batched Mul of one base: table from n scalars on, width, no reuse
 *     MLHIP_EDWARDS=0                  BLS12-377 G1 over a checked SRS: keep the Weierstrass bucket sums
 *     MLHIP_BASES_TABLES=0|1           mlhip_bases_create: never / always keep shifted-base tables (default: see there)
 *     MLHIP_FOLD_WINDOW=c, MLHIP_FOLD_TILE_LOG2=t   ... their digit width (default 13 .. 20 by size) and tile (2^20 bases)
 *     MLHIP_PAIRING_QUAD=0|1           BLS12-381: never / always one pairing per quad of lanes (default: up to 2^14 elements)
 *   2nd impl (parity tests; DESIGN.md section 2 lists which test runs which)
 *     MLHIP_ACC32=1                    boundary-form (32-bit limb) bucket accumulation, G1 and G2
 *     MLHIP_REDUCE32=1, MLHIP_REDUCE_ONE_LANE=1   boundary-form / one-point-per-lane bucket reduction
 *     MLHIP_LEGACY_SORT=1              global-atomic sort (also the default above 2^24 pairs at c = 16)
 *     MLHIP_SCATTER_STAGED=0           round-1 coarse scatter (one store per entry)
 *     MLHIP_NO_QUAD_ACC=1              small MSMs: one bucket per lane instead of per quad
 *     MLHIP_G2_KC=1                    G2 buckets split by coordinate (one-lane Karatsuba Fp2 products; measured 3-5 % slower)
 *     MLHIP_PAIRING_ONE_LANE=1, MLHIP_PAIRING_SAT=1, MLHIP_SCALAR_MUL_ONE_LANE=1   one-lane / saturated-limb pairing and G2.Mul kernels
 *   tests only
 *     MLHIP_FAULT_INJECT=sort_helper_alloc   the sort-ahead helper allocation "fails" (tests/test_gpu_parity.py)
 */

#ifdef __cplusplus
}
#endif
#endif /* MLHIP_H */
