/*
Package hip is the MI355X backend for the hot path of github.com/IBM/mathlib: it plugs in beside
driver/amcl, driver/gurvy and driver/kilic behind the unchanged driver.Curve interface
(driver/math.go:49-180) and overrides ONLY the data-parallel methods -- MultiScalarMul, Pairing,
Pairing2, FExp -- with calls into libmlhip.so (C ABI: include/mlhip.h).  Everything else (the ~40
remaining Curve methods, element arithmetic, serialization, hashing) is inherited bit-for-bit from the
gurvy BLS12-381 driver by embedding, the way BBSCurve overrides four methods of Curve in
driver/gurvy/bls12381/bls12-381.go:785-867.

Memory is handed over with zero conversion: gnark-crypto's G1Affine / G2Affine / fr.Element / GT are
plain arrays of little-endian uint64 limbs in Montgomery form, which is exactly the layout the kernels
use (driver/gurvy/custom.go:24-40 proves the layout for fp.Element; init() below re-checks the sizes).

NOTE: this file has never been compiled -- the build image has no Go toolchain (SURVEY.md, headline
fact 3).  It is the binding a maintainer would add; INTEGRATION.md walks through it.  The same ABI is
exercised by the Python mirror mathlib_amd/driver.py, which the test-suite runs on the GPU.

Failure convention: like every reference driver this package panics on failure (there are no error
returns in package driver; cf. driver/gurvy/bn254.go:249-251).  There is no CPU fallback: without a
usable GPU the hot methods panic with the library's message; the inherited methods keep working, so
registering the curve in math.Curves never needs a GPU (GenGt at package init uses the embedded
driver: driver/gurvy/bls12381/bls12-381.go:490-497).
*/
package hip

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../mathlib_amd -lmlhip -Wl,-rpath,${SRCDIR}/../../../mathlib_amd
#include <stdlib.h>
#include "mlhip.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"unsafe"

	"github.com/IBM/mathlib/driver"
	gurvy381 "github.com/IBM/mathlib/driver/gurvy/bls12381"
	bls12381 "github.com/consensys/gnark-crypto/ecc/bls12-381"
	"github.com/consensys/gnark-crypto/ecc/bls12-381/fr"
)

// WindowC is the Pippenger window (0 = chosen from n; BASELINE config 2 uses 16).
var WindowC = 0

// The GPU is a throughput device.  Measured on one MI355X through this ABI (tools/perf_latency.py,
// profiles/r03_perf_latency.txt): a MultiScalarMul of 2 points takes 0.3-0.4 ms, of 2^20 points 4.2 ms; but ONE Miller
// loop takes 2.5 ms and ONE final exponentiation 5.2 ms, because a single pairing occupies a single quad of lanes
// (65 536 pairings take 17.7 ms).  gnark on the CPU does a single pairing in about a millisecond.  Hence:
//
// MinDeviceMSM: MultiScalarMul with fewer pairs stays on the embedded gurvy driver.  A host-slice MSM costs the device
// a flat 0.39-0.51 ms up to 2^10 pairs and 0.55-0.9 ms from 2^12 to 2^16 (profiles/r03_min_device_msm.txt,
// profiles/r03_perf_small_msm.txt).  The CPU side cannot be measured with gnark (no Go toolchain here or on the GPU
// box: profiles/r03_go_probe.txt); the stated stand-in, the C restatement (oracle/cref, tools/perf_min_device_msm.py)
// timed on the GPU box's own cores, needs on ONE thread 0.40 ms for 2 pairs, 2.6 ms for 32 and 29 ms for 2^10, on 8
// threads 0.52 / 0.78 / 4.4 ms, and never gets under 2 ms on all 64 -- the device is level with it from the second
// pair.  gnark with ADX assembly is several times faster than that plain-C port, which puts the break-even near 8-32
// pairs; 32 keeps the tiny proofs-of-knowledge MSMs (perf_test.go:198-224, 3-7 pairs) on the CPU.
var MinDeviceMSM = 32

// MinDevicePairingBatch: PairingBatch with fewer pairs stays on the embedded gurvy driver.  Any batch up to 16 384
// pairs costs one wave time on the device (5.2 ms on an MI355X, one pairing per quad of lanes:
// profiles/r02_perf_pairing_quads.txt); a CPU
// core needs about a millisecond per pairing, so a few hundred pairs are where the device starts to win.
var MinDevicePairingBatch = 256

// DevicePairing: when false (default) the single-shot Pairing / Pairing2 / FExp stay on the embedded gurvy driver
// and only the batched entry points (PairingBatch) use the GPU.  Mixing is safe: FExp of either Miller loop's
// output is the same canonical Gt (SURVEY.md section 8c).
var DevicePairing = false

func init() {
	// the C ABI assumes gnark's in-memory layout; refuse to run if it ever changes
	if unsafe.Sizeof(bls12381.G1Affine{}) != 96 || unsafe.Sizeof(bls12381.G2Affine{}) != 192 ||
		unsafe.Sizeof(bls12381.GT{}) != 576 || unsafe.Sizeof(fr.Element{}) != 32 {
		panic("hip: gnark-crypto element layout changed; libmlhip.so expects 96/192/576/32-byte elements")
	}
}

// check runs one library call and panics with the library's message when it fails.  libmlhip keeps the message (and
// a thread's optional device pin) per OS thread, and goroutines migrate between threads from one cgo call to the
// next: the goroutine is locked to its thread from the call until the message has been read.  Device selection does
// not rely on thread state at all -- it is the process-wide list of SetDevices below.
func check(call func() C.int) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := call(); rc != 0 {
		panic(fmt.Sprintf("hip: libmlhip error %d: %s", int(rc), C.GoString(C.mlhip_last_error())))
	}
}

// SetDevices sets the GPUs of this process (mlhip_init; no argument = every visible device; the environment variable
// MLHIP_DEVICES does the same).  With two or more, MultiScalarMul / MultiScalarMulG2 / NewBases of 2^21 pairs and more
// and PairingBatch of 2^17 pairs and more are cut into contiguous shards, one per GPU, each run by its own host thread
// inside the library; the per-GPU partial sums are added on the host (SURVEY.md 8e; BASELINE configs 4 and 5).
// Smaller calls run on the first listed device.
func SetDevices(devices ...int) {
	var p *C.int
	d := make([]C.int, len(devices))
	for i, v := range devices {
		d[i] = C.int(v)
	}
	if len(d) > 0 {
		p = &d[0]
	}
	check(func() C.int { return C.mlhip_init(p, C.int(len(d))) })
}

// Curve is the gurvy BLS12-381 curve with its hot methods moved to the GPU.
type Curve struct {
	gurvy381.Curve
}

func NewCurve() *Curve { return &Curve{*gurvy381.NewCurve()} }

// MultiScalarMul replaces driver/gurvy/bls12381/bls12-381.go:766-783 (gather + G1Jac.MultiExp +
// FromJacobian).  The gather into contiguous arrays stays (the interface hands over boxed elements);
// scalars are passed as fr.Element in Montgomery form, exactly what the gurvy driver feeds MultiExp.
func (c *Curve) MultiScalarMul(a []driver.G1, b []driver.Zr) driver.G1 {
	n := len(a)
	points := make([]bls12381.G1Affine, n)
	scalars := make([]fr.Element, len(b))
	for i := range a {
		points[i] = a[i].(*gurvy381.G1).G1Affine
		scalars[i] = gurvy381.ZrValue(b[i]) // accessor for Zr.val (bls12-381.go:49-52); see INTEGRATION.md
	}
	out := &gurvy381.G1{}
	if len(b) != n || n == 0 {
		// gnark's MultiExp errors on a length mismatch and the driver drops the error: identity
		return out
	}
	if n < MinDeviceMSM {
		return c.Curve.MultiScalarMul(a, b)
	}
	check(func() C.int {
		return C.mlhip_msm_g1(C.MLHIP_CURVE_BLS12_381,
		unsafe.Pointer(&points[0]), unsafe.Pointer(&scalars[0]), 1, C.size_t(n), C.int(WindowC),
		unsafe.Pointer(&out.G1Affine))
	})
	return out
}

// Pairing replaces bls12-381.go:448-455: Miller loop only, compare after FExp.
func (c *Curve) Pairing(p2 driver.G2, p1 driver.G1) driver.Gt {
	if !DevicePairing {
		return c.Curve.Pairing(p2, p1)
	}
	out := &gurvy381.Gt{}
	check(func() C.int {
		return C.mlhip_miller_loop(C.MLHIP_CURVE_BLS12_381,
		unsafe.Pointer(&p1.(*gurvy381.G1).G1Affine), unsafe.Pointer(&p2.(*gurvy381.G2).G2Affine),
		1, 1, unsafe.Pointer(&out.GT))
	})
	return out
}

// Pairing2 replaces bls12-381.go:457-464 (one shared Miller loop over two pairs).
func (c *Curve) Pairing2(p2a, p2b driver.G2, p1a, p1b driver.G1) driver.Gt {
	if !DevicePairing {
		return c.Curve.Pairing2(p2a, p2b, p1a, p1b)
	}
	g1 := [2]bls12381.G1Affine{p1a.(*gurvy381.G1).G1Affine, p1b.(*gurvy381.G1).G1Affine}
	g2 := [2]bls12381.G2Affine{p2a.(*gurvy381.G2).G2Affine, p2b.(*gurvy381.G2).G2Affine}
	out := &gurvy381.Gt{}
	check(func() C.int {
		return C.mlhip_miller_loop(C.MLHIP_CURVE_BLS12_381,
		unsafe.Pointer(&g1[0]), unsafe.Pointer(&g2[0]), 2, 1, unsafe.Pointer(&out.GT))
	})
	return out
}

// FExp replaces bls12-381.go:466-468.
func (c *Curve) FExp(a driver.Gt) driver.Gt {
	if !DevicePairing {
		return c.Curve.FExp(a)
	}
	out := &gurvy381.Gt{}
	check(func() C.int {
		return C.mlhip_final_exp(C.MLHIP_CURVE_BLS12_381,
		unsafe.Pointer(&a.(*gurvy381.Gt).GT), 1, unsafe.Pointer(&out.GT))
	})
	return out
}

// ---- additive API (not in driver.Curve; needed by BASELINE configs 3 and 4) --------------------

// MultiScalarMulG2 = sum of G2.Mul + Add (bls12-381.go:342-358) as one MSM.
func (c *Curve) MultiScalarMulG2(a []driver.G2, b []driver.Zr) driver.G2 {
	n := len(a)
	points := make([]bls12381.G2Affine, n)
	scalars := make([]fr.Element, len(b))
	for i := range a {
		points[i] = a[i].(*gurvy381.G2).G2Affine
		scalars[i] = gurvy381.ZrValue(b[i])
	}
	out := &gurvy381.G2{}
	if len(b) != n || n == 0 {
		return out
	}
	check(func() C.int {
		return C.mlhip_msm_g2(C.MLHIP_CURVE_BLS12_381,
		unsafe.Pointer(&points[0]), unsafe.Pointer(&scalars[0]), 1, C.size_t(n), C.int(WindowC),
		unsafe.Pointer(&out.G2Affine))
	})
	return out
}

// MultiScalarMulG1G2 = (MultiScalarMul(a1, b), MultiScalarMulG2(a2, b)) for ONE scalar vector (BASELINE configs[3]: a
// prover's G1 and G2 MSM over the same witness): the scalars travel and are sorted once, both groups accumulate from
// the same bucket lists.  Mismatched lengths give the identities, as MultiScalarMul does (bls12-381.go:777).
func (c *Curve) MultiScalarMulG1G2(a1 []driver.G1, a2 []driver.G2, b []driver.Zr) (driver.G1, driver.G2) {
	n := len(a1)
	o1, o2 := &gurvy381.G1{}, &gurvy381.G2{}
	if len(a2) != n || len(b) != n || n == 0 {
		return o1, o2
	}
	p1 := make([]bls12381.G1Affine, n)
	p2 := make([]bls12381.G2Affine, n)
	scalars := make([]fr.Element, n)
	for i := 0; i < n; i++ {
		p1[i] = a1[i].(*gurvy381.G1).G1Affine
		p2[i] = a2[i].(*gurvy381.G2).G2Affine
		scalars[i] = gurvy381.ZrValue(b[i])
	}
	check(func() C.int {
		return C.mlhip_msm_g1g2(C.MLHIP_CURVE_BLS12_381,
			unsafe.Pointer(&p1[0]), unsafe.Pointer(&p2[0]), unsafe.Pointer(&scalars[0]), 1, C.size_t(n), C.int(WindowC),
			unsafe.Pointer(&o1.G1Affine), unsafe.Pointer(&o2.G2Affine))
	})
	return o1, o2
}

// PairingBatch returns FExp(Pairing(g2s[i], g1s[i])) for every i with one kernel launch.
func (c *Curve) PairingBatch(g2s []driver.G2, g1s []driver.G1) []driver.Gt {
	n := len(g1s)
	if len(g2s) != n {
		panic("hip: PairingBatch length mismatch")
	}
	if n == 0 {
		return nil
	}
	if n < MinDevicePairingBatch {
		out := make([]driver.Gt, n)
		for i := range g1s {
			out[i] = c.Curve.FExp(c.Curve.Pairing(g2s[i], g1s[i]))
		}
		return out
	}
	p := make([]bls12381.G1Affine, n)
	q := make([]bls12381.G2Affine, n)
	for i := range g1s {
		p[i] = g1s[i].(*gurvy381.G1).G1Affine
		q[i] = g2s[i].(*gurvy381.G2).G2Affine
	}
	gts := make([]bls12381.GT, n)
	check(func() C.int {
		return C.mlhip_pairing_batch(C.MLHIP_CURVE_BLS12_381,
		unsafe.Pointer(&p[0]), unsafe.Pointer(&q[0]), C.size_t(n), unsafe.Pointer(&gts[0]))
	})
	out := make([]driver.Gt, n)
	for i := range gts {
		out[i] = &gurvy381.Gt{GT: gts[i]}
	}
	return out
}

// MulBatchG1 returns a[i].Mul(b[i]) for every i with one kernel launch (the batched form of G1.Mul,
// bls12-381.go:238-247; SURVEY.md 8f row 3).
func (c *Curve) MulBatchG1(a []driver.G1, b []driver.Zr) []driver.G1 {
	n := len(a)
	if len(b) != n {
		panic("hip: MulBatchG1 length mismatch")
	}
	if n == 0 {
		return nil
	}
	points := make([]bls12381.G1Affine, n)
	for i := range a {
		points[i] = a[i].(*gurvy381.G1).G1Affine
	}
	return c.scalarMulG1(points, 1, b)
}

// BaseMulBatchG1 returns base.Mul(b[i]) for every i: one base (a generator, a Pedersen base), many scalars.  From 2^12
// scalars on the library multiplies through a table of the base's multiples, which it keeps on the device: later calls
// with the same base skip the table's construction.
func (c *Curve) BaseMulBatchG1(base driver.G1, b []driver.Zr) []driver.G1 {
	if len(b) == 0 {
		return nil
	}
	return c.scalarMulG1([]bls12381.G1Affine{base.(*gurvy381.G1).G1Affine}, 0, b)
}

func (c *Curve) scalarMulG1(points []bls12381.G1Affine, stride int, b []driver.Zr) []driver.G1 {
	n := len(b)
	scalars := make([]fr.Element, n)
	for i := range b {
		scalars[i] = gurvy381.ZrValue(b[i])
	}
	res := make([]bls12381.G1Affine, n)
	check(func() C.int {
		return C.mlhip_scalar_mul(C.MLHIP_CURVE_BLS12_381, C.MLHIP_GROUP_G1,
			unsafe.Pointer(&points[0]), C.size_t(stride), unsafe.Pointer(&scalars[0]), 1, C.size_t(n), unsafe.Pointer(&res[0]))
	})
	out := make([]driver.G1, n)
	for i := range res {
		out[i] = &gurvy381.G1{G1Affine: res[i]}
	}
	return out
}

// MulBatchG2 / BaseMulBatchG2: the same for G2 (bls12-381.go:342-351).
func (c *Curve) MulBatchG2(a []driver.G2, b []driver.Zr) []driver.G2 {
	n := len(a)
	if len(b) != n {
		panic("hip: MulBatchG2 length mismatch")
	}
	if n == 0 {
		return nil
	}
	points := make([]bls12381.G2Affine, n)
	for i := range a {
		points[i] = a[i].(*gurvy381.G2).G2Affine
	}
	return c.scalarMulG2(points, 1, b)
}

func (c *Curve) BaseMulBatchG2(base driver.G2, b []driver.Zr) []driver.G2 {
	if len(b) == 0 {
		return nil
	}
	return c.scalarMulG2([]bls12381.G2Affine{base.(*gurvy381.G2).G2Affine}, 0, b)
}

func (c *Curve) scalarMulG2(points []bls12381.G2Affine, stride int, b []driver.Zr) []driver.G2 {
	n := len(b)
	scalars := make([]fr.Element, n)
	for i := range b {
		scalars[i] = gurvy381.ZrValue(b[i])
	}
	res := make([]bls12381.G2Affine, n)
	check(func() C.int {
		return C.mlhip_scalar_mul(C.MLHIP_CURVE_BLS12_381, C.MLHIP_GROUP_G2,
			unsafe.Pointer(&points[0]), C.size_t(stride), unsafe.Pointer(&scalars[0]), 1, C.size_t(n), unsafe.Pointer(&res[0]))
	})
	out := make([]driver.G2, n)
	for i := range res {
		out[i] = &gurvy381.G2{G2Affine: res[i]}
	}
	return out
}

// ExpBatch returns gts[i].Exp(b[i]) for every i with one kernel launch (Gt.Exp, bls12-381.go:399-407; SURVEY.md 8f row 2).
func (c *Curve) ExpBatch(gts []driver.Gt, b []driver.Zr) []driver.Gt {
	n := len(gts)
	if len(b) != n {
		panic("hip: ExpBatch length mismatch")
	}
	if n == 0 {
		return nil
	}
	in := make([]bls12381.GT, n)
	scalars := make([]fr.Element, n)
	for i := range gts {
		in[i] = gts[i].(*gurvy381.Gt).GT
		scalars[i] = gurvy381.ZrValue(b[i])
	}
	res := make([]bls12381.GT, n)
	check(func() C.int {
		return C.mlhip_gt_exp(C.MLHIP_CURVE_BLS12_381, unsafe.Pointer(&in[0]), unsafe.Pointer(&scalars[0]), 1, C.size_t(n),
			unsafe.Pointer(&res[0]))
	})
	out := make([]driver.Gt, n)
	for i := range res {
		out[i] = &gurvy381.Gt{GT: res[i]}
	}
	return out
}

// NewG1sFromCompressed decodes n compressed G1 points (the wire form of G1.Compressed, bls12-381.go:292-296)
// on the device: decompression, curve check and subgroup check per point.  It is the bulk form of
// NewG1FromCompressed (bls12-381.go:551-559) and panics like it on the first invalid encoding, with gnark's
// message class ("set bytes failed [...]"); the façade's deserializers recover the panic into an error
// (math.go:761-832).
func (c *Curve) NewG1sFromCompressed(raw []byte) []driver.G1 {
	const sz = bls12381.SizeOfG1AffineCompressed
	if len(raw)%sz != 0 {
		panic("set bytes failed [invalid length]")
	}
	n := len(raw) / sz
	if n == 0 {
		return nil
	}
	pts := make([]bls12381.G1Affine, n)
	status := make([]byte, n)
	check(func() C.int {
		return C.mlhip_g1_from_bytes(C.MLHIP_CURVE_BLS12_381, unsafe.Pointer(&raw[0]), C.size_t(n), 1, 1,
		unsafe.Pointer(&pts[0]), (*C.uchar)(unsafe.Pointer(&status[0])))
	})
	out := make([]driver.G1, n)
	for i := range pts {
		switch status[i] {
		case 0:
			out[i] = &gurvy381.G1{G1Affine: pts[i]}
		case 1:
			panic("set bytes failed [invalid point encoding]")
		case 2:
			panic("set bytes failed [invalid point: not on the curve]")
		default:
			panic("set bytes failed [invalid point: subgroup check failed]")
		}
	}
	return out
}

// G1sCompressed is the inverse: the compressed wire form of every point, concatenated.
func (c *Curve) G1sCompressed(pts []driver.G1) []byte {
	n := len(pts)
	if n == 0 {
		return nil
	}
	aff := make([]bls12381.G1Affine, n)
	for i := range pts {
		aff[i] = pts[i].(*gurvy381.G1).G1Affine
	}
	out := make([]byte, n*bls12381.SizeOfG1AffineCompressed)
	check(func() C.int {
		return C.mlhip_g1_to_bytes(C.MLHIP_CURVE_BLS12_381, unsafe.Pointer(&aff[0]), C.size_t(n), 1, unsafe.Pointer(&out[0]))
	})
	return out
}

// Bases is a G1 point table kept on the device (SURVEY.md 8f row 1): NewBases uploads it once, MultiScalarMul then
// moves only the scalars (32 bytes each) per call -- the shape of a prover with a fixed SRS.
type Bases struct {
	h *C.mlhip_bases
	n int
}

func (c *Curve) NewBases(points []driver.G1) *Bases {
	n := len(points)
	if n == 0 {
		panic("hip: NewBases needs at least one point")
	}
	aff := make([]bls12381.G1Affine, n)
	for i := range points {
		aff[i] = points[i].(*gurvy381.G1).G1Affine
	}
	b := &Bases{n: n}
	check(func() C.int {
		return C.mlhip_bases_create(C.MLHIP_CURVE_BLS12_381, C.MLHIP_GROUP_G1, unsafe.Pointer(&aff[0]), C.size_t(n),
		C.int(WindowC), &b.h)
	})
	return b
}

// MultiScalarMul returns sum_i [scalars[i]] bases[i] over the first len(scalars) bases.
func (b *Bases) MultiScalarMul(scalars []driver.Zr) driver.G1 {
	out := &gurvy381.G1{}
	if len(scalars) == 0 {
		return out
	}
	if len(scalars) > b.n {
		panic("hip: more scalars than resident bases")
	}
	sc := make([]fr.Element, len(scalars))
	for i := range scalars {
		sc[i] = gurvy381.ZrValue(scalars[i])
	}
	check(func() C.int {
		return C.mlhip_bases_msm(b.h, unsafe.Pointer(&sc[0]), 1, C.size_t(len(sc)), unsafe.Pointer(&out.G1Affine))
	})
	return out
}

// CheckedSubgroup reports whether the library verified, on the device, that every point of the table lies in G1.  A
// BLS12-377 table that passes has its bucket sums done in twisted Edwards coordinates (7 field products per addition
// instead of 10); one that does not keeps the Weierstrass kernels and gnark's result for that input.
func (b *Bases) CheckedSubgroup() bool {
	return b.h != nil && C.mlhip_bases_checked_subgroup(b.h) == 1
}

// ShiftedTables reports whether the handle keeps shifted-base tables (include/mlhip.h: mlhip_bases_create): 2^off(j) P_i for
// every digit position, so that all digits of all scalars share one bucket set.
func (b *Bases) ShiftedTables() bool {
	if b.h == nil {
		return false
	}
	p := C.mlhip_bases_plan(b.h)
	if p == nil {
		return false
	}
	var t [11]C.float
	return C.mlhip_msm_plan_timings(p, &t[0], 11) >= 11 && t[10] == 1
}

func (b *Bases) Close() {
	if b.h != nil {
		C.mlhip_bases_destroy(b.h)
		b.h = nil
	}
}
