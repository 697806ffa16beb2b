/*
MI355X backend for the gurvy BLS12-377 driver -- the curve of BASELINE config 5 (2^22-point G1 MSM, the
recursive-SNARK field).

Like bn254_hip.go this file is meant to be dropped INTO package gurvy, next to bls12-377.go: the element types
bls12377G1 / bls12377G2 / bls12377Gt are unexported (driver/gurvy/bls12-377.go:21-219).  It overrides only
MultiScalarMul (driver/gurvy/bls12-377.go:229-242) and adds the batched entry points; Pairing / Pairing2 / FExp
(:244-264) and everything else are inherited from Bls12_377 by embedding (a single pairing occupies one lane pair of
the GPU and takes milliseconds there; gnark does it in about one on a CPU core).

hipCheck (bn254_hip.go) pins the goroutine to its OS thread for the call, because the library reports errors per
thread.  NOTE: never compiled (no Go toolchain in the build image); the same C ABI is exercised on the GPU for this
curve by the C++ and Python mirrors of the driver interface (include/mlhip_driver.hpp, mathlib_amd/driver.py) and at
config 5's full size by tests/test_gpu_fullsize.py.
*/
package gurvy

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../mathlib_amd -lmlhip -Wl,-rpath,${SRCDIR}/../../../mathlib_amd
#include "mlhip.h"
*/
import "C"

import (
	"unsafe"

	"github.com/IBM/mathlib/driver"
	"github.com/IBM/mathlib/driver/common"
	bls12377 "github.com/consensys/gnark-crypto/ecc/bls12-377"
	"github.com/consensys/gnark-crypto/ecc/bls12-377/fr"
)

// Bls12_377Hip is Bls12_377 with its large multi-scalar multiplications and pairing batches moved to the GPU.
type Bls12_377Hip struct {
	Bls12_377
	// MinDeviceMSM: smaller MSMs stay on the CPU (see go/driver/hip/hip.go for how the default was chosen).
	MinDeviceMSM int
	// WindowC: Pippenger window, 0 = chosen from n by the library (BASELINE config 5 runs at 16).
	WindowC int
}

func NewBls12_377Hip() *Bls12_377Hip {
	if unsafe.Sizeof(bls12377.G1Affine{}) != 96 || unsafe.Sizeof(bls12377.G2Affine{}) != 192 ||
		unsafe.Sizeof(bls12377.GT{}) != 576 || unsafe.Sizeof(fr.Element{}) != 32 {
		panic("hip: gnark-crypto BLS12-377 element layout changed; libmlhip.so expects 96/192/576/32-byte elements")
	}
	return &Bls12_377Hip{Bls12_377: *NewBls12_377(), MinDeviceMSM: 32}
}

// MultiScalarMul replaces driver/gurvy/bls12-377.go:229-242.  The scalars are *common.BaseZr (big.Int, possibly
// negative or >= r): SetBigInt reduces them exactly as the CPU driver does (bls12-377.go:236) and yields Montgomery
// fr.Elements, which is what scalars_mont = 1 expects.  With a process device list (mlhip_init / MLHIP_DEVICES) the
// library shards a large call over the listed GPUs by itself.
func (c *Bls12_377Hip) MultiScalarMul(a []driver.G1, b []driver.Zr) driver.G1 {
	n := len(a)
	if n < c.MinDeviceMSM || len(b) != n {
		return c.Bls12_377.MultiScalarMul(a, b) // small, or the mismatched-length case whose error the driver drops
	}
	points := make([]bls12377.G1Affine, n)
	scalars := make([]fr.Element, n)
	for i := range a {
		points[i] = a[i].(*bls12377G1).G1Affine
		scalars[i].SetBigInt(&b[i].(*common.BaseZr).Int)
	}
	var result bls12377.G1Affine
	hipCheck(func() C.int {
		return C.mlhip_msm_g1(C.MLHIP_CURVE_BLS12_377, unsafe.Pointer(&points[0]), unsafe.Pointer(&scalars[0]), 1,
			C.size_t(n), C.int(c.WindowC), unsafe.Pointer(&result))
	})
	return &bls12377G1{result}
}

// MultiScalarMulG2 = sum of G2.Mul + Add (driver/gurvy/bls12-377.go:133-170) as one MSM (additive API).
func (c *Bls12_377Hip) MultiScalarMulG2(a []driver.G2, b []driver.Zr) driver.G2 {
	n := len(a)
	var result bls12377.G2Affine
	if n == 0 || len(b) != n {
		return &bls12377G2{result}
	}
	points := make([]bls12377.G2Affine, n)
	scalars := make([]fr.Element, n)
	for i := range a {
		points[i] = a[i].(*bls12377G2).G2Affine
		scalars[i].SetBigInt(&b[i].(*common.BaseZr).Int)
	}
	hipCheck(func() C.int {
		return C.mlhip_msm_g2(C.MLHIP_CURVE_BLS12_377, unsafe.Pointer(&points[0]), unsafe.Pointer(&scalars[0]), 1,
			C.size_t(n), C.int(c.WindowC), unsafe.Pointer(&result))
	})
	return &bls12377G2{result}
}

// PairingBatch returns FExp(Pairing(g2s[i], g1s[i])) for every i in one launch.
func (c *Bls12_377Hip) PairingBatch(g2s []driver.G2, g1s []driver.G1) []driver.Gt {
	n := len(g1s)
	if len(g2s) != n {
		panic("hip: PairingBatch length mismatch")
	}
	if n == 0 {
		return nil
	}
	p := make([]bls12377.G1Affine, n)
	q := make([]bls12377.G2Affine, n)
	for i := range g1s {
		p[i] = g1s[i].(*bls12377G1).G1Affine
		q[i] = g2s[i].(*bls12377G2).G2Affine
	}
	gts := make([]bls12377.GT, n)
	hipCheck(func() C.int {
		return C.mlhip_pairing_batch(C.MLHIP_CURVE_BLS12_377, unsafe.Pointer(&p[0]), unsafe.Pointer(&q[0]), C.size_t(n),
			unsafe.Pointer(&gts[0]))
	})
	out := make([]driver.Gt, n)
	for i := range gts {
		out[i] = &bls12377Gt{gts[i]}
	}
	return out
}

// Bls12_377Bases is a G1 point table kept on the device: NewBases uploads it once, MultiScalarMul then moves only the
// scalars per call -- the shape of a prover with a fixed SRS, and on this curve the fast path: when the table is made
// the library checks every point on the device (on the curve and in the prime-order subgroup, gnark's IsInSubGroup
// test), and a table that passes has its buckets summed and reduced in twisted Edwards coordinates (7 field products
// per addition instead of 10: 2^22 points 11.1 -> 9.1 ms on one MI355X).  A table with a point outside G1 keeps the
// Weierstrass kernels and gnark's result for that input; CheckedSubgroup tells which.
type Bls12_377Bases struct {
	h *C.mlhip_bases
	n int
}

func (c *Bls12_377Hip) NewBases(points []driver.G1) *Bls12_377Bases {
	n := len(points)
	if n == 0 {
		panic("hip: NewBases needs at least one point")
	}
	aff := make([]bls12377.G1Affine, n)
	for i := range points {
		aff[i] = points[i].(*bls12377G1).G1Affine
	}
	b := &Bls12_377Bases{n: n}
	hipCheck(func() C.int {
		return C.mlhip_bases_create(C.MLHIP_CURVE_BLS12_377, C.MLHIP_GROUP_G1, unsafe.Pointer(&aff[0]), C.size_t(n),
			C.int(c.WindowC), &b.h)
	})
	return b
}

// MultiScalarMul returns sum_i [scalars[i]] bases[i] over the first len(scalars) bases.
func (b *Bls12_377Bases) MultiScalarMul(scalars []driver.Zr) driver.G1 {
	var result bls12377.G1Affine
	if len(scalars) == 0 {
		return &bls12377G1{result}
	}
	if len(scalars) > b.n {
		panic("hip: more scalars than resident bases")
	}
	sc := make([]fr.Element, len(scalars))
	for i := range scalars {
		sc[i].SetBigInt(&scalars[i].(*common.BaseZr).Int)
	}
	hipCheck(func() C.int {
		return C.mlhip_bases_msm(b.h, unsafe.Pointer(&sc[0]), 1, C.size_t(len(sc)), unsafe.Pointer(&result))
	})
	return &bls12377G1{result}
}

// CheckedSubgroup: every point of the table was verified to lie in G1 (the table takes the twisted Edwards kernels).
func (b *Bls12_377Bases) CheckedSubgroup() bool {
	return b.h != nil && C.mlhip_bases_checked_subgroup(b.h) == 1
}

func (b *Bls12_377Bases) Close() {
	if b.h != nil {
		C.mlhip_bases_destroy(b.h)
		b.h = nil
	}
}
