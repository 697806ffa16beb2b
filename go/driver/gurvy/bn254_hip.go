/*
MI355X backend for the gurvy BN254 driver (the north star names BN254 next to BLS12-381).

The BN254 element types of package gurvy are unexported (bn254G1, bn254G2, bn254Gt: driver/gurvy/bn254.go:23-221),
so -- unlike the BLS12-381 shim in go/driver/hip, which wraps the exported types of driver/gurvy/bls12381 -- this file
is meant to be dropped INTO package gurvy, next to bn254.go.  It overrides only the data-parallel methods of Bn254
(driver/gurvy/bn254.go:232-267) and adds the batched entry points; everything else is inherited by embedding.
BLS12-377 (driver/gurvy/bls12-377.go:229-264) takes the same treatment in bls12-377_hip.go.

NOTE: never compiled (no Go toolchain in the build image); the same C ABI is exercised on the GPU for all three
curves by the C++ and Python mirrors of the driver interface (include/mlhip_driver.hpp, mathlib_amd/driver.py).
*/
package gurvy

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../mathlib_amd -lmlhip -Wl,-rpath,${SRCDIR}/../../../mathlib_amd
#include "mlhip.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"unsafe"

	"github.com/IBM/mathlib/driver"
	"github.com/IBM/mathlib/driver/common"
	"github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
)

// Bn254Hip is Bn254 with its large multi-scalar multiplications and pairing batches moved to the GPU.
type Bn254Hip struct {
	Bn254
	// MinDeviceMSM: smaller MSMs stay on the CPU (a 2-point MSM costs 0.23 ms on the device, mostly launch latency).
	MinDeviceMSM int
	// WindowC: Pippenger window, 0 = chosen from n by the library.
	WindowC int
}

func NewBn254Hip() *Bn254Hip {
	if unsafe.Sizeof(bn254.G1Affine{}) != 64 || unsafe.Sizeof(bn254.G2Affine{}) != 128 ||
		unsafe.Sizeof(bn254.GT{}) != 384 || unsafe.Sizeof(fr.Element{}) != 32 {
		panic("hip: gnark-crypto BN254 element layout changed; libmlhip.so expects 64/128/384/32-byte elements")
	}
	return &Bn254Hip{Bn254: *NewBn254(), MinDeviceMSM: 32}
}

// hipCheck runs one library call and panics with the library's message when it fails.  The message is kept per OS
// thread and goroutines migrate between threads from one cgo call to the next, so the goroutine is locked to its
// thread from the call until the message has been read.
func hipCheck(call func() C.int) {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := call(); rc != 0 {
		panic(fmt.Sprintf("hip: libmlhip error %d: %s", int(rc), C.GoString(C.mlhip_last_error())))
	}
}

// MultiScalarMul replaces driver/gurvy/bn254.go:232-245.  The scalars are *common.BaseZr (big.Int, possibly
// negative or >= r): SetBigInt reduces them exactly as the CPU driver does (bn254.go:239) and yields Montgomery
// fr.Elements, which is what scalars_mont = 1 expects.
func (c *Bn254Hip) MultiScalarMul(a []driver.G1, b []driver.Zr) driver.G1 {
	n := len(a)
	if n < c.MinDeviceMSM || len(b) != n {
		return c.Bn254.MultiScalarMul(a, b) // small, or the mismatched-length case whose error the driver drops
	}
	points := make([]bn254.G1Affine, n)
	scalars := make([]fr.Element, n)
	for i := range a {
		points[i] = a[i].(*bn254G1).G1Affine
		scalars[i].SetBigInt(&b[i].(*common.BaseZr).Int)
	}
	var result bn254.G1Affine
	hipCheck(func() C.int {
		return C.mlhip_msm_g1(C.MLHIP_CURVE_BN254, unsafe.Pointer(&points[0]), unsafe.Pointer(&scalars[0]), 1,
			C.size_t(n), C.int(c.WindowC), unsafe.Pointer(&result))
	})
	return &bn254G1{result}
}

// PairingBatch returns FExp(Pairing(g2s[i], g1s[i])) for every i in one launch (single-shot Pairing / Pairing2 / FExp
// stay on the CPU: one pairing occupies one lane pair of the GPU and takes milliseconds there).
func (c *Bn254Hip) PairingBatch(g2s []driver.G2, g1s []driver.G1) []driver.Gt {
	n := len(g1s)
	if len(g2s) != n {
		panic("hip: PairingBatch length mismatch")
	}
	if n == 0 {
		return nil
	}
	p := make([]bn254.G1Affine, n)
	q := make([]bn254.G2Affine, n)
	for i := range g1s {
		p[i] = g1s[i].(*bn254G1).G1Affine
		q[i] = g2s[i].(*bn254G2).G2Affine
	}
	gts := make([]bn254.GT, n)
	hipCheck(func() C.int {
		return C.mlhip_pairing_batch(C.MLHIP_CURVE_BN254, unsafe.Pointer(&p[0]), unsafe.Pointer(&q[0]), C.size_t(n),
			unsafe.Pointer(&gts[0]))
	})
	out := make([]driver.Gt, n)
	for i := range gts {
		out[i] = &bn254Gt{gts[i]}
	}
	return out
}
