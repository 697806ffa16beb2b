package hip

// Deferred cross-check (SURVEY.md 8c): run on the first machine that has Go >= 1.25.7, the module
// cache of go.mod and an MI355X.  Clone of Test381Compat (math_test.go:879-911): the HIP backend
// against the gurvy and kilic drivers, comparing serialized bytes.  Never executed so far.

import (
	"testing"

	math "github.com/IBM/mathlib"
	"github.com/IBM/mathlib/driver"
	"github.com/stretchr/testify/assert"
)

func TestHipCompat(t *testing.T) {
	hip := NewCurve()
	gurvy := math.Curves[math.BLS12_381_GURVY]
	kilic := math.Curves[math.BLS12_381]

	const n = 1000
	g1s := make([]*math.G1, n)
	zrs := make([]*math.Zr, n)
	hg1 := make([]driver.G1, n)
	hzr := make([]driver.Zr, n)
	rng, err := gurvy.Rand()
	assert.NoError(t, err)
	for i := 0; i < n; i++ {
		g1s[i] = gurvy.GenG1.Mul(gurvy.NewRandomZr(rng))
		zrs[i] = gurvy.NewRandomZr(rng)
		hg1[i] = math.DriverG1(g1s[i]) // accessor for G1.g1, see INTEGRATION.md
		hzr[i] = math.DriverZr(zrs[i])
	}
	want := gurvy.MultiScalarMul(g1s, zrs)
	got := hip.MultiScalarMul(hg1, hzr)
	assert.Equal(t, want.Bytes(), got.Bytes())
	assert.Equal(t, want.Compressed(), got.Compressed())

	// FExp(Pairing) against gurvy and kilic bytes
	r := gurvy.NewRandomZr(rng)
	p := gurvy.GenG1.Mul(r)
	q := gurvy.GenG2.Mul(r)
	wantGt := gurvy.FExp(gurvy.Pairing(q, p))
	gotGt := hip.FExp(hip.Pairing(math.DriverG2(q), math.DriverG1(p)))
	assert.Equal(t, wantGt.Bytes(), gotGt.Bytes())
	kp, _ := kilic.NewG1FromBytes(p.Bytes())
	kq, _ := kilic.NewG2FromBytes(q.Bytes())
	assert.Equal(t, kilic.FExp(kilic.Pairing(kq, kp)).Bytes(), gotGt.Bytes())
}
