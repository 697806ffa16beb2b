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

// TestHipDevicePaths forces the single-shot pairings onto the device (they default to the CPU driver, see
// DevicePairing) and checks the additive API: resident bases, the batched pairing, the bulk wire codec.
func TestHipDevicePaths(t *testing.T) {
	hip := NewCurve()
	gurvy := math.Curves[math.BLS12_381_GURVY]
	rng, err := gurvy.Rand()
	assert.NoError(t, err)

	DevicePairing = true
	defer func() { DevicePairing = false }()
	minBatch := MinDevicePairingBatch
	MinDevicePairingBatch = 1 // the 64-pair batch below must reach the device
	defer func() { MinDevicePairingBatch = minBatch }()
	r := gurvy.NewRandomZr(rng)
	p := gurvy.GenG1.Mul(r)
	q := gurvy.GenG2.Mul(r)
	want := gurvy.FExp(gurvy.Pairing(q, p))
	got := hip.FExp(hip.Pairing(math.DriverG2(q), math.DriverG1(p)))
	assert.Equal(t, want.Bytes(), got.Bytes())
	// the Miller-loop values themselves may differ between implementations; FExp of a CPU Miller loop must agree too
	DevicePairing = false
	cpuMiller := hip.Pairing(math.DriverG2(q), math.DriverG1(p))
	DevicePairing = true
	assert.Equal(t, want.Bytes(), hip.FExp(cpuMiller).Bytes())

	const n = 64
	g1s := make([]driver.G1, n)
	g2s := make([]driver.G2, n)
	zrs := make([]driver.Zr, n)
	for i := 0; i < n; i++ {
		s := gurvy.NewRandomZr(rng)
		g1s[i] = math.DriverG1(gurvy.GenG1.Mul(s))
		g2s[i] = math.DriverG2(gurvy.GenG2.Mul(s))
		zrs[i] = math.DriverZr(gurvy.NewRandomZr(rng))
	}
	batch := hip.PairingBatch(g2s, g1s)
	for i := 0; i < n; i++ {
		assert.Equal(t, hip.FExp(hip.Pairing(g2s[i], g1s[i])).Bytes(), batch[i].Bytes())
	}

	bases := hip.NewBases(g1s)
	defer bases.Close()
	assert.False(t, bases.CheckedSubgroup()) // BLS12-381 has no twisted Edwards model: its tables are not checked
	assert.Equal(t, hip.Curve.MultiScalarMul(g1s, zrs).Bytes(), bases.MultiScalarMul(zrs).Bytes())
	assert.Equal(t, hip.Curve.MultiScalarMul(g1s[:7], zrs[:7]).Bytes(), bases.MultiScalarMul(zrs[:7]).Bytes())

	// batched Mul / Gt.Exp against the embedded driver's single calls (bls12-381.go:238-247, :342-351, :399-407)
	m1 := hip.MulBatchG1(g1s, zrs)
	m2 := hip.MulBatchG2(g2s, zrs)
	b1 := hip.BaseMulBatchG1(g1s[3], zrs)
	b2 := hip.BaseMulBatchG2(g2s[3], zrs)
	ex := hip.ExpBatch(batch, zrs)
	for i := 0; i < n; i++ {
		assert.True(t, m1[i].Equals(g1s[i].Mul(zrs[i])))
		assert.True(t, m2[i].Equals(g2s[i].Mul(zrs[i])))
		assert.True(t, b1[i].Equals(g1s[3].Mul(zrs[i])))
		assert.True(t, b2[i].Equals(g2s[3].Mul(zrs[i])))
		assert.Equal(t, batch[i].Exp(zrs[i]).Bytes(), ex[i].Bytes())
	}
	assert.Equal(t, b1[5].Bytes(), hip.BaseMulBatchG1(g1s[3], zrs)[5].Bytes()) // the second call runs on the kept table

	wire := hip.G1sCompressed(g1s)
	for i := 0; i < n; i++ {
		assert.Equal(t, g1s[i].Compressed(), wire[i*48:(i+1)*48])
	}
	back := hip.NewG1sFromCompressed(wire)
	for i := 0; i < n; i++ {
		assert.True(t, back[i].Equals(g1s[i]))
	}
	wire[5] ^= 1 // almost surely off the curve or out of the subgroup now
	assert.Panics(t, func() { hip.NewG1sFromCompressed(wire) })
}
