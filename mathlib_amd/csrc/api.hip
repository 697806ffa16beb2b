// api.hip -- the extern "C" surface of libmlhip.so (include/mlhip.h): argument checking, device
// selection, host-buffer staging and dispatch to the per-curve translation units.  No kernels here.
// There is no CPU fallback: every compute entry point needs a HIP device (MLHIP_ENODEVICE otherwise).
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <string>

#include "ec.h"
#include "mlhip_internal.h"
#include "msm_body.h"

using namespace mlhip;

namespace {
thread_local std::string g_err;
thread_local int g_device_sel = -1;  // mlhip_set_device on this thread; -1: follow the process's device list
thread_local int g_device = 0;       // device of the call in progress on this thread (set by ensure_device)

// ---- the process's device list (mlhip_init / MLHIP_DEVICES) ---------------------------------------------------------
// one process, one host thread per device, the C ABI takes a device list.  A host-buffer MSM / pairing
// batch issued by a thread that has not pinned itself to one device (mlhip_set_device) is cut into contiguous shards,
// one per listed device, when it is large enough to pay (MLHIP_MULTI_MIN pairs, MLHIP_MULTI_MIN_PAIRINGS pairings).
std::mutex g_devs_mu;
std::vector<int> g_devs;
bool g_devs_set = false;
std::atomic<bool> g_devs_bad{false};  // (read without the lock by ensure_device / mlhip_get_devices) MLHIP_DEVICES did not parse: every compute call fails until mlhip_init / mlhip_shutdown
constexpr int MLHIP_MAX_DEVICES = 64;  // device indices 0 .. 63 (the per-device tables below are indexed by them)
size_t g_multi_min_msm = (size_t)1 << 21, g_multi_min_pairing = (size_t)1 << 17;

bool parse_device_list(const char* e, std::vector<int>& out) {
  out.clear();
  if (!e || !*e) return true;
  if (!strcmp(e, "all")) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    for (int i = 0; i < n && i < MLHIP_MAX_DEVICES; i++) out.push_back(i);
    return true;
  }
  const char* q = e;
  while (*q) {
    char* end = nullptr;
    long v = strtol(q, &end, 10);
    if (end == q || v < 0 || v >= MLHIP_MAX_DEVICES || out.size() >= 64) return false;
    out.push_back((int)v);
    q = end;
    if (*q == ',') q++;
    else if (*q) return false;
  }
  return true;
}

std::vector<int> device_list() {
  std::lock_guard<std::mutex> lk(g_devs_mu);
  if (!g_devs_set) {
    g_devs_set = true;
    g_devs_bad = !parse_device_list(getenv("MLHIP_DEVICES"), g_devs);
    if (g_devs_bad) g_devs.clear();
    if (const char* e = getenv("MLHIP_MULTI_MIN")) g_multi_min_msm = strtoull(e, nullptr, 10);
    if (const char* e = getenv("MLHIP_MULTI_MIN_PAIRINGS")) g_multi_min_pairing = strtoull(e, nullptr, 10);
  }
  return g_devs;
}

int ensure_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return mlhip_rt::fail(MLHIP_ENODEVICE,
                          std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count is 0"));
  int d = g_device_sel;
  const std::vector<int> l = device_list();
  if (g_devs_bad)
    return mlhip_rt::fail(MLHIP_EINVAL, "MLHIP_DEVICES is malformed (want \"all\" or a comma-separated list of device indices 0 .. 63)");
  if (d < 0) d = l.empty() ? 0 : l[0];
  if (d < 0 || d >= n || d >= MLHIP_MAX_DEVICES) return mlhip_rt::fail(MLHIP_EINVAL, "device index out of range");
  g_device = d;
  HIPCHK(hipSetDevice(d));
  return 0;
}

// Devices a call of `units` items issued by this thread is spread over: empty = stay on one device.
std::vector<int> spread_devices(size_t units, bool pairing) {
  if (g_device_sel >= 0) return {};
  std::vector<int> l = device_list();
  if (l.size() < 2 || units < (pairing ? g_multi_min_pairing : g_multi_min_msm)) return {};
  if (l.size() > units) l.resize(units);
  return l;
}

// fn(shard, lo, hi) runs on one host thread per listed device, with that device selected for the thread: contiguous
// shards [n r / D, n (r + 1) / D).  A device may be listed more than once (two shards in flight on it).
template <class Fn>
int run_on_devices(const std::vector<int>& devs, size_t n, Fn fn) {
  const size_t D = devs.size();
  std::vector<int> rcs(D, 0);
  std::vector<std::string> errs(D);
  auto body = [&](size_t r) {
    const int saved = g_device_sel;
    g_device_sel = devs[r];
    rcs[r] = fn(r, n * r / D, n * (r + 1) / D);
    if (rcs[r]) errs[r] = g_err;
    g_device_sel = saved;
  };
  std::vector<std::thread> th;
  th.reserve(D);
  for (size_t r = 1; r < D; r++) th.emplace_back(body, r);
  body(0);
  for (std::thread& t : th) t.join();
  for (size_t r = 0; r < D; r++)
    if (rcs[r]) return mlhip_rt::fail(rcs[r], "device " + std::to_string(devs[r]) + " (shard " + std::to_string(r) + "): " + errs[r]);
  return 0;
}

int ilog2(size_t v) {
  int l = 0;
  while (v > 1) {
    v >>= 1;
    l++;
  }
  return l;
}

// measured optimum on one MI355X (tools/sweep_window.py, profiles/r02_sweep_window.txt).  With the balanced window
// layout (msm_body.h: msm_win_layout) every width splits the scalar evenly, so the mid sizes no longer have to jump
// from 8 to 16: 13-14 bits win from 2^11 to 2^15 points (2^14: 0.81 ms instead of 1.11), 16 from 2^16 on.  For large n on
// the curves whose 254 / 255 scalar bits fit 15 windows of 17 bits (BLS12-377, BN254) one window less is one addition
// per scalar less (BLS12-377 2^22: 12.5 ms instead of 13.5); BLS12-381's 256 bits need 16 windows either way.
// Round 4 (profiles/r04_sweep_window.txt, the same sweep on this round's kernels): 10 bits from 2^10 to 2^11 points
// (0.47 / 0.49 ms against 0.50 at c = 8 / 0.53 at c = 13), and for BN254 (254-bit order, 10-limb field: its reduction weighs
// more against its additions) 15 bits from 2^16 to 2^17 points (0.505 / 0.587 ms against 0.554 / 0.617 at c = 16) -- not for
// BLS12-377, whose 253 bits also fit 17 windows of 15: 0.93 / 1.14 ms against 0.88 / 1.00 at c = 16.
int pick_window(size_t n, int fr_bits) {
  if (n <= 128) return 4;
  if (n <= 512) return 8;
  if (n <= 2048) return 10;
  if (n <= 8192) return 13;
  if (n <= 32768) return 14;
  if (n < ((size_t)1 << 18) && fr_bits == 254) return 15;
  if (n >= ((size_t)1 << 22) && msm_num_windows(fr_bits, 17) < msm_num_windows(fr_bits, 16)) return 17;
  return 16;
}

template <class F>
int host_sum(const void* pts, size_t n, void* out) {
  const Affine<F>* p = (const Affine<F>*)pts;
  XYZZ<F> acc;
  xyzz_set_inf<F>(acc);
  for (size_t i = 0; i < n; i++) xyzz_madd<F>(acc, p[i], false);
  Affine<F> r;
  xyzz_to_affine<F>(r, acc);
  memcpy(out, &r, sizeof(r));
  return 0;
}

struct Sizes {
  size_t fp, g1, g2, gt;
  int fr_bits;
};

bool curve_sizes(int curve, Sizes& s) {
  switch (curve) {
    case MLHIP_CURVE_BN254:
      s = {sizeof(Fp<Bn254>), sizeof(Affine<FpField<Bn254>>), sizeof(Affine<Fp2Field<Bn254>>), 12 * sizeof(Fp<Bn254>), Bn254::FR_BITS};
      return true;
    case MLHIP_CURVE_BLS12_381:
      s = {sizeof(Fp<Bls381>), sizeof(Affine<FpField<Bls381>>), sizeof(Affine<Fp2Field<Bls381>>), 12 * sizeof(Fp<Bls381>), Bls381::FR_BITS};
      return true;
    case MLHIP_CURVE_BLS12_377:
      s = {sizeof(Fp<Bls377>), sizeof(Affine<FpField<Bls377>>), sizeof(Affine<Fp2Field<Bls377>>), 12 * sizeof(Fp<Bls377>), Bls377::FR_BITS};
      return true;
    default:
      return false;
  }
}

int tu_pairing(int curve, int what, const void* d1, const void* d2, size_t ppp, size_t n, const void* din, void* dout,
               hipStream_t st) {
  switch (curve) {
    case MLHIP_CURVE_BN254: return mlhip_tu_pairing_Bn254(what, d1, d2, ppp, n, din, dout, st);
    case MLHIP_CURVE_BLS12_381: return mlhip_tu_pairing_Bls381(what, d1, d2, ppp, n, din, dout, st);
    case MLHIP_CURVE_BLS12_377: return mlhip_tu_pairing_Bls377(what, d1, d2, ppp, n, din, dout, st);
    default: return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  }
}

// ---- one host-buffer call: a leased non-blocking stream with its own scratch arena ------------------------------
// hipMalloc / hipFree per call and the null stream would serialize concurrent callers (hipFree waits for the whole
// device).  Every host-buffer entry point below leases a (stream, arena) pair from a small per-device free list:
// device buffers are bump-allocated from the arena (one hipMalloc, grown when a call needs more), copies and kernels
// go to the leased stream, and the call waits for that stream only.  (hipMallocAsync was tried first and gave
// intermittently wrong results on this runtime.)
struct Lease {
  hipStream_t st = nullptr;
  char* arena = nullptr;
  size_t cap = 0;
};
std::mutex g_leases_mu;
std::vector<Lease> g_leases[MLHIP_MAX_DEVICES];  // idle leases per device

struct HostCall {
  int device;
  Lease l;
  size_t used = 0;
  int rc = 0;
  HostCall() : device(g_device) {
    {
      std::lock_guard<std::mutex> lk(g_leases_mu);
      std::vector<Lease>& idle = g_leases[device];
      if (!idle.empty()) {
        l = idle.back();
        idle.pop_back();
      }
    }
    if (!l.st && hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking) != hipSuccess) {
      l.st = nullptr;
      rc = mlhip_rt::fail(MLHIP_EHIP, "hipStreamCreate failed");
    }
  }
  ~HostCall() {
    if (!l.st) return;
    (void)hipStreamSynchronize(l.st);  // nothing of this call is left in flight when the caller gets its buffers back
    std::lock_guard<std::mutex> lk(g_leases_mu);
    g_leases[device].push_back(l);
  }
  // call once, before the first dev() / up(): the total number of device bytes this call needs
  void reserve(size_t bytes) {
    if (rc) return;
    bytes += 8 * 256;  // alignment slack for up to 8 buffers
    if (bytes <= l.cap) return;
    if (l.arena) (void)hipFree(l.arena);  // idle lease: nothing of ours is in flight
    l.arena = nullptr;
    l.cap = 0;
    const size_t want = bytes + bytes / 4;
    if (hipMalloc((void**)&l.arena, want) != hipSuccess) {
      (void)hipGetLastError();
      l.arena = nullptr;
      rc = mlhip_rt::fail(MLHIP_ENOMEM, "hipMalloc of the call's scratch failed");
      return;
    }
    l.cap = want;
  }
  void* dev(size_t bytes) {
    if (rc) return nullptr;
    const size_t start = (used + 255) & ~(size_t)255;
    if (start + bytes > l.cap) {
      rc = mlhip_rt::fail(MLHIP_EINVAL, "internal: scratch arena overrun");
      return nullptr;
    }
    used = start + bytes;
    return l.arena + start;
  }
  void* up(const void* src, size_t bytes) {
    void* p = dev(bytes);
    if (p && bytes && hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, l.st) != hipSuccess)
      rc = mlhip_rt::fail(MLHIP_EHIP, "hipMemcpy H2D failed");
    return rc ? nullptr : p;
  }
  int down(void* dst, const void* dsrc, size_t bytes) {
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(dst, dsrc, bytes, hipMemcpyDeviceToHost, l.st);
    if (e == hipSuccess) e = hipStreamSynchronize(l.st);
    if (e != hipSuccess) rc = mlhip_rt::fail(MLHIP_EHIP, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
    return rc;
  }
};

// host-buffer wrapper around the pairing kernels: upload, run, download
int pairing_host(int curve, int what, const void* g1, const void* g2, size_t ppp, size_t n, const void* in, void* out) {
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (n == 0) return 0;
  if (!out || (what == 1 ? !in : (!g1 || !g2))) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  {
    // independent per element: a large batch is split over the process's devices, no exchange at all
    const std::vector<int> devs = spread_devices(n, true);
    if (!devs.empty())
      return run_on_devices(devs, n, [&](size_t, size_t lo, size_t hi) {
        return pairing_host(curve, what, g1 ? (const char*)g1 + lo * ppp * sz.g1 : nullptr,
                            g2 ? (const char*)g2 + lo * ppp * sz.g2 : nullptr, ppp, hi - lo,
                            in ? (const char*)in + lo * sz.gt : nullptr, (char*)out + lo * sz.gt);
      });
  }
  int rc = ensure_device();
  if (rc) return rc;
  HostCall hc;
  void *d1 = nullptr, *d2 = nullptr, *din = nullptr;
  if (what == 1) {
    hc.reserve(2 * n * sz.gt);
    din = hc.up(in, n * sz.gt);
  } else {
    hc.reserve(n * ppp * (sz.g1 + sz.g2) + n * sz.gt);
    d1 = hc.up(g1, n * ppp * sz.g1);
    d2 = hc.up(g2, n * ppp * sz.g2);
  }
  void* dout = hc.dev(n * sz.gt);
  if (hc.rc) return hc.rc;
  rc = tu_pairing(curve, what, d1, d2, ppp, n, din, dout, hc.l.st);
  if (rc) return rc;
  return hc.down(out, dout, n * sz.gt);
}

// ---- a small pool of plans + device input buffers for the host-buffer entry points ---------------------------
// The reference's MultiScalarMul takes fresh host slices per call; creating and, above all, destroying a plan
// (a dozen hipFree's, ~2.5 ms) and the input buffers per call cost as much as the kernels of a 2^20-point MSM.
// Entries are reused across calls and threads when curve / group / window / device match and the size fits
// (capacity between n and 4 n).  At most POOL_MAX entries and POOL_MAX_BYTES of device memory stay allocated (an entry
// is ~0.6 GB at n = 2^20, ~10 GB at 2^24; many goroutines with small MSMs each find their own entry);
// MLHIP_NO_PLAN_CACHE=1 disables the pool, mlhip_release_cache() empties it.
struct PoolEntry {
  mlhip_msm_plan* plan = nullptr;
  void *d_pts = nullptr, *d_sc = nullptr;
  hipStream_t stream = nullptr;  // the entry's own non-blocking stream: concurrent callers do not meet on the null stream
  int curve = 0, group = 0, c = 0, device = 0;
  size_t cap = 0;
  bool busy = false, pooled = false;
  unsigned long stamp = 0;
  size_t bytes = 0;  // device memory the entry took (free memory before - after its creation)
};
constexpr size_t POOL_MAX = 16;  // per device
constexpr size_t POOL_MAX_BYTES = (size_t)32 << 30;  // per device
std::mutex g_pool_mu;
std::vector<PoolEntry*> g_pool;
unsigned long g_pool_clock = 0;

void pool_free_entry(PoolEntry* e) {
  (void)hipSetDevice(e->device);
  if (e->d_pts) (void)hipFree(e->d_pts);
  if (e->d_sc) (void)hipFree(e->d_sc);
  if (e->plan) mlhip_msm_plan_destroy(e->plan);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

PoolEntry* pool_acquire(int curve, int group, int c, size_t n, size_t ptsz, int& rc) {
  const char* off = getenv("MLHIP_NO_PLAN_CACHE");
  const bool use_pool = !(off && off[0] == '1');
  std::vector<PoolEntry*> victims;
  if (use_pool) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (PoolEntry* e : g_pool)
      if (!e->busy && e->curve == curve && e->group == group && e->c == c && e->device == g_device && e->cap >= n &&
          e->cap <= 4 * n) {
        e->busy = true;
        e->stamp = ++g_pool_clock;
        return e;
      }
    // evict this device's least recently used idle entries while its share of the pool is full or over budget
    for (;;) {
      size_t total = 0, count = 0;
      for (PoolEntry* e : g_pool)
        if (e->device == g_device) {
          total += e->bytes;
          count++;
        }
      if (count < POOL_MAX && total <= POOL_MAX_BYTES) break;
      size_t vi = g_pool.size();
      for (size_t i = 0; i < g_pool.size(); i++)
        if (g_pool[i]->device == g_device && !g_pool[i]->busy && (vi == g_pool.size() || g_pool[i]->stamp < g_pool[vi]->stamp)) vi = i;
      if (vi == g_pool.size()) break;  // everything is in use
      victims.push_back(g_pool[vi]);
      g_pool.erase(g_pool.begin() + vi);
    }
  }
  for (PoolEntry* v : victims) pool_free_entry(v);
  size_t free_before = 0, free_after = 0, total_mem = 0;
  (void)hipMemGetInfo(&free_before, &total_mem);
  PoolEntry* e = new PoolEntry();
  e->curve = curve;
  e->group = group;
  e->c = c;
  e->device = g_device;
  e->cap = n;
  e->busy = true;
  for (int attempt = 0;; attempt++) {
    rc = mlhip_msm_plan_create(curve, group, n, c, &e->plan);
    if (!rc && (hipMalloc(&e->d_pts, n * ptsz) != hipSuccess || hipMalloc(&e->d_sc, n * 32) != hipSuccess))
      rc = mlhip_rt::fail(MLHIP_ENOMEM, "hipMalloc of MSM inputs failed");
    if (!rc || attempt == 1 || !use_pool) break;
    // out of device memory with idle entries pooled: give them back and try once more
    if (e->d_pts) (void)hipFree(e->d_pts);
    if (e->d_sc) (void)hipFree(e->d_sc);
    if (e->plan) mlhip_msm_plan_destroy(e->plan);
    e->d_pts = e->d_sc = nullptr;
    e->plan = nullptr;
    (void)hipGetLastError();
    mlhip_release_cache();
  }
  if (!rc && hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
    rc = mlhip_rt::fail(MLHIP_EHIP, "hipStreamCreate failed");
  if (rc) {
    pool_free_entry(e);
    return nullptr;
  }
  (void)hipMemGetInfo(&free_after, &total_mem);
  e->bytes = free_before > free_after ? free_before - free_after : 0;
  if (use_pool) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t count = 0;
    for (PoolEntry* o : g_pool) count += o->device == e->device;
    if (count < POOL_MAX) {
      e->pooled = true;
      e->stamp = ++g_pool_clock;
      g_pool.push_back(e);
    }
  }
  return e;
}

void pool_release(PoolEntry* e, bool failed) {
  if (e->pooled && failed) {  // do not keep an entry whose last run ended in an error
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++)
      if (g_pool[i] == e) {
        g_pool.erase(g_pool.begin() + i);
        break;
      }
    e->pooled = false;
  }
  if (!e->pooled) {
    pool_free_entry(e);
    return;
  }
  std::lock_guard<std::mutex> lk(g_pool_mu);
  e->busy = false;
}

// Number of segments a host-buffer MSM is streamed in (1 = one upload, one pass).  Measured on MI355X / PCIe gen5
// (tools/perf_hostapi.py): from 2^19 points the transfer is worth hiding; MLHIP_STREAM_SEGMENTS overrides (0/1 = off).
int stream_segments(int group, size_t n, const mlhip_msm_plan* plan) {
  // G1 and G2 stream through the carry-free kernels and their bucket state (always there unless MLHIP_ACC32=1, which runs
  // one pass) -- the condition stream_begin checks
  (void)group;
  if (!plan->aux || !plan->d_points28) return 1;
  if (getenv("MLHIP_STREAM_SCHEDULE")) return n >= 2 ? 2 : 1;  // explicit segment weights (msm_plan.h: stream_schedule), any n
  if (const char* e = getenv("MLHIP_STREAM_SEGMENTS")) {
    int v = atoi(e);
    if (v < 2) return 1;
    if (v > MLHIP_MAX_SEGMENTS) v = MLHIP_MAX_SEGMENTS;
    return n >= (size_t)v ? v : 1;
  }
  // segments of 2^18 pairs: at 2^20 the call drops from 5.9 to 4.5 ms, at 2^22 from 21.9 to 12.8 ms (the device-only time)
  // G2 (BLS12-381): segments of 2^17 pairs, 14.9 -> 11.4 ms at 2^20.  For G1 from 2^20 pairs on the count returned here only
  // says "stream": plan_stream replaces the equal segments by a growing schedule (stream_schedule, round 4)
  const size_t k = n >> (group == MLHIP_GROUP_G1 ? 18 : 17);
  return k < 2 ? 1 : (k > MLHIP_MAX_SEGMENTS ? MLHIP_MAX_SEGMENTS : (int)k);
}

int tu_plan_stream(mlhip_msm_plan* p, void* d_pts, void* d_sc, const void* points, const void* scalars, int mont, size_t n,
                   int segments, hipStream_t st) {
  int rc;
  switch (p->curve) {
    case MLHIP_CURVE_BN254: rc = mlhip_tu_plan_stream_Bn254(p, d_pts, d_sc, points, scalars, mont, n, segments, st); break;
    case MLHIP_CURVE_BLS12_381: rc = mlhip_tu_plan_stream_Bls381(p, d_pts, d_sc, points, scalars, mont, n, segments, st); break;
    default: rc = mlhip_tu_plan_stream_Bls377(p, d_pts, d_sc, points, scalars, mont, n, segments, st); break;
  }
  if (rc) {
    // a failure part-way: copies from the caller's buffers may still be queued -- let them drain before the caller gets
    // its memory back, and leave the plan reusable
    (void)hipDeviceSynchronize();
    p->pending = false;
  }
  return rc;
}

int host_group_sum(int curve, int group, const void* pts, size_t n, void* out) {
  return group == MLHIP_GROUP_G1 ? mlhip_g1_sum(curve, pts, n, out) : mlhip_g2_sum(curve, pts, n, out);
}

int msm_host_buffers(int curve, int group, const void* points, const void* scalars, int mont, size_t n, int window_c,
                     void* out);

// One MSM over several devices (reference semantics math.go:960-969 /
// driver/gurvy/bls12381/bls12-381.go:766-783): contiguous shards of the pairs, one host thread per device running the
// whole single-device pipeline on its shard (its own pooled plan, its own PCIe link), the per-device partial sums --
// already in host memory, where each shard's Horner tail leaves them -- added on the host.  The caller wants the sum in
// host memory, so there is nothing for a device-side collective to do here; the RCCL all-gather lives in the
// process-per-GPU form (mathlib_amd/dist.py), where every rank wants the total.
int msm_multi(const std::vector<int>& devs, int curve, int group, const void* points, const void* scalars, int mont,
              size_t n, int window_c, void* out, size_t ptsz) {
  std::vector<char> partial(devs.size() * ptsz);
  int rc = run_on_devices(devs, n, [&](size_t r, size_t lo, size_t hi) {
    return msm_host_buffers(curve, group, (const char*)points + lo * ptsz, (const char*)scalars + lo * 32, mont, hi - lo,
                            window_c, &partial[r * ptsz]);
  });
  if (rc) return rc;
  return host_group_sum(curve, group, partial.data(), devs.size(), out);
}

int msm_host_buffers(int curve, int group, const void* points, const void* scalars, int mont, size_t n, int window_c,
                     void* out) {
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (!out) return mlhip_rt::fail(MLHIP_EINVAL, "null output pointer");
  const size_t ptsz = group == MLHIP_GROUP_G1 ? sz.g1 : sz.g2;
  if (n == 0) {
    // the point at infinity, as gnark's MultiExp gives for empty slices (and, via the dropped
    // error, for mismatched lengths: bls12-381.go:777)
    memset(out, 0, ptsz);
    return 0;
  }
  if (!points || !scalars) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  {
    const std::vector<int> devs = spread_devices(n, false);
    if (!devs.empty()) return msm_multi(devs, curve, group, points, scalars, mont, n, window_c, out, ptsz);
  }
  if (window_c == 0) window_c = pick_window(n, sz.fr_bits);
  int rc = ensure_device();
  if (rc) return rc;
  PoolEntry* e = pool_acquire(curve, group, window_c, n, ptsz, rc);
  if (!e) return rc;
  const int segments = stream_segments(group, n, e->plan);
  do {
    if (segments > 1) {
      // large G1 MSMs: upload, sort and accumulate segment by segment, so the PCIe transfer hides under the kernels
      rc = tu_plan_stream(e->plan, e->d_pts, e->d_sc, points, scalars, mont, n, segments, e->stream);
      if (!rc) rc = mlhip_msm_finish(e->plan, out, nullptr);
      break;
    }
    // scalars first (the sort needs only them); the points follow on the plan's auxiliary stream while the sort runs
    if (hipMemcpy(e->d_sc, scalars, n * 32, hipMemcpyHostToDevice) != hipSuccess) {
      rc = mlhip_rt::fail(MLHIP_EHIP, "hipMemcpy of MSM scalars failed");
      break;
    }
    e->plan->upload_src = points;
    e->plan->upload_bytes = n * ptsz;
    rc = mlhip_msm_run(e->plan, e->d_pts, e->d_sc, mont, n, e->stream, out, nullptr);
    e->plan->upload_src = nullptr;
  } while (0);
  pool_release(e, rc != 0);
  return rc;
}

}  // namespace

namespace mlhip_rt {
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

// ---- host worker threads (the per-window half of an MSM's host tail) ----------------------------------------------------
// One small pool per process, started on first use and never joined (a Go process loads the library for its lifetime;
// the workers sleep on a condition variable between calls, after a short spin so that back-to-back MSMs do not pay a
// futex wake-up each).  One call at a time owns the pool; a second caller arriving meanwhile does its own jobs.
// A call never waits for a worker longer than the job would take the caller itself: jobs are pure functions of a blob
// the call COPIES into the (reference-counted) job record, so the caller can run a job a worker has claimed but not
// finished a second time and take whichever result is there first -- a worker that the scheduler parked behind a
// spinning thread for a timeslice (seen: 6 ms, once in 200 MSMs) costs nothing, and a worker that wakes up late only
// ever touches the record, never the caller's memory.
namespace {
struct HostJob {
  void (*fn)(const void*, int, void*);
  int njobs = 0;
  size_t out_stride = 0;
  std::vector<unsigned char> in, out_worker, out_caller;
  std::atomic<int> next{0};
  std::atomic<int> state[64];  // 0 not started, 1 claimed by a worker, 2 worker's result valid, 3 caller's result valid
};
struct HostPool {
  std::mutex owner;  // held by the call that is using the workers
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<unsigned long> gen{0};
  std::shared_ptr<HostJob> job;  // guarded by mu
  int workers = 0;
  int spin = 2000;  // pause iterations a worker spins for the next job before it sleeps (host_pool_start)
};
HostPool* g_host_pool = nullptr;
std::once_flag g_host_pool_once;

void host_worker(HostPool* pool) {
  unsigned long seen = 0;
  for (;;) {
    // spin for a few tens of microseconds (a job is often followed by another one at once), then sleep
    bool fresh = false;
    for (int i = 0; i < pool->spin && !fresh; i++) {
      fresh = pool->gen.load(std::memory_order_acquire) != seen;
      if (!fresh) __builtin_ia32_pause();
    }
    std::shared_ptr<HostJob> j;
    {
      std::unique_lock<std::mutex> lk(pool->mu);
      pool->cv.wait(lk, [&] { return pool->gen.load(std::memory_order_relaxed) != seen; });
      seen = pool->gen.load(std::memory_order_relaxed);
      j = pool->job;
    }
    if (!j) continue;
    for (;;) {
      const int k = j->next.fetch_add(1, std::memory_order_relaxed);
      if (k >= j->njobs) break;
      int expect = 0;
      if (!j->state[k].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) continue;
      j->fn(j->in.data(), k, j->out_worker.data() + (size_t)k * j->out_stride);
      expect = 1;
      (void)j->state[k].compare_exchange_strong(expect, 2, std::memory_order_acq_rel);  // lost: the caller redid it
    }
  }
}

// Pool size: MLHIP_HOST_THREADS (threads per call incl. the caller) or, by default, what this PROCESS may use -- the
// affinity mask, not the machine's core count -- divided among the ranks that share the host (LOCAL_WORLD_SIZE, set by
// torch.distributed.run: `bench.py --gpus 8` is 8 processes on one host, each with its own pool, beside torch's threads):
// min(8, share) for a lone process, min(8, share / 2) when several ranks share the host, and the workers then spin a
// tenth as long before they sleep (a spinning worker of one rank is a core another rank's tail cannot have).
void host_pool_start() {
  int total = 0;
  if (const char* e = getenv("MLHIP_HOST_THREADS")) total = atoi(e);
  int local_world = 1;
  if (const char* e = getenv("LOCAL_WORLD_SIZE")) local_world = atoi(e) > 1 ? atoi(e) : 1;
  if (total <= 0) {
    int cores = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) cores = CPU_COUNT(&set);
    if (cores <= 0) cores = (int)std::thread::hardware_concurrency();
    if (cores <= 0) cores = 1;
    int share = cores / local_world;
    if (local_world > 1) share /= 2;
    total = share >= 8 ? 8 : (share > 0 ? share : 1);
  }
  if (total > 64) total = 64;
  HostPool* pool = new HostPool;  // never freed: the workers outlive every static destructor
  pool->workers = total - 1;
  pool->spin = local_world > 1 ? 200 : 2000;
  for (int i = 0; i < pool->workers; i++) {
    std::thread t(host_worker, pool);
    (void)pthread_setname_np(t.native_handle(), "mlhip-host");  // tests count them (tests/test_dist_gpu.py)
    t.detach();
  }
  g_host_pool = pool;
}
}  // namespace

void host_parallel(int njobs, void (*fn)(const void*, int, void*), const void* in, size_t in_bytes, void* out, size_t out_stride) {
  std::call_once(g_host_pool_once, host_pool_start);
  HostPool* pool = g_host_pool;
  std::unique_lock<std::mutex> own(pool->owner, std::try_to_lock);
  if (njobs < 2 || njobs > 64 || pool->workers == 0 || !own.owns_lock()) {
    for (int k = 0; k < njobs; k++) fn(in, k, (unsigned char*)out + (size_t)k * out_stride);
    return;
  }
  auto j = std::make_shared<HostJob>();
  j->fn = fn;
  j->njobs = njobs;
  j->out_stride = out_stride;
  j->in.assign((const unsigned char*)in, (const unsigned char*)in + in_bytes);
  j->out_worker.resize((size_t)njobs * out_stride);
  j->out_caller.resize((size_t)njobs * out_stride);
  for (int k = 0; k < njobs; k++) j->state[k].store(0, std::memory_order_relaxed);
  {
    std::lock_guard<std::mutex> lk(pool->mu);
    pool->job = j;
    pool->gen.fetch_add(1, std::memory_order_release);
  }
  pool->cv.notify_all();
  // the caller takes jobs from the top end, the workers from the bottom
  for (int k = njobs - 1; k >= 0; k--) {
    int st = j->state[k].load(std::memory_order_acquire);
    if (st == 0) {
      int expect = 0;
      if (j->state[k].compare_exchange_strong(expect, 3, std::memory_order_acq_rel)) {
        // claimed and (below) computed by the caller; nobody else looks at out_caller before the call returns
        fn(j->in.data(), k, j->out_caller.data() + (size_t)k * out_stride);
        continue;
      }
      st = expect;
    }
    if (st == 1) {
      // a worker is on it: give it about the time of one job, then do the job here as well
      for (int spin = 0; spin < 400 && j->state[k].load(std::memory_order_acquire) == 1; spin++) __builtin_ia32_pause();
      if (j->state[k].load(std::memory_order_acquire) == 1) {
        fn(j->in.data(), k, j->out_caller.data() + (size_t)k * out_stride);
        int expect = 1;
        (void)j->state[k].compare_exchange_strong(expect, 3, std::memory_order_acq_rel);  // lost: the worker's is there
      }
    }
  }
  for (int k = 0; k < njobs; k++) {
    const int st = j->state[k].load(std::memory_order_acquire);
    const unsigned char* src = (st == 2 ? j->out_worker.data() : j->out_caller.data()) + (size_t)k * out_stride;
    memcpy((unsigned char*)out + (size_t)k * out_stride, src, out_stride);
  }
}
}  // namespace mlhip_rt

extern "C" {

// 104 = round 4; bit 16 set in the test build (MLHIP_BUILD_ALT=1: the second implementations are compiled in)
int mlhip_version(void) { return 104 | (kBuildAlt ? 0x10000 : 0); }

const char* mlhip_last_error(void) { return g_err.c_str(); }

int mlhip_device_count(int* count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) n = 0;
  if (count) *count = n;
  return 0;
}

int mlhip_set_device(int device) {
  if (device < -1 || device >= MLHIP_MAX_DEVICES) return mlhip_rt::fail(MLHIP_EINVAL, "device index: -1 (unpin) or 0 .. 63");
  g_device_sel = device;
  return 0;
}

int mlhip_init(const int* devices, int n_devices) {
  if (n_devices < 0 || n_devices > 64 || (n_devices > 0 && !devices))
    return mlhip_rt::fail(MLHIP_EINVAL, "device list: 0 .. 64 entries");
  std::vector<int> l;
  if (n_devices == 0) {
    parse_device_list("all", l);
  } else {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) count = 0;  // no device here: the first compute call reports it
    for (int i = 0; i < n_devices; i++) {
      if (devices[i] < 0 || devices[i] >= MLHIP_MAX_DEVICES || (count > 0 && devices[i] >= count))
        return mlhip_rt::fail(MLHIP_EINVAL, "device list: index out of range (0 .. min(63, device count - 1))");
      l.push_back(devices[i]);
    }
  }
  (void)device_list();  // the environment's thresholds are read once, before the list is replaced
  std::lock_guard<std::mutex> lk(g_devs_mu);
  g_devs = l;
  g_devs_bad = false;
  return 0;
}

int mlhip_get_devices(int* devices, int cap) {
  const std::vector<int> l = device_list();
  if (g_devs_bad) return mlhip_rt::fail(MLHIP_EINVAL, "MLHIP_DEVICES is malformed");
  for (size_t i = 0; i < l.size() && (int)i < cap; i++)
    if (devices) devices[i] = l[i];
  return (int)l.size();
}

int mlhip_shutdown(void) {
  mlhip_release_cache();
  std::lock_guard<std::mutex> lk(g_devs_mu);
  g_devs.clear();
  g_devs_set = false;  // the next call reads MLHIP_DEVICES again
  g_devs_bad = false;
  return 0;
}

int mlhip_msm_multi(int curve, int group, const int* devices, int n_devices, const void* points, const void* scalars,
                    int scalars_mont, size_t n, int window_c, void* out_affine) {
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (group != MLHIP_GROUP_G1 && group != MLHIP_GROUP_G2) return mlhip_rt::fail(MLHIP_EINVAL, "group must be 1 (G1) or 2 (G2)");
  if (!out_affine) return mlhip_rt::fail(MLHIP_EINVAL, "null output pointer");
  if (n_devices < 1 || n_devices > 64 || !devices) return mlhip_rt::fail(MLHIP_EINVAL, "device list: 1 .. 64 entries");
  const size_t ptsz = group == MLHIP_GROUP_G1 ? sz.g1 : sz.g2;
  if (n == 0) {
    memset(out_affine, 0, ptsz);
    return 0;
  }
  if (!points || !scalars) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  std::vector<int> devs(devices, devices + n_devices);
  for (int d : devs)
    if (d < 0 || d >= MLHIP_MAX_DEVICES) return mlhip_rt::fail(MLHIP_EINVAL, "device list: index out of range (0 .. 63)");
  if (devs.size() > n) devs.resize(n);
  return msm_multi(devs, curve, group, points, scalars, scalars_mont, n, window_c, out_affine, ptsz);
}

int mlhip_sizes(int curve, size_t* fp, size_t* g1, size_t* g2, size_t* gt) {
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (fp) *fp = sz.fp;
  if (g1) *g1 = sz.g1;
  if (g2) *g2 = sz.g2;
  if (gt) *gt = sz.gt;
  return 0;
}

// fold_tile != 0: a plan over shifted-base tables (msm_fold.h, mlhip_internal.h) -- window_c is the digit width, the
// 2^(c-1) buckets all digits share are cut into groups of at most 2^15 for the reduction; the table itself is built by
// mlhip_tu_plan_fold_build_* (mlhip_bases_create).
static int plan_create_ex(int curve, int group, size_t max_n, int window_c, size_t fold_tile, mlhip_msm_plan** out) {
  if (!out) return mlhip_rt::fail(MLHIP_EINVAL, "null plan pointer");
  *out = nullptr;
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (group != MLHIP_GROUP_G1 && group != MLHIP_GROUP_G2)
    return mlhip_rt::fail(MLHIP_EINVAL, "group must be 1 (G1) or 2 (G2)");
  if (max_n == 0 || max_n > ((size_t)1 << 27)) return mlhip_rt::fail(MLHIP_EINVAL, "max_n out of range (1 .. 2^27)");
  if (window_c == 0) window_c = pick_window(max_n, sz.fr_bits);
  if (window_c < 4 || window_c > 20) return mlhip_rt::fail(MLHIP_EINVAL, "window_c out of range (4 .. 20)");
  const int digits = msm_num_windows(sz.fr_bits, window_c);
  // sorted-entry offsets, cursors and scans are 32-bit: W * max_n entries must be addressable
  if (!fold_tile && (size_t)digits * max_n > 0xFFFFFFFFull)
    return mlhip_rt::fail(MLHIP_EINVAL, "window_c too small for max_n: W * max_n entries exceed 2^32 - 1");
  if (fold_tile) {
    // an entry = table row index (below Wd fold_tile) | sign: 31 bits + 1; the tiles of one MSM are its segments
    if ((size_t)digits * fold_tile > ((size_t)1 << 30)) return mlhip_rt::fail(MLHIP_EINVAL, "shifted-base tables: tile too long");
    // the sort's per-block bin counts are 16-bit and a block of 1024 scalars may put all its 1024 Wd entries into one bin
    if ((size_t)digits * 1024 >= 65536) return mlhip_rt::fail(MLHIP_EINVAL, "shifted-base tables: digit width below 5 bits");
    if ((max_n + fold_tile - 1) / fold_tile > MLHIP_MAX_SEGMENTS) return mlhip_rt::fail(MLHIP_EINVAL, "shifted-base tables: too many tiles");
  }
  int rc = ensure_device();
  if (rc) return rc;
  mlhip_msm_plan* p = new mlhip_msm_plan();
  p->curve = curve;
  p->group = group;
  p->device = g_device;
  p->c = window_c;
  p->Wd = digits;
  p->max_n = max_n;
  if (fold_tile) {
    p->fold = 1;
    p->fold_tile = fold_tile;
    const uint32_t nbuckets = 1u << (window_c - 1);
    p->M = nbuckets < 32768u ? nbuckets : 32768u;
    p->W = (int)(nbuckets / p->M);
  } else {
    p->W = digits;
    p->M = 1u << (window_c - 1);
  }
  // buckets per level-1 reduction chunk: 16 for G1 (the quad-lane kernels are bound by work, and a longer chunk
  // halves the second level), 8 for G2 on the boundary-form curves and for tiny windows
  // (and for small bucket sets, where the chunk pass is a dependent chain rather than work: 2^12 points, c = 13:
  // reduction 0.25 -> 0.22 ms)
  // G2 (carry-free lane-pair reduction): 16 as well -- BLS12-381: reduction 1.43 -> 1.31 ms at c = 16
  const bool chunks16 = true;
  p->lgL = (chunks16 && p->M >= 256 && (size_t)p->W * p->M >= ((size_t)1 << 17)) ? 4 : 3;
  if (const char* e = getenv("MLHIP_CHUNK_LOG2")) {
    int v = atoi(e);
    if (v >= 1 && v <= 6 && (1u << v) <= p->M) p->lgL = v;
  }
  p->L = 1 << p->lgL;
  p->T = p->M / p->L;
  p->nb = ilog2(p->T);
  p->nsel = 4 + p->nb;  // two half-sums of W0, two of A, nb bit-masked sums
  switch (curve) {
    case MLHIP_CURVE_BN254: rc = mlhip_tu_plan_alloc_Bn254(p); break;
    case MLHIP_CURVE_BLS12_381: rc = mlhip_tu_plan_alloc_Bls381(p); break;
    default: rc = mlhip_tu_plan_alloc_Bls377(p); break;
  }
  if (!rc && p->fold && (p->sort_low <= 0 || !p->reduce28))
    rc = mlhip_rt::fail(MLHIP_EINVAL, "shifted-base tables need the two-level sort and the carry-free kernels");
  if (rc) {
    mlhip_msm_plan_destroy(p);
    return rc;
  }
  *out = p;
  return 0;
}

int mlhip_msm_plan_create(int curve, int group, size_t max_n, int window_c, mlhip_msm_plan** out) {
  return plan_create_ex(curve, group, max_n, window_c, 0, out);
}

int mlhip_msm_plan_destroy(mlhip_msm_plan* p) {
  if (!p) return 0;
  (void)hipSetDevice(p->device);
  void* ptrs[] = {p->d_digits, p->d_sorted, p->d_zero, p->d_offsets, p->d_biglist, p->d_buckets, p->d_A, p->d_W0, p->d_out,
                  p->d_order, p->d_hist, p->d_tilesums, p->d_coarse_off, p->d_points28, p->d_blockhist, p->d_state28, p->d_bigprefix, p->d_bigpart, p->d_binprefix};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  if (p->h_out) (void)hipHostFree(p->h_out);
  for (int i = 0; i < 5; i++)
    if (p->ev[i]) (void)hipEventDestroy(p->ev[i]);
  if (p->done) (void)hipEventDestroy(p->done);
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  if (p->ev_join) (void)hipEventDestroy(p->ev_join);
  for (hipEvent_t e : p->ev_seg)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->ev_seg_sc)
    if (e) (void)hipEventDestroy(e);
  for (auto& tile : p->ev_tile)
    for (hipEvent_t e : tile)
      if (e) (void)hipEventDestroy(e);
  if (p->aux) (void)hipStreamDestroy(p->aux);
  for (int i = 0; i < 2; i++) {
    if (p->sort_helper[i]) (void)mlhip_msm_plan_destroy(p->sort_helper[i]);
    if (p->ev_sorted[i]) (void)hipEventDestroy(p->ev_sorted[i]);
    if (p->ev_lists_free[i]) (void)hipEventDestroy(p->ev_lists_free[i]);
  }
  if (p->sort_stream) (void)hipStreamDestroy(p->sort_stream);
  delete p;
  return 0;
}

int mlhip_msm_launch(mlhip_msm_plan* p, const void* d_points, const void* d_scalars, int scalars_mont, size_t n,
                     void* stream) {
  if (!p) return mlhip_rt::fail(MLHIP_EINVAL, "null plan");
  if (p->pending) return mlhip_rt::fail(MLHIP_EINVAL, "plan already has a pending launch; call mlhip_msm_finish first");
  if (n > p->max_n) return mlhip_rt::fail(MLHIP_EINVAL, "n exceeds the plan's max_n");
  if (n && (!d_points || !d_scalars)) return mlhip_rt::fail(MLHIP_EINVAL, "null device pointer");
  HIPCHK(hipSetDevice(p->device));
  hipStream_t st = (hipStream_t)stream;
  int rc;
  switch (p->curve) {
    case MLHIP_CURVE_BN254: rc = mlhip_tu_plan_launch_Bn254(p, d_points, d_scalars, scalars_mont, n, st); break;
    case MLHIP_CURVE_BLS12_381: rc = mlhip_tu_plan_launch_Bls381(p, d_points, d_scalars, scalars_mont, n, st); break;
    default: rc = mlhip_tu_plan_launch_Bls377(p, d_points, d_scalars, scalars_mont, n, st); break;
  }
  if (rc) {
    // a failure part-way through the launch train: `done` may never have been recorded, so a later finish must not
    // read h_out.  Drain what was queued (the error text survives: the drain calls do not go through fail()) and leave
    // the plan reusable, as tu_plan_stream does.
    (void)hipStreamSynchronize(st);
    if (p->aux) (void)hipStreamSynchronize(p->aux);
    if (p->sort_stream) (void)hipStreamSynchronize(p->sort_stream);
    (void)hipGetLastError();
    p->pending = false;
    p->upload_src = nullptr;
  }
  return rc;
}

// can this plan run the segment train (plan_stream / plan_stream_shared)?  The condition stream_begin checks.
static bool plan_can_stream(const mlhip_msm_plan* p) {
  return p->aux && p->d_points28;
}

// the shared-scalar train on two plans; h_* = nullptr: everything is already at the d_* pointers
static int tu_plan_shared(mlhip_msm_plan* g1, mlhip_msm_plan* g2, void* d1, void* d2, void* dsc, const void* h1, const void* h2,
                          const void* hsc, int mont, size_t n, hipStream_t st) {
  int rc;
  switch (g1->curve) {
    case MLHIP_CURVE_BN254: rc = mlhip_tu_plan_shared_Bn254(g1, g2, d1, d2, dsc, h1, h2, hsc, mont, n, st); break;
    case MLHIP_CURVE_BLS12_381: rc = mlhip_tu_plan_shared_Bls381(g1, g2, d1, d2, dsc, h1, h2, hsc, mont, n, st); break;
    default: rc = mlhip_tu_plan_shared_Bls377(g1, g2, d1, d2, dsc, h1, h2, hsc, mont, n, st); break;
  }
  if (rc) {  // as mlhip_msm_launch: drain what was queued (copies from the caller's buffers too) and leave both plans reusable
    (void)hipStreamSynchronize(st);
    if (g1->aux) (void)hipStreamSynchronize(g1->aux);
    if (g1->sort_stream) (void)hipStreamSynchronize(g1->sort_stream);
    if (g2->aux) (void)hipStreamSynchronize(g2->aux);
    (void)hipGetLastError();
    g1->pending = g2->pending = false;
  }
  return rc;
}

int mlhip_msm_launch_shared(mlhip_msm_plan* g1, mlhip_msm_plan* g2, const void* d_points_g1, const void* d_points_g2,
                            const void* d_scalars, int scalars_mont, size_t n, void* stream) {
  if (!g1 || !g2) return mlhip_rt::fail(MLHIP_EINVAL, "null plan");
  if (g1->group != MLHIP_GROUP_G1 || g2->group != MLHIP_GROUP_G2 || g1->curve != g2->curve || g1->device != g2->device)
    return mlhip_rt::fail(MLHIP_EINVAL, "shared-scalar MSM needs a G1 plan and a G2 plan of one curve on one device");
  if (g1->pending || g2->pending)
    return mlhip_rt::fail(MLHIP_EINVAL, "plan already has a pending launch; call mlhip_msm_finish first");
  if (n > g1->max_n || n > g2->max_n) return mlhip_rt::fail(MLHIP_EINVAL, "n exceeds a plan's max_n");
  if (n && (!d_points_g1 || !d_points_g2 || !d_scalars)) return mlhip_rt::fail(MLHIP_EINVAL, "null device pointer");
  const bool share = n != 0 && g1->c == g2->c && plan_can_stream(g1) && plan_can_stream(g2);
  if (!share) {  // nothing to share (or a plan on a second-implementation path): two ordinary launches, one after the other
    int rc = mlhip_msm_launch(g1, d_points_g1, d_scalars, scalars_mont, n, stream);
    if (rc) return rc;
    rc = mlhip_msm_launch(g2, d_points_g2, d_scalars, scalars_mont, n, stream);
    if (rc) {  // leave neither plan pending: the caller gets one error for the pair
      (void)hipStreamSynchronize((hipStream_t)stream);
      g1->pending = false;
    }
    return rc;
  }
  HIPCHK(hipSetDevice(g1->device));
  hipStream_t st = (hipStream_t)stream;
  void *p1 = const_cast<void*>(d_points_g1), *p2 = const_cast<void*>(d_points_g2), *sc = const_cast<void*>(d_scalars);
  return tu_plan_shared(g1, g2, p1, p2, sc, nullptr, nullptr, nullptr, scalars_mont, n, st);
}

int mlhip_msm_g1g2(int curve, const void* points_g1, const void* points_g2, const void* scalars, int scalars_mont, size_t n,
                   int window_c, void* out_g1, void* out_g2) {
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (!out_g1 || !out_g2) return mlhip_rt::fail(MLHIP_EINVAL, "null output pointer");
  if (n == 0) {  // the points at infinity, as for mlhip_msm_g1 / _g2
    memset(out_g1, 0, sz.g1);
    memset(out_g2, 0, sz.g2);
    return 0;
  }
  if (!points_g1 || !points_g2 || !scalars) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  if (!spread_devices(n, false).empty()) {
    // spread over the device list: every device sorts its own shard anyway -- two sharded MSMs; the caller's window_c
    // travels as given (0 = each shard picks the width of its own size)
    int rc = msm_host_buffers(curve, MLHIP_GROUP_G1, points_g1, scalars, scalars_mont, n, window_c, out_g1);
    if (rc) return rc;
    return msm_host_buffers(curve, MLHIP_GROUP_G2, points_g2, scalars, scalars_mont, n, window_c, out_g2);
  }
  if (window_c == 0) window_c = pick_window(n, sz.fr_bits);
  int rc = ensure_device();
  if (rc) return rc;
  PoolEntry* e1 = pool_acquire(curve, MLHIP_GROUP_G1, window_c, n, sz.g1, rc);
  if (!e1) return rc;
  PoolEntry* e2 = pool_acquire(curve, MLHIP_GROUP_G2, window_c, n, sz.g2, rc);
  if (!e2) {
    pool_release(e1, false);
    return rc;
  }
  if (plan_can_stream(e1->plan) && plan_can_stream(e2->plan)) {
    rc = tu_plan_shared(e1->plan, e2->plan, e1->d_pts, e2->d_pts, e1->d_sc, points_g1, points_g2, scalars, scalars_mont, n,
                        e1->stream);
    if (!rc) rc = mlhip_msm_finish(e1->plan, out_g1, nullptr);
    if (!rc) rc = mlhip_msm_finish(e2->plan, out_g2, nullptr);
    if (rc) {  // whatever is still queued reads the caller's buffers: let it drain, leave the plans reusable
      (void)hipDeviceSynchronize();
      e1->plan->pending = e2->plan->pending = false;
    }
    pool_release(e2, rc != 0);
    pool_release(e1, rc != 0);
    return rc;
  }
  // a second-implementation path (MLHIP_ACC32=1 ...): nothing to share
  pool_release(e2, false);
  pool_release(e1, false);
  rc = msm_host_buffers(curve, MLHIP_GROUP_G1, points_g1, scalars, scalars_mont, n, window_c, out_g1);
  if (rc) return rc;
  return msm_host_buffers(curve, MLHIP_GROUP_G2, points_g2, scalars, scalars_mont, n, window_c, out_g2);
}

int mlhip_msm_finish(mlhip_msm_plan* p, void* out_affine, void* out_xyzz) {
  if (!p || !out_affine) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  HIPCHK(hipSetDevice(p->device));
  switch (p->curve) {
    case MLHIP_CURVE_BN254: return mlhip_tu_plan_finish_Bn254(p, out_affine, out_xyzz);
    case MLHIP_CURVE_BLS12_381: return mlhip_tu_plan_finish_Bls381(p, out_affine, out_xyzz);
    default: return mlhip_tu_plan_finish_Bls377(p, out_affine, out_xyzz);
  }
}

int mlhip_msm_run(mlhip_msm_plan* p, const void* d_points, const void* d_scalars, int scalars_mont, size_t n,
                  void* stream, void* out_affine, void* out_xyzz) {
  if (!p || !out_affine) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  int rc = mlhip_msm_launch(p, d_points, d_scalars, scalars_mont, n, stream);
  if (rc) return rc;
  return mlhip_msm_finish(p, out_affine, out_xyzz);
}

int mlhip_msm_plan_set_profiling(mlhip_msm_plan* p, int on) {
  if (!p) return mlhip_rt::fail(MLHIP_EINVAL, "null plan");
  p->profiling = on != 0;
  return 0;
}

// Room for the twisted Edwards form of the points (168-byte Niels triples instead of 112-byte rows) in a plan that may take
// that path; called when the SRS promise is made, with nothing in flight on the plan.  A failed allocation is not an error:
// the plan keeps (or gets back) the smaller buffer and stays on the Weierstrass kernels (plan_use_edwards checks the size).
static void plan_reserve_edwards(mlhip_msm_plan* p) {
  if (!p->points28_elem_ed || p->points28_elem >= p->points28_elem_ed || !p->d_points28 || p->fold) return;
  const char* e = getenv("MLHIP_EDWARDS");
  if (e && e[0] == '0') return;
  (void)hipSetDevice(p->device);
  if (p->aux) (void)hipStreamSynchronize(p->aux);
  void* bigger = nullptr;
  if (hipMalloc(&bigger, p->max_n * p->points28_elem_ed) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  (void)hipFree(p->d_points28);
  p->d_points28 = bigger;
  p->points28_elem = p->points28_elem_ed;
  p->conv_src = nullptr;
}

int mlhip_msm_plan_assume_srs(mlhip_msm_plan* p, int on) {
  if (!p) return mlhip_rt::fail(MLHIP_EINVAL, "null plan");
  if (p->pending) return mlhip_rt::fail(MLHIP_EINVAL, "mlhip_msm_plan_assume_srs with a launch pending");
  if (p->fold) return mlhip_rt::fail(MLHIP_EINVAL, "mlhip_msm_plan_assume_srs: this plan reads the tables of a mlhip_bases handle");
  p->conv_src = nullptr;  // whatever carry-free copy the plan holds was made under the other promise
  p->points_static = p->trust_subgroup = on != 0;
  if (on) plan_reserve_edwards(p);
  return 0;
}

int mlhip_msm_plan_timings(mlhip_msm_plan* p, float* ms, int cap) {
  if (!p || !ms) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  int k = cap < 11 ? cap : 11;
  for (int i = 0; i < k && i < 6; i++) ms[i] = p->ms[i];
  if (k >= 7) ms[6] = p->tiles_timed > 0 ? (float)p->tiles_timed : 1.0f;
  if (k >= 8) ms[7] = (float)p->c;
  if (k >= 9) ms[8] = (float)p->Wd;
  if (k >= 10) ms[9] = p->last_ed ? 1.0f : 0.0f;
  if (k >= 11) ms[10] = p->fold ? 1.0f : 0.0f;
  return k;
}

struct mlhip_bases {
  mlhip_msm_plan* plan = nullptr;
  void *d_pts = nullptr, *d_sc = nullptr;
  size_t n = 0, ptsz = 0;
  int curve = 0, group = 0, device = 0;
  hipStream_t stream = nullptr;  // own non-blocking stream (see PoolEntry)
  std::mutex mu;  // one MSM at a time per handle: the plan and the scalar buffer are shared state
  // a table spread over several devices: contiguous shards, shard r = bases [lo[r], lo[r + 1]) on devs[r]
  std::vector<mlhip_bases*> shards;
  std::vector<size_t> lo;
  std::vector<int> devs;
};

int mlhip_bases_destroy(mlhip_bases* b) {
  if (!b) return 0;
  for (mlhip_bases* sh : b->shards) mlhip_bases_destroy(sh);
  if (b->shards.empty()) {
    (void)hipSetDevice(b->device);
    if (b->d_pts) (void)hipFree(b->d_pts);
    if (b->d_sc) (void)hipFree(b->d_sc);
    if (b->plan) mlhip_msm_plan_destroy(b->plan);
    if (b->stream) (void)hipStreamDestroy(b->stream);
  }
  delete b;
  return 0;
}

static int tu_plan_fold_build(mlhip_msm_plan* p, const void* d_pts, size_t n, hipStream_t st) {
  switch (p->curve) {
    case MLHIP_CURVE_BN254: return mlhip_tu_plan_fold_build_Bn254(p, d_pts, n, st);
    case MLHIP_CURVE_BLS12_381: return mlhip_tu_plan_fold_build_Bls381(p, d_pts, n, st);
    default: return mlhip_tu_plan_fold_build_Bls377(p, d_pts, n, st);
  }
}

// Shifted-base tables for a table of n resident bases (msm_fold.h)?  They cost Wd rows per base (112 B a row for a 48-byte
// field: 1.5 GB for 2^20 BLS12-381 G1 bases) and ~60 ms per 2^20 bases to build, and pay from the first few MSMs on.
//   MLHIP_BASES_TABLES = 0: never; = 1: always (any size: what the tests use); unset: for tables of at least 2^10 bases
//   created with window_c = 0 (an explicit window width asks for that Pippenger geometry) that fit a quarter of the free memory.
//   MLHIP_FOLD_WINDOW = c: the digit width (default by size, see below); MLHIP_FOLD_TILE_LOG2 = t: tiles of 2^t bases (20).
static bool bases_want_tables(int group, size_t n, int window_c, int fr_bits, size_t ptsz, int* c_out, size_t* tile_out) {
  const char* e = getenv("MLHIP_BASES_TABLES");
  const bool forced = e && e[0] == '1';
  if (e && e[0] == '0') return false;
  // (G2 was measured from 2^20 bases down to 2^17 only: its small tables stay plain)
  if (!forced && (n < ((size_t)1 << (group == MLHIP_GROUP_G1 ? 10 : 17)) || window_c != 0)) return false;
  int lg_tile = 20;
  if (const char* t = getenv("MLHIP_FOLD_TILE_LOG2")) {
    const int v = atoi(t);
    if (v >= 4 && v <= 24) lg_tile = v;
  }
  size_t tile = (size_t)1 << lg_tile;
  if (n < tile) tile = n;
  int c = 0;
  if (const char* w = getenv("MLHIP_FOLD_WINDOW")) c = atoi(w);
  // 20 bits (13 digits for a 253-255-bit group order) at every size from 2^16 on: narrower digits mean more of them and, in
  // the even digit layout, most of the 2^(c-1) buckets half-used -- same-box runs (profiles/r04_fold.txt), BLS12-381 G1,
  // resident scalars, c = 18 / 19 / 20 against the plain table: 2^17 1.11 / 0.83 / 0.80 (0.90) ms, 2^18 1.68 / 1.10 / 1.02
  // (1.19), 2^19 - / 1.61 / 1.50 (1.75), 2^20 5.15 / 3.10 / 2.75 (3.16).  Below 2^16 bases an MSM is latency, not work -- the
  // reduction's dependent chains grow with the bucket count, the accumulation's with the entries per bucket -- and what the
  // tables save is mostly the host tail's 256 doublings (0.14 ms): 13 / 14 / 16 bits from 2^10 / 2^12 / 2^13 bases:
  // 2^10 0.35 (plain 0.49) ms, 2^11 0.39 (0.52), 2^12 0.46 (0.56), 2^14 0.57 (0.67), 2^15 0.70 (0.72), 2^16 at 20 bits 0.75 (0.81)
  if (c < 5 || c > 20) c = n < ((size_t)1 << 12) ? 13 : n < ((size_t)1 << 13) ? 14 : n < ((size_t)1 << 16) ? 16 : 20;
  const size_t tiles = (n + tile - 1) / tile;
  const size_t rows = tiles * (size_t)msm_num_windows(fr_bits, c) * tile;
  if (!forced) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    // carry-free rows: at most 5/4 of the boundary form's bytes a coordinate (10 x 4 B for a 32-byte field, 14 x 4 B for a 48-byte
    // one); a G1 row may be a Niels triple (three coordinates instead of two); + one tile of boundary-form rows during the build
    const size_t row_bytes = group == MLHIP_GROUP_G1 ? ptsz / 2 * 3 * 5 / 4 : ptsz * 5 / 4;
    if (rows * row_bytes + (size_t)msm_num_windows(fr_bits, c) * tile * ptsz > free_b / 4) return false;
  }
  *c_out = c;
  *tile_out = tile;
  return true;
}

static int bases_create_single(int curve, int group, const void* points, size_t n, int window_c, size_t ptsz,
                               mlhip_bases** out, bool points_on_device = false) {
  int rc = ensure_device();
  if (rc) return rc;
  mlhip_bases* b = new mlhip_bases();
  b->device = g_device;
  b->curve = curve;
  b->group = group;
  b->n = n;
  b->ptsz = ptsz;
  {
    Sizes sz;
    int fold_c = 0;
    size_t fold_tile = 0;
    if (curve_sizes(curve, sz) && bases_want_tables(group, n, window_c, sz.fr_bits, ptsz, &fold_c, &fold_tile)) {
      if (plan_create_ex(curve, group, n, fold_c, fold_tile, &b->plan) != 0) b->plan = nullptr;  // (the plain plan below)
    }
  }
  rc = b->plan ? 0 : mlhip_msm_plan_create(curve, group, n, window_c, &b->plan);
  if (!rc && (hipMalloc(&b->d_pts, n * b->ptsz) != hipSuccess || hipMalloc(&b->d_sc, n * 32) != hipSuccess))
    rc = mlhip_rt::fail(MLHIP_ENOMEM, "hipMalloc of the bases failed");
  if (!rc && hipMemcpy(b->d_pts, points, n * b->ptsz, points_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) != hipSuccess)
    rc = mlhip_rt::fail(MLHIP_EHIP, "upload of the bases failed");
  if (!rc && hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess)
    rc = mlhip_rt::fail(MLHIP_EHIP, "hipStreamCreate failed");
  if (rc) {
    mlhip_bases_destroy(b);
    return rc;
  }
  b->plan->points_static = true;  // the buffer is ours and never rewritten: convert it on the first MSM only
  // BLS12-377 G1: a table whose every point is in the prime-order subgroup (what an SRS is; checked here, once, on the
  // device: on the curve and phi(P) = [-x^2]P) has its buckets summed in twisted Edwards coordinates (ed28.h)
  if (curve == MLHIP_CURVE_BLS12_377 && group == MLHIP_GROUP_G1) {
    const char* e = getenv("MLHIP_EDWARDS");
    if (!(e && e[0] == '0')) {
      uint32_t* d_bad = nullptr;
      uint32_t bad = 1;
      if (hipMalloc(&d_bad, 4) == hipSuccess) {
        if (hipMemsetAsync(d_bad, 0, 4, b->stream) == hipSuccess &&
            mlhip_tu_g1_count_outside_subgroup_Bls377(b->d_pts, n, d_bad, b->stream) == 0 &&
            hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, b->stream) == hipSuccess &&
            hipStreamSynchronize(b->stream) == hipSuccess) {
          b->plan->trust_subgroup = bad == 0;
          if (bad == 0) plan_reserve_edwards(b->plan);
        }
        (void)hipFree(d_bad);
      }
    }
  }
  if (b->plan->fold) {
    // the shifted-base table, in the form the plan will read (after the subgroup check: Niels triples or Weierstrass rows).
    // If it cannot be built (memory), the handle falls back to a plain plan over the uploaded bases.
    rc = tu_plan_fold_build(b->plan, b->d_pts, n, b->stream);
    if (!rc && hipStreamSynchronize(b->stream) != hipSuccess) rc = MLHIP_EHIP;
    if (rc) {
      (void)hipGetLastError();
      const bool trusted = b->plan->trust_subgroup;
      mlhip_msm_plan_destroy(b->plan);
      b->plan = nullptr;
      rc = mlhip_msm_plan_create(curve, group, n, window_c, &b->plan);
      if (rc) {
        mlhip_bases_destroy(b);
        return rc;
      }
      b->plan->points_static = true;
      b->plan->trust_subgroup = trusted;
      if (trusted) plan_reserve_edwards(b->plan);
    }
  }
  *out = b;
  return 0;
}

static int bases_create_on(const std::vector<int>& devs, int curve, int group, const void* points, size_t n, int window_c,
                           size_t ptsz, mlhip_bases** out) {
  mlhip_bases* b = new mlhip_bases();
  b->curve = curve;
  b->group = group;
  b->n = n;
  b->ptsz = ptsz;
  b->devs = devs;
  b->shards.assign(devs.size(), nullptr);
  b->lo.assign(devs.size() + 1, n);
  int rc = run_on_devices(devs, n, [&](size_t r, size_t lo, size_t hi) {
    b->lo[r] = lo;
    return bases_create_single(curve, group, (const char*)points + lo * ptsz, hi - lo, window_c, ptsz, &b->shards[r]);
  });
  if (rc) {
    std::string msg = g_err;
    mlhip_bases_destroy(b);
    return mlhip_rt::fail(rc, msg);
  }
  *out = b;
  return 0;
}

int mlhip_bases_create(int curve, int group, const void* points, size_t n, int window_c, mlhip_bases** out) {
  if (!out) return mlhip_rt::fail(MLHIP_EINVAL, "null output pointer");
  *out = nullptr;
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (group != MLHIP_GROUP_G1 && group != MLHIP_GROUP_G2) return mlhip_rt::fail(MLHIP_EINVAL, "group must be 1 (G1) or 2 (G2)");
  if (!points || n == 0) return mlhip_rt::fail(MLHIP_EINVAL, "bases need at least one point");
  const size_t ptsz = group == MLHIP_GROUP_G1 ? sz.g1 : sz.g2;
  const std::vector<int> devs = spread_devices(n, false);
  if (!devs.empty()) return bases_create_on(devs, curve, group, points, n, window_c, ptsz, out);
  return bases_create_single(curve, group, points, n, window_c, ptsz, out);
}

int mlhip_bases_create_device(int curve, int group, const void* d_points, size_t n, int window_c, mlhip_bases** out) {
  if (!out) return mlhip_rt::fail(MLHIP_EINVAL, "null output pointer");
  *out = nullptr;
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (group != MLHIP_GROUP_G1 && group != MLHIP_GROUP_G2) return mlhip_rt::fail(MLHIP_EINVAL, "group must be 1 (G1) or 2 (G2)");
  if (!d_points || n == 0) return mlhip_rt::fail(MLHIP_EINVAL, "bases need at least one point");
  return bases_create_single(curve, group, d_points, n, window_c, group == MLHIP_GROUP_G1 ? sz.g1 : sz.g2, out, true);
}

int mlhip_bases_create_multi(int curve, int group, const int* devices, int n_devices, const void* points, size_t n,
                             int window_c, mlhip_bases** out) {
  if (!out) return mlhip_rt::fail(MLHIP_EINVAL, "null output pointer");
  *out = nullptr;
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (group != MLHIP_GROUP_G1 && group != MLHIP_GROUP_G2) return mlhip_rt::fail(MLHIP_EINVAL, "group must be 1 (G1) or 2 (G2)");
  if (!points || n == 0) return mlhip_rt::fail(MLHIP_EINVAL, "bases need at least one point");
  if (n_devices < 1 || n_devices > 64 || !devices) return mlhip_rt::fail(MLHIP_EINVAL, "device list: 1 .. 64 entries");
  std::vector<int> devs(devices, devices + n_devices);
  for (int d : devs)
    if (d < 0 || d >= MLHIP_MAX_DEVICES) return mlhip_rt::fail(MLHIP_EINVAL, "device list: index out of range (0 .. 63)");
  if (devs.size() > n) devs.resize(n);
  return bases_create_on(devs, curve, group, points, n, window_c, group == MLHIP_GROUP_G1 ? sz.g1 : sz.g2, out);
}

int mlhip_bases_checked_subgroup(mlhip_bases* b) {
  if (!b) return 0;
  if (!b->shards.empty()) {
    for (mlhip_bases* s : b->shards)
      if (!s || !mlhip_bases_checked_subgroup(s)) return 0;
    return 1;
  }
  return b->plan && b->plan->trust_subgroup ? 1 : 0;
}

int mlhip_bases_msm(mlhip_bases* b, const void* scalars, int scalars_mont, size_t n, void* out_affine) {
  if (!b || !out_affine) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  if (n > b->n) return mlhip_rt::fail(MLHIP_EINVAL, "more scalars than resident bases");
  if (n == 0) {
    memset(out_affine, 0, b->ptsz);
    return 0;
  }
  if (!scalars) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  if (!b->shards.empty()) {
    // every device adds up its part of the first n bases; the partial sums meet on the host (see msm_multi)
    const size_t D = b->shards.size();
    std::vector<char> partial(D * b->ptsz, 0);
    int rc = run_on_devices(b->devs, D, [&](size_t r, size_t, size_t) {
      const size_t lo = b->lo[r], hi = std::min(b->lo[r + 1], n);
      if (lo >= hi) return 0;  // this shard's bases lie beyond the call's scalars: identity
      return mlhip_bases_msm(b->shards[r], (const char*)scalars + lo * 32, scalars_mont, hi - lo, &partial[r * b->ptsz]);
    });
    if (rc) return rc;
    return host_group_sum(b->curve, b->group, partial.data(), D, out_affine);
  }
  if (hipSetDevice(b->device) != hipSuccess) return mlhip_rt::fail(MLHIP_EHIP, "hipSetDevice failed");
  std::lock_guard<std::mutex> lk(b->mu);
  {
    // after the first MSM (which leaves the converted copy of the bases) large calls stream their scalars
    const mlhip_msm_plan* p = b->plan;
    const int segments = stream_segments(p->group, n, p);
    if (segments > 1 && p->points_static && p->conv_src == b->d_pts && n <= p->conv_n) {
      int rc = tu_plan_stream(b->plan, b->d_pts, b->d_sc, nullptr, scalars, scalars_mont, n, segments, b->stream);
      return rc ? rc : mlhip_msm_finish(b->plan, out_affine, nullptr);
    }
  }
  if (hipMemcpy(b->d_sc, scalars, n * 32, hipMemcpyHostToDevice) != hipSuccess)
    return mlhip_rt::fail(MLHIP_EHIP, "hipMemcpy of MSM scalars failed");
  return mlhip_msm_run(b->plan, b->d_pts, b->d_sc, scalars_mont, n, b->stream, out_affine, nullptr);
}

int mlhip_bases_msm_device(mlhip_bases* b, const void* d_scalars, int scalars_mont, size_t n, void* stream, void* out_affine) {
  if (!b || !out_affine) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  if (!b->shards.empty()) return mlhip_rt::fail(MLHIP_EINVAL, "mlhip_bases_msm_device: the handle is spread over several devices");
  if (n > b->n) return mlhip_rt::fail(MLHIP_EINVAL, "more scalars than resident bases");
  if (n == 0) {
    memset(out_affine, 0, b->ptsz);
    return 0;
  }
  if (!d_scalars) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  if (hipSetDevice(b->device) != hipSuccess) return mlhip_rt::fail(MLHIP_EHIP, "hipSetDevice failed");
  std::lock_guard<std::mutex> lk(b->mu);
  return mlhip_msm_run(b->plan, b->d_pts, d_scalars, scalars_mont, n, stream, out_affine, nullptr);
}

mlhip_msm_plan* mlhip_bases_plan(mlhip_bases* b) { return b && b->shards.empty() ? b->plan : nullptr; }

int mlhip_release_cache(void) {
  // the fixed-base tables of the batched scalar multiplication (one per curve and device; msm_scalar_mul.h)
  mlhip_tu_release_cache_Bn254();
  mlhip_tu_release_cache_Bls381();
  mlhip_tu_release_cache_Bls377();
  std::vector<PoolEntry*> idle;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size();)
      if (!g_pool[i]->busy) {
        idle.push_back(g_pool[i]);
        g_pool.erase(g_pool.begin() + i);
      } else {
        i++;
      }
  }
  for (PoolEntry* e : idle) pool_free_entry(e);
  // the idle leases (stream + scratch arena) of the other host-buffer entry points go too
  std::vector<Lease> leases;
  {
    std::lock_guard<std::mutex> lk(g_leases_mu);
    for (std::vector<Lease>& v : g_leases) {
      leases.insert(leases.end(), v.begin(), v.end());
      v.clear();
    }
  }
  for (Lease& l : leases) {
    if (l.arena) (void)hipFree(l.arena);
    if (l.st) (void)hipStreamDestroy(l.st);
  }
  return 0;
}

int mlhip_msm_g1(int curve, const void* points, const void* scalars, int scalars_mont, size_t n, int window_c,
                 void* out_affine) {
  return msm_host_buffers(curve, MLHIP_GROUP_G1, points, scalars, scalars_mont, n, window_c, out_affine);
}

int mlhip_msm_g2(int curve, const void* points, const void* scalars, int scalars_mont, size_t n, int window_c,
                 void* out_affine) {
  return msm_host_buffers(curve, MLHIP_GROUP_G2, points, scalars, scalars_mont, n, window_c, out_affine);
}

int mlhip_miller_loop(int curve, const void* g1, const void* g2, size_t ppp, size_t n_products, void* out_gt) {
  if (ppp < 1 || ppp > 4) return mlhip_rt::fail(MLHIP_EINVAL, "pairs_per_product must be 1..4");
  return pairing_host(curve, 0, g1, g2, ppp, n_products, nullptr, out_gt);
}

int mlhip_final_exp(int curve, const void* in_gt, size_t n, void* out_gt) {
  return pairing_host(curve, 1, nullptr, nullptr, 1, n, in_gt, out_gt);
}

int mlhip_pairing_batch(int curve, const void* g1, const void* g2, size_t n, void* out_gt) {
  return pairing_host(curve, 2, g1, g2, 1, n, nullptr, out_gt);
}

int mlhip_miller_loop_device(int curve, const void* d_g1, const void* d_g2, size_t ppp, size_t n_products,
                             void* d_out_gt, void* stream) {
  if (ppp < 1 || ppp > 4) return mlhip_rt::fail(MLHIP_EINVAL, "pairs_per_product must be 1..4");
  int rc = ensure_device();
  if (rc) return rc;
  return tu_pairing(curve, 0, d_g1, d_g2, ppp, n_products, nullptr, d_out_gt, (hipStream_t)stream);
}

int mlhip_final_exp_device(int curve, const void* d_in_gt, size_t n, void* d_out_gt, void* stream) {
  int rc = ensure_device();
  if (rc) return rc;
  return tu_pairing(curve, 1, nullptr, nullptr, 1, n, d_in_gt, d_out_gt, (hipStream_t)stream);
}

int mlhip_pairing_batch_device(int curve, const void* d_g1, const void* d_g2, size_t n, void* d_out_gt, void* stream) {
  int rc = ensure_device();
  if (rc) return rc;
  return tu_pairing(curve, 2, d_g1, d_g2, 1, n, nullptr, d_out_gt, (hipStream_t)stream);
}

int mlhip_gt_mul_device(int curve, const void* d_a, const void* d_b, size_t n, void* d_out, void* stream) {
  int rc = ensure_device();
  if (rc) return rc;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (curve) {
    case MLHIP_CURVE_BN254: return mlhip_tu_gt_mul_Bn254(d_a, d_b, n, d_out, st);
    case MLHIP_CURVE_BLS12_381: return mlhip_tu_gt_mul_Bls381(d_a, d_b, n, d_out, st);
    case MLHIP_CURVE_BLS12_377: return mlhip_tu_gt_mul_Bls377(d_a, d_b, n, d_out, st);
    default: return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  }
}

int mlhip_gt_mul(int curve, const void* a, const void* b, size_t n, void* out) {
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (n == 0) return 0;
  if (!a || !b || !out) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  int rc = ensure_device();
  if (rc) return rc;
  HostCall hc;
  hc.reserve(3 * n * sz.gt);
  void* da = hc.up(a, n * sz.gt);
  void* db = hc.up(b, n * sz.gt);
  void* dout = hc.dev(n * sz.gt);
  if (hc.rc) return hc.rc;
  rc = mlhip_gt_mul_device(curve, da, db, n, dout, hc.l.st);
  if (rc) return rc;
  return hc.down(out, dout, n * sz.gt);
}

int mlhip_scalar_mul_device(int curve, int group, const void* d_points, size_t point_stride, const void* d_scalars,
                            int mont, size_t n, void* d_out, void* stream) {
  if (group != MLHIP_GROUP_G1 && group != MLHIP_GROUP_G2) return mlhip_rt::fail(MLHIP_EINVAL, "group must be 1 or 2");
  if (point_stride > 1) return mlhip_rt::fail(MLHIP_EINVAL, "point_stride must be 0 or 1");
  int rc = ensure_device();
  if (rc) return rc;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (curve) {
    case MLHIP_CURVE_BN254: return mlhip_tu_scalar_mul_Bn254(group, d_points, point_stride, d_scalars, mont, n, d_out, st);
    case MLHIP_CURVE_BLS12_381: return mlhip_tu_scalar_mul_Bls381(group, d_points, point_stride, d_scalars, mont, n, d_out, st);
    case MLHIP_CURVE_BLS12_377: return mlhip_tu_scalar_mul_Bls377(group, d_points, point_stride, d_scalars, mont, n, d_out, st);
    default: return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  }
}

int mlhip_scalar_mul(int curve, int group, const void* points, size_t point_stride, const void* scalars, int mont,
                     size_t n, void* out) {
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (group != MLHIP_GROUP_G1 && group != MLHIP_GROUP_G2) return mlhip_rt::fail(MLHIP_EINVAL, "group must be 1 or 2");
  if (n == 0) return 0;
  if (!points || !scalars || !out) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  int rc = ensure_device();
  if (rc) return rc;
  const size_t ptsz = group == MLHIP_GROUP_G1 ? sz.g1 : sz.g2;
  const size_t npts = point_stride ? n : 1;
  HostCall hc;
  hc.reserve(npts * ptsz + n * 32 + n * ptsz);
  void* dp = hc.up(points, npts * ptsz);
  void* ds = hc.up(scalars, n * 32);
  void* dout = hc.dev(n * ptsz);
  if (hc.rc) return hc.rc;
  rc = mlhip_scalar_mul_device(curve, group, dp, point_stride, ds, mont, n, dout, hc.l.st);
  if (rc) return rc;
  return hc.down(out, dout, n * ptsz);
}

int mlhip_gt_exp_device(int curve, const void* d_in, const void* d_scalars, int mont, size_t n, void* d_out, void* stream) {
  int rc = ensure_device();
  if (rc) return rc;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (curve) {
    case MLHIP_CURVE_BN254: return mlhip_tu_gt_exp_Bn254(d_in, d_scalars, mont, n, d_out, st);
    case MLHIP_CURVE_BLS12_381: return mlhip_tu_gt_exp_Bls381(d_in, d_scalars, mont, n, d_out, st);
    case MLHIP_CURVE_BLS12_377: return mlhip_tu_gt_exp_Bls377(d_in, d_scalars, mont, n, d_out, st);
    default: return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  }
}

int mlhip_gt_exp(int curve, const void* in, const void* scalars, int mont, size_t n, void* out) {
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (n == 0) return 0;
  if (!in || !scalars || !out) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  int rc = ensure_device();
  if (rc) return rc;
  HostCall hc;
  hc.reserve(2 * n * sz.gt + n * 32);
  void* din = hc.up(in, n * sz.gt);
  void* ds = hc.up(scalars, n * 32);
  void* dout = hc.dev(n * sz.gt);
  if (hc.rc) return hc.rc;
  rc = mlhip_gt_exp_device(curve, din, ds, mont, n, dout, hc.l.st);
  if (rc) return rc;
  return hc.down(out, dout, n * sz.gt);
}

int mlhip_pairing_product(int curve, const void* g1, const void* g2, size_t n, void* out) {
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (!out) return mlhip_rt::fail(MLHIP_EINVAL, "null output pointer");
  if (n && (!g1 || !g2)) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  int rc = ensure_device();
  if (rc) return rc;
  // n == 0: the empty product, FExp(1) = 1; run it through the same kernels with one infinity pair
  const size_t m0 = n ? n : 1;
  HostCall hc;
  hc.reserve(m0 * (sz.g1 + sz.g2 + sz.gt));
  void *d1, *d2;
  if (n) {
    d1 = hc.up(g1, n * sz.g1);
    d2 = hc.up(g2, n * sz.g2);
  } else {
    d1 = hc.dev(sz.g1);
    d2 = hc.dev(sz.g2);
    if (!hc.rc && (hipMemsetAsync(d1, 0, sz.g1, hc.l.st) != hipSuccess || hipMemsetAsync(d2, 0, sz.g2, hc.l.st) != hipSuccess))
      hc.rc = mlhip_rt::fail(MLHIP_EHIP, "hipMemset failed");
  }
  void* dgt = hc.dev(m0 * sz.gt);
  if (hc.rc) return hc.rc;
  // One Miller loop per lane pair while the pairs do not fill the GPU (latency: 7 pairs take 21 ms this way, 45 ms
  // grouped -- measured); beyond 2^17 pairs four pairs share one accumulator's squarings (throughput).
  const size_t per = m0 >= ((size_t)1 << 17) ? 4 : 1;
  const size_t groups = m0 / per, rest = m0 % per;
  rc = tu_pairing(curve, 0, d1, d2, per, groups, nullptr, dgt, hc.l.st);
  if (!rc && rest)
    rc = tu_pairing(curve, 0, (const char*)d1 + per * groups * sz.g1, (const char*)d2 + per * groups * sz.g2, rest, 1,
                    nullptr, (char*)dgt + groups * sz.gt, hc.l.st);
  // tree product: fold the upper part onto the lower part until one value is left
  size_t m = groups + (rest ? 1 : 0);
  while (m > 1 && rc == 0) {
    size_t half = m / 2;
    rc = mlhip_gt_mul_device(curve, dgt, (const char*)dgt + (m - half) * sz.gt, half, dgt, hc.l.st);
    m -= half;
  }
  if (!rc) rc = tu_pairing(curve, 1, nullptr, nullptr, 1, 1, dgt, dgt, hc.l.st);
  if (rc) return rc;
  return hc.down(out, dgt, sz.gt);
}

static int tu_wire_codec(int curve, int group, int encode, const void* d_in, size_t n, int compressed, int subgroup,
                         void* d_out, void* d_status, hipStream_t st) {
  switch (curve) {
    case MLHIP_CURVE_BN254: return mlhip_tu_wire_codec_Bn254(group, encode, d_in, n, compressed, subgroup, d_out, d_status, st);
    case MLHIP_CURVE_BLS12_381: return mlhip_tu_wire_codec_Bls381(group, encode, d_in, n, compressed, subgroup, d_out, d_status, st);
    case MLHIP_CURVE_BLS12_377: return mlhip_tu_wire_codec_Bls377(group, encode, d_in, n, compressed, subgroup, d_out, d_status, st);
    default: return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  }
}

static int from_bytes_device(int curve, int group, const void* d_wire, size_t n, int compressed, int subgroup_check,
                             void* d_out, unsigned char* d_status, void* stream) {
  int rc = ensure_device();
  if (rc) return rc;
  return tu_wire_codec(curve, group, 0, d_wire, n, compressed ? 1 : 0, subgroup_check == 2 ? 2 : (subgroup_check ? 1 : 0), d_out, d_status,
                       (hipStream_t)stream);
}

static int to_bytes_device(int curve, int group, const void* d_affine, size_t n, int compressed, void* d_wire, void* stream) {
  int rc = ensure_device();
  if (rc) return rc;
  return tu_wire_codec(curve, group, 1, d_affine, n, compressed ? 1 : 0, 0, d_wire, nullptr, (hipStream_t)stream);
}

static int from_bytes_host(int curve, int group, const void* wire, size_t n, int compressed, int subgroup_check, void* out,
                           unsigned char* status) {
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (n == 0) return 0;
  if (!wire || !out || !status) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  int rc = ensure_device();
  if (rc) return rc;
  const size_t psz = group == 2 ? sz.g2 : sz.g1;
  const size_t wlen = compressed ? psz / 2 : psz;
  HostCall hc;
  hc.reserve(n * (wlen + psz + 1));
  void* dw = hc.up(wire, n * wlen);
  void* dout = hc.dev(n * psz);
  void* dst = hc.dev(n);
  if (hc.rc) return hc.rc;
  rc = from_bytes_device(curve, group, dw, n, compressed, subgroup_check, dout, (unsigned char*)dst, hc.l.st);
  if (rc) return rc;
  if (hipMemcpyAsync(status, dst, n, hipMemcpyDeviceToHost, hc.l.st) != hipSuccess)
    return mlhip_rt::fail(MLHIP_EHIP, "hipMemcpy D2H failed");
  return hc.down(out, dout, n * psz);
}

static int to_bytes_host(int curve, int group, const void* affine, size_t n, int compressed, void* wire) {
  Sizes sz;
  if (!curve_sizes(curve, sz)) return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  if (n == 0) return 0;
  if (!affine || !wire) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  int rc = ensure_device();
  if (rc) return rc;
  const size_t psz = group == 2 ? sz.g2 : sz.g1;
  const size_t wlen = compressed ? psz / 2 : psz;
  HostCall hc;
  hc.reserve(n * (psz + wlen));
  void* dp = hc.up(affine, n * psz);
  void* dw = hc.dev(n * wlen);
  if (hc.rc) return hc.rc;
  rc = to_bytes_device(curve, group, dp, n, compressed, dw, hc.l.st);
  if (rc) return rc;
  return hc.down(wire, dw, n * wlen);
}

int mlhip_g1_from_bytes_device(int curve, const void* d_wire, size_t n, int compressed, int subgroup_check, void* d_out,
                               unsigned char* d_status, void* stream) {
  return from_bytes_device(curve, 1, d_wire, n, compressed, subgroup_check, d_out, d_status, stream);
}
int mlhip_g2_from_bytes_device(int curve, const void* d_wire, size_t n, int compressed, int subgroup_check, void* d_out,
                               unsigned char* d_status, void* stream) {
  return from_bytes_device(curve, 2, d_wire, n, compressed, subgroup_check, d_out, d_status, stream);
}
int mlhip_g1_to_bytes_device(int curve, const void* d_affine, size_t n, int compressed, void* d_wire, void* stream) {
  return to_bytes_device(curve, 1, d_affine, n, compressed, d_wire, stream);
}
int mlhip_g2_to_bytes_device(int curve, const void* d_affine, size_t n, int compressed, void* d_wire, void* stream) {
  return to_bytes_device(curve, 2, d_affine, n, compressed, d_wire, stream);
}
int mlhip_g1_from_bytes(int curve, const void* wire, size_t n, int compressed, int subgroup_check, void* out, unsigned char* status) {
  return from_bytes_host(curve, 1, wire, n, compressed, subgroup_check, out, status);
}
int mlhip_g2_from_bytes(int curve, const void* wire, size_t n, int compressed, int subgroup_check, void* out, unsigned char* status) {
  return from_bytes_host(curve, 2, wire, n, compressed, subgroup_check, out, status);
}
int mlhip_g1_to_bytes(int curve, const void* affine, size_t n, int compressed, void* wire) {
  return to_bytes_host(curve, 1, affine, n, compressed, wire);
}
int mlhip_g2_to_bytes(int curve, const void* affine, size_t n, int compressed, void* wire) {
  return to_bytes_host(curve, 2, affine, n, compressed, wire);
}

int mlhip_g1_sum(int curve, const void* pts, size_t n, void* out) {
  if (!out || (n && !pts)) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  switch (curve) {
    case MLHIP_CURVE_BN254: return host_sum<FpField<Bn254>>(pts, n, out);
    case MLHIP_CURVE_BLS12_381: return host_sum<FpField<Bls381>>(pts, n, out);
    case MLHIP_CURVE_BLS12_377: return host_sum<FpField<Bls377>>(pts, n, out);
    default: return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  }
}

int mlhip_g2_sum(int curve, const void* pts, size_t n, void* out) {
  if (!out || (n && !pts)) return mlhip_rt::fail(MLHIP_EINVAL, "null pointer");
  switch (curve) {
    case MLHIP_CURVE_BN254: return host_sum<Fp2Field<Bn254>>(pts, n, out);
    case MLHIP_CURVE_BLS12_381: return host_sum<Fp2Field<Bls381>>(pts, n, out);
    case MLHIP_CURVE_BLS12_377: return host_sum<Fp2Field<Bls377>>(pts, n, out);
    default: return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  }
}

int mlhip_fp_mul_device(int curve, const void* d_a, const void* d_b, size_t n, int repeat, void* d_out, void* stream) {
  int rc = ensure_device();
  if (rc) return rc;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (curve) {
    case MLHIP_CURVE_BN254: return mlhip_tu_fp_mul_Bn254(d_a, d_b, n, repeat, d_out, st);
    case MLHIP_CURVE_BLS12_381: return mlhip_tu_fp_mul_Bls381(d_a, d_b, n, repeat, d_out, st);
    case MLHIP_CURVE_BLS12_377: return mlhip_tu_fp_mul_Bls377(d_a, d_b, n, repeat, d_out, st);
    default: return mlhip_rt::fail(MLHIP_EINVAL, "unknown curve id");
  }
}

}  // extern "C"
