"""ctypes binding of libmlhip.so (C ABI: include/mlhip.h).

The library is the product; this module only declares argument types.  There is no fallback of any
kind: if the shared object is missing, or a call fails (no GPU, HIP error), an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_float, c_int, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# MLHIP_LIB=<path> loads another build of the same library (A/B measurements of kernel variants on one box)
LIB_PATH = os.environ.get("MLHIP_LIB") or os.path.join(HERE, "libmlhip.so")

CURVE_BN254, CURVE_BLS12_381, CURVE_BLS12_377 = 0, 1, 2
GROUP_G1, GROUP_G2 = 1, 2
EINVAL = -1
ENODEVICE = -2

# every symbol include/mlhip.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = [
    "mlhip_version",
    "mlhip_last_error",
    "mlhip_device_count",
    "mlhip_set_device",
    "mlhip_init",
    "mlhip_get_devices",
    "mlhip_shutdown",
    "mlhip_msm_multi",
    "mlhip_bases_create_multi",
    "mlhip_sizes",
    "mlhip_msm_g1",
    "mlhip_msm_g2",
    "mlhip_miller_loop",
    "mlhip_final_exp",
    "mlhip_pairing_batch",
    "mlhip_gt_mul",
    "mlhip_gt_exp",
    "mlhip_pairing_product",
    "mlhip_msm_plan_create",
    "mlhip_msm_plan_destroy",
    "mlhip_msm_run",
    "mlhip_msm_launch",
    "mlhip_msm_launch_shared",
    "mlhip_msm_g1g2",
    "mlhip_msm_finish",
    "mlhip_msm_plan_set_profiling",
    "mlhip_msm_plan_assume_srs",
    "mlhip_msm_plan_timings",
    "mlhip_miller_loop_device",
    "mlhip_final_exp_device",
    "mlhip_pairing_batch_device",
    "mlhip_gt_mul_device",
    "mlhip_gt_exp_device",
    "mlhip_scalar_mul_device",
    "mlhip_scalar_mul",
    "mlhip_bases_create",
    "mlhip_bases_msm",
    "mlhip_bases_create_device",
    "mlhip_bases_msm_device",
    "mlhip_bases_plan",
    "mlhip_bases_checked_subgroup",
    "mlhip_bases_destroy",
    "mlhip_release_cache",
    "mlhip_g1_from_bytes",
    "mlhip_g1_to_bytes",
    "mlhip_g1_from_bytes_device",
    "mlhip_g1_to_bytes_device",
    "mlhip_g2_from_bytes",
    "mlhip_g2_to_bytes",
    "mlhip_g2_from_bytes_device",
    "mlhip_g2_to_bytes_device",
    "mlhip_g1_sum",
    "mlhip_g2_sum",
    "mlhip_fp_mul_device",
]


class MlhipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libmlhip error %d: %s" % (code, msg))
        self.code = code


_lib = None
ALT_PATH = os.path.join(HERE, "libmlhip_alt.so")  # the test build (python -m mathlib_amd.build --alt)
_alt = None
_use_alt = False


class _Active:
    """What load() hands out: the product library -- or, while a TEST has asked for a second implementation the product
    does not contain (tests/conftest.py: use_alt), the test build.  Attribute access goes to whichever is active, so the
    `lib` objects tests already hold follow the switch.  Nothing outside the tests ever calls use_alt."""

    def __getattr__(self, name):
        return getattr(_alt if (_use_alt and _alt is not None) else _lib, name)


_active = _Active()


def alt_available() -> bool:
    """Is an up-to-date test build next to the product library?  (build.py stamps both with the hash of the sources.)"""
    if LIB_PATH == ALT_PATH:
        return True
    try:
        from .build import source_hash

        with open(ALT_PATH + ".srchash") as f:
            return os.path.exists(ALT_PATH) and f.read().strip() == source_hash()
    except OSError:
        return False


def concrete():
    """The library that is active right now, as the ctypes object itself: what an object that owns a library handle (MsmPlan,
    driver.Bases) binds at creation, so that the handle is used and destroyed by the library that made it whatever is active
    later."""
    load()
    return _alt if (_use_alt and _alt is not None) else _lib


def use_alt(on: bool) -> None:
    """Tests only: route every call through the test build (True) or back to the product library (False)."""
    global _alt, _use_alt
    if on and _alt is None and LIB_PATH != ALT_PATH:
        load()
        _alt = _bind(ctypes.CDLL(ALT_PATH))
    _use_alt = bool(on)


def load():
    """Load libmlhip.so; raises if it has not been built (python -m mathlib_amd.build)."""
    global _lib
    if _lib is not None:
        return _active
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "mathlib_amd: %s is missing -- the HIP extension has not been built "
            "(run `python -m mathlib_amd.build`); there is no CPU fallback" % LIB_PATH
        )
    # PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7) but its libraries ask for
    # it by file name; if /opt/rocm's copy is mapped first the process ends up with two HIP runtimes
    # and the second one sees no GPU.  Import torch first so both share one runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    _lib = _bind(ctypes.CDLL(LIB_PATH))
    return _active


def _bind(lib: ctypes.CDLL) -> ctypes.CDLL:
    """argument and result types of every entry point"""
    vp, sz, ci = c_void_p, c_size_t, c_int
    lib.mlhip_version.restype = ci
    lib.mlhip_last_error.restype = c_char_p
    lib.mlhip_device_count.argtypes = [POINTER(ci)]
    lib.mlhip_set_device.argtypes = [ci]
    lib.mlhip_init.argtypes = [POINTER(ci), ci]
    lib.mlhip_get_devices.argtypes = [POINTER(ci), ci]
    lib.mlhip_shutdown.argtypes = []
    lib.mlhip_msm_multi.argtypes = [ci, ci, POINTER(ci), ci, vp, vp, ci, sz, ci, vp]
    lib.mlhip_bases_create_multi.argtypes = [ci, ci, POINTER(ci), ci, vp, sz, ci, ctypes.POINTER(c_void_p)]
    lib.mlhip_sizes.argtypes = [ci, POINTER(sz), POINTER(sz), POINTER(sz), POINTER(sz)]
    for f in (lib.mlhip_msm_g1, lib.mlhip_msm_g2):
        f.argtypes = [ci, vp, vp, ci, sz, ci, vp]
    lib.mlhip_msm_g1g2.argtypes = [ci, vp, vp, vp, ci, sz, ci, vp, vp]
    lib.mlhip_miller_loop.argtypes = [ci, vp, vp, sz, sz, vp]
    lib.mlhip_final_exp.argtypes = [ci, vp, sz, vp]
    lib.mlhip_pairing_batch.argtypes = [ci, vp, vp, sz, vp]
    lib.mlhip_gt_mul.argtypes = [ci, vp, vp, sz, vp]
    lib.mlhip_gt_exp.argtypes = [ci, vp, vp, ci, sz, vp]
    lib.mlhip_pairing_product.argtypes = [ci, vp, vp, sz, vp]
    lib.mlhip_gt_exp_device.argtypes = [ci, vp, vp, ci, sz, vp, vp]
    lib.mlhip_msm_plan_create.argtypes = [ci, ci, sz, ci, POINTER(vp)]
    lib.mlhip_msm_plan_destroy.argtypes = [vp]
    lib.mlhip_msm_run.argtypes = [vp, vp, vp, ci, sz, vp, vp, vp]
    lib.mlhip_msm_launch.argtypes = [vp, vp, vp, ci, sz, vp]
    lib.mlhip_msm_launch_shared.argtypes = [vp, vp, vp, vp, vp, ci, sz, vp]
    lib.mlhip_msm_finish.argtypes = [vp, vp, vp]
    lib.mlhip_msm_plan_set_profiling.argtypes = [vp, ci]
    lib.mlhip_msm_plan_assume_srs.argtypes = [vp, ci]
    lib.mlhip_msm_plan_timings.argtypes = [vp, POINTER(c_float), ci]
    lib.mlhip_miller_loop_device.argtypes = [ci, vp, vp, sz, sz, vp, vp]
    lib.mlhip_final_exp_device.argtypes = [ci, vp, sz, vp, vp]
    lib.mlhip_pairing_batch_device.argtypes = [ci, vp, vp, sz, vp, vp]
    lib.mlhip_gt_mul_device.argtypes = [ci, vp, vp, sz, vp, vp]
    lib.mlhip_scalar_mul_device.argtypes = [ci, ci, vp, sz, vp, ci, sz, vp, vp]
    lib.mlhip_scalar_mul.argtypes = [ci, ci, vp, sz, vp, ci, sz, vp]
    lib.mlhip_bases_create.argtypes = [ci, ci, vp, sz, ci, ctypes.POINTER(c_void_p)]
    lib.mlhip_bases_msm.argtypes = [vp, vp, ci, sz, vp]
    lib.mlhip_bases_create_device.argtypes = [ci, ci, vp, sz, ci, ctypes.POINTER(c_void_p)]
    lib.mlhip_bases_msm_device.argtypes = [vp, vp, ci, sz, vp, vp]
    lib.mlhip_bases_plan.argtypes = [vp]
    lib.mlhip_bases_plan.restype = c_void_p
    lib.mlhip_bases_destroy.argtypes = [vp]
    lib.mlhip_bases_checked_subgroup.argtypes = [vp]
    lib.mlhip_g1_from_bytes.argtypes = [ci, vp, sz, ci, ci, vp, vp]
    lib.mlhip_g1_to_bytes.argtypes = [ci, vp, sz, ci, vp]
    lib.mlhip_g1_from_bytes_device.argtypes = [ci, vp, sz, ci, ci, vp, vp, vp]
    lib.mlhip_g1_to_bytes_device.argtypes = [ci, vp, sz, ci, vp, vp]
    lib.mlhip_g2_from_bytes.argtypes = [ci, vp, sz, ci, ci, vp, vp]
    lib.mlhip_g2_to_bytes.argtypes = [ci, vp, sz, ci, vp]
    lib.mlhip_g2_from_bytes_device.argtypes = [ci, vp, sz, ci, ci, vp, vp, vp]
    lib.mlhip_g2_to_bytes_device.argtypes = [ci, vp, sz, ci, vp, vp]
    lib.mlhip_g1_sum.argtypes = [ci, vp, sz, vp]
    lib.mlhip_g2_sum.argtypes = [ci, vp, sz, vp]
    lib.mlhip_fp_mul_device.argtypes = [ci, vp, vp, sz, ci, vp, vp]
    for name in SYMBOLS:
        if name not in ("mlhip_last_error", "mlhip_bases_plan"):
            getattr(lib, name).restype = ci
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise MlhipError(rc, load().mlhip_last_error().decode(errors="replace"))


def sizes(curve: int):
    lib = load()
    a, b, c, d = c_size_t(), c_size_t(), c_size_t(), c_size_t()
    check(lib.mlhip_sizes(curve, byref(a), byref(b), byref(c), byref(d)))
    return a.value, b.value, c.value, d.value


def init_devices(devices=None) -> None:
    """the process's device list (mlhip_init): None / [] = every visible device"""
    devices = list(devices or [])
    arr = (c_int * max(1, len(devices)))(*devices)
    check(load().mlhip_init(arr, len(devices)))


def get_devices():
    buf = (c_int * 64)()
    k = load().mlhip_get_devices(buf, 64)
    if k < 0:  # MLHIP_DEVICES did not parse
        raise MlhipError(k, load().mlhip_last_error().decode())
    return [buf[i] for i in range(min(k, 64))]


def device_count() -> int:
    n = c_int()
    load().mlhip_device_count(byref(n))
    return n.value


def plan_timings(lib, handle):
    """mlhip_msm_plan_timings as a dict (include/mlhip.h): phase times in ms, tiles, and the two path flags"""
    buf = (c_float * 11)()
    k = lib.mlhip_msm_plan_timings(handle, buf, 11)
    names = ["digits", "sort", "accumulate", "reduce", "device_total", "host_tail", "tiles"]
    t = {names[i]: float(buf[i]) for i in range(min(k, 7))}
    if k >= 9:
        t["window_c"], t["digits_per_scalar"] = int(buf[7]), int(buf[8])
    if k >= 10:
        t["edwards"] = float(buf[9])
    if k >= 11:
        t["tables"] = float(buf[10])
    return t


class MsmPlan:
    """Device workspace for repeated MSMs over device-resident points/scalars (include/mlhip.h)."""

    def __init__(self, curve: int, group: int, max_n: int, window_c: int = 0):
        self._h = c_void_p()
        lib = self._lib = concrete()
        check(lib.mlhip_msm_plan_create(curve, group, max_n, window_c, byref(self._h)))
        _, g1, g2, _ = sizes(curve)
        self.point_bytes = g1 if group == GROUP_G1 else g2
        self.curve, self.group = curve, group

    def set_profiling(self, on: bool) -> None:
        check(self._lib.mlhip_msm_plan_set_profiling(self._h, 1 if on else 0))

    def assume_srs(self, on: bool) -> None:
        """the points are a fixed SRS: immutable at their address and in the prime-order subgroup (include/mlhip.h)"""
        check(self._lib.mlhip_msm_plan_assume_srs(self._h, 1 if on else 0))

    def timings(self):
        return plan_timings(self._lib, self._h)

    def window(self):
        """(c, W): the window width the plan runs with (the library's pick for window_c = 0) and its number of windows"""
        buf = (c_float * 9)()
        self._lib.mlhip_msm_plan_timings(self._h, buf, 9)
        return int(buf[7]), int(buf[8])

    def run(self, d_points: int, d_scalars: int, n: int, scalars_mont: bool, stream: int = 0, want_xyzz: bool = False):
        out = ctypes.create_string_buffer(self.point_bytes)
        xyzz = ctypes.create_string_buffer(2 * self.point_bytes) if want_xyzz else None
        check(
            self._lib.mlhip_msm_run(
                self._h, c_void_p(d_points), c_void_p(d_scalars), 1 if scalars_mont else 0, n, c_void_p(stream), out, xyzz
            )
        )
        return (out.raw, xyzz.raw) if want_xyzz else out.raw

    def launch(self, d_points: int, d_scalars: int, n: int, scalars_mont: bool, stream: int = 0) -> None:
        check(self._lib.mlhip_msm_launch(self._h, c_void_p(d_points), c_void_p(d_scalars), 1 if scalars_mont else 0, n, c_void_p(stream)))

    def launch_shared(self, g2_plan: "MsmPlan", d_points_g1: int, d_points_g2: int, d_scalars: int, n: int, scalars_mont: bool,
                      stream: int = 0) -> None:
        """self = the G1 plan: the G1 and the G2 MSM of one scalar vector, sorted once (finish both plans afterwards)"""
        check(self._lib.mlhip_msm_launch_shared(self._h, g2_plan._h, c_void_p(d_points_g1), c_void_p(d_points_g2), c_void_p(d_scalars),
                                             1 if scalars_mont else 0, n, c_void_p(stream)))

    def finish(self, want_xyzz: bool = False):
        out = ctypes.create_string_buffer(self.point_bytes)
        xyzz = ctypes.create_string_buffer(2 * self.point_bytes) if want_xyzz else None
        check(self._lib.mlhip_msm_finish(self._h, out, xyzz))
        return (out.raw, xyzz.raw) if want_xyzz else out.raw

    def close(self) -> None:
        if self._h:
            self._lib.mlhip_msm_plan_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
