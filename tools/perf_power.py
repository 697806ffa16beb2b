#!/usr/bin/env python3
"""Clock and power of the GPU while a kernel family runs back to back (rocm-smi polled from a second thread):
the pairing batch, the G1 MSM and the G2 MSM.  Answers "is the kernel power limited, and at which clock".
Usage: python tools/perf_power.py [seconds per family]"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mathlib_amd import _lib  # noqa: E402
from mathlib_amd.driver import Curve  # noqa: E402


def poll(stop, rows):
    while not stop.is_set():
        t = time.time()
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
        sclk = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
        pw = re.search(r"Socket Graphics Package Power \(W\):\s*([\d.]+)", out) or re.search(r"Power \(W\):\s*([\d.]+)", out)
        tj = re.search(r"Temperature \(Sensor junction\) \(C\):\s*([\d.]+)", out)
        rows.append((t, int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0, float(tj.group(1)) if tj else -1.0))
        time.sleep(0.05)


def run(name, fn, seconds):
    fn()
    torch.cuda.synchronize()
    stop, rows = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, rows))
    th.start()
    t0, k = time.time(), 0
    while time.time() - t0 < seconds:
        fn()
        k += 1
    torch.cuda.synchronize()
    dt = time.time() - t0
    stop.set()
    th.join()
    rows = rows[len(rows) // 4 :]  # the steady part
    f = [r[1] for r in rows if r[1] > 0]
    p = [r[2] for r in rows if r[2] > 0]
    tj = [r[3] for r in rows if r[3] > 0]
    avg = lambda v: sum(v) / len(v) if v else float("nan")  # noqa: E731
    print("%-44s %7.3f ms per call   sclk %6.0f MHz (min %d max %d)   power %6.0f W (max %.0f)   Tj %.0f C   [%d samples]" % (
        name, dt / k * 1e3, avg(f), min(f or [0]), max(f or [0]), avg(p), max(p or [0]), avg(tj), len(rows)), flush=True)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    lib = _lib.load()
    cid = _lib.CURVE_BLS12_381
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    rnd = lambda m: torch.randint(-(1 << 63), (1 << 63) - 1, (m, 4), dtype=torch.int64, generator=gen, device=dev).view(torch.uint8).reshape(m, 32).contiguous()  # noqa: E731
    cv = Curve(cid)
    n = 1 << 20
    pts = {}
    for group, raw, sz in ((1, cv.GenG1().raw, 96), (2, cv.GenG2().raw, 192)):
        base = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        pts[group] = torch.empty(n * sz, dtype=torch.uint8, device=dev)
        _lib.check(lib.mlhip_scalar_mul_device(cid, group, base.data_ptr(), 0, rnd(n).data_ptr(), 0, n, pts[group].data_ptr(), st))
    s = rnd(n)
    torch.cuda.synchronize()
    print(subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout.strip().splitlines()[-3:], flush=True)
    npair = 1 << 16
    gt = torch.empty(npair * 576, dtype=torch.uint8, device=dev)
    for env, label in ((None, "lane pairs"), ("1", "quads")):
        if env:
            os.environ["MLHIP_PAIRING_QUAD"] = env
        run("65 536 pairings, fused kernel (%s)" % label, lambda: (_lib.check(lib.mlhip_pairing_batch_device(cid, pts[1].data_ptr(), pts[2].data_ptr(), npair, gt.data_ptr(), st)), torch.cuda.synchronize()), seconds)
    os.environ.pop("MLHIP_PAIRING_QUAD", None)
    run("65 536 Miller loops", lambda: (_lib.check(lib.mlhip_miller_loop_device(cid, pts[1].data_ptr(), pts[2].data_ptr(), 1, npair, gt.data_ptr(), st)), torch.cuda.synchronize()), seconds)
    gt2 = torch.empty_like(gt)
    run("65 536 final exponentiations", lambda: (_lib.check(lib.mlhip_final_exp_device(cid, gt.data_ptr(), npair, gt2.data_ptr(), st)), torch.cuda.synchronize()), seconds)
    for group in (1, 2):
        plan = _lib.MsmPlan(cid, group, n, 16)
        run("G%d MSM 2^20, c = 16 (whole step)" % group, lambda: plan.run(pts[group].data_ptr(), s.data_ptr(), n, False, st), seconds)
        plan.close()


if __name__ == "__main__":
    main()
