#!/usr/bin/env python3
"""Static instruction histogram per function of a gfx950 assembly dump (hipcc --cuda-device-only -S).
Usage: isa_hist.py file.s [name-substring ...]   -- prints, for every function whose demangled name contains one of the
substrings (all functions when none is given), the instruction count per class, plus scratch / vgpr metadata of kernels."""
import collections
import re
import subprocess
import sys

CLASSES = [
    ("mad64", re.compile(r"^v_mad_[iu]64_[iu]32")),
    ("mul_lo/hi", re.compile(r"^v_mul_(lo|hi)_[iu]32")),
    ("shift64", re.compile(r"^v_(ashrrev|lshlrev|lshrrev)_[ib]64|^v_lshl_add_u64")),
    ("dpp_mov", re.compile(r"^v_mov_b32_dpp")),
    ("mov", re.compile(r"^v_mov_b32|^v_accvgpr|^v_mov_b64")),
    ("cndmask", re.compile(r"^v_cndmask")),
    ("addsub32", re.compile(r"^v_(add|sub|subrev)(_co)?_[iu]32|^v_(addc|subb|subbrev)_co_u32|^v_add3|^v_lshl_add_u32|^v_add_lshl")),
    ("logic", re.compile(r"^v_(and|or|xor|not|bfe|bfi|lshlrev_b32|lshrrev_b32|ashrrev_i32|and_or|or3|lshl_or|alignbit)")),
    ("cmp", re.compile(r"^v_cmp")),
    ("scratch_ld", re.compile(r"^scratch_load")),
    ("scratch_st", re.compile(r"^scratch_store")),
    ("flat/global", re.compile(r"^(flat|global|buffer)_")),
    ("lds", re.compile(r"^ds_")),
    ("v_other", re.compile(r"^v_")),
    ("salu", re.compile(r"^s_")),
]


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    cur, hist = None, collections.OrderedDict()
    for line in open(path):
        m = re.match(r"^([A-Za-z_][\w.$]*):", line)
        if m and not m.group(1).startswith((".L", "BB")):
            cur = m.group(1)
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) or cur is None:
            continue
        op = s.split()[0]
        for name, rx in CLASSES:
            if rx.match(op):
                hist.setdefault(cur, collections.Counter())[name] += 1
                break
    names = list(hist)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for n, d in zip(names, dem):
        if subs and not any(x in d for x in subs):
            continue
        h = hist[n]
        tot = sum(h.values())
        if tot < 50:
            continue
        valu = sum(v for k, v in h.items() if k not in ("scratch_ld", "scratch_st", "flat/global", "lds", "salu"))
        print("%s\n   total %d  valu %d  mad64 %.1f%% of valu | %s" % (d[:200], tot, valu, 100.0 * h["mad64"] / max(valu, 1),
              "  ".join("%s %d" % (k, h[k]) for k, _ in CLASSES if h[k])))


if __name__ == "__main__":
    main()
