"""Randomised parity run on the GPU box: the HIP scan (C ABI) against the CPU oracle on freshly drawn cases for a given
number of seconds.  Every case draws its own whitelist (with FAMILIES of near-identical barcodes, so ties and the
order-dependent fold are the rule, not the exception), rows, error rates, irregular rows / entries, max_ed, tie rule,
shortcut, scan mode and row stride.  Bit-exact on hit / ed / end / flags or the case is saved and the exit code is 1.

usage: python tools/fuzz_parity.py [seconds] [first_seed] [--big]      (test infrastructure: uses oracle/ as the checker)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from blaze_amd import _lib, synth  # noqa: E402
from blaze_amd.engine import Engine  # noqa: E402
from oracle import oracle  # noqa: E402

FLAG_MASK = _lib.BZ_F_EXACT | _lib.BZ_F_BLANK | _lib.BZ_F_AMBIG
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def mutate(rng, w, n_edits):
    """n_edits random substitutions / insertions / deletions of a barcode, cut or padded back to 16 nt."""
    s = list(w)
    for _ in range(n_edits):
        kind = rng.integers(0, 3)
        j = int(rng.integers(0, len(s)))
        if kind == 0:
            s[j] = int(ACGT[rng.integers(0, 4)])
        elif kind == 1:
            s.insert(j, int(ACGT[rng.integers(0, 4)]))
        elif len(s) > 1:
            del s[j]
    while len(s) < 16:
        s.append(int(ACGT[rng.integers(0, 4)]))
    return bytes(s[:16])


def draw_whitelist(rng, n_cells, family_frac):
    """Distinct 16-mers; a share of them are 1..3-edit variants of other entries."""
    out, seen = [], set()
    while len(out) < n_cells:
        if out and rng.random() < family_frac:
            w = mutate(rng, out[int(rng.integers(0, len(out)))], int(rng.integers(1, 4)))
        else:
            w = ACGT[rng.integers(0, 4, 16)].tobytes()
        if w not in seen:
            seen.add(w)
            out.append(w)
    order = rng.permutation(n_cells)
    return [out[k] for k in order]


def draw_case(rng, big=False):
    """big: one case in eight takes a whitelist of 4,000 ... 70,000 entries (long candidate lists of the two-kernel path, and
    above ~45,000 the streaming pair-seeded kernel) with a few hundred rows."""
    n_cells = int(rng.choice([1, 2, 3, 5, 31, 63, 64, 65, 127, 200, 257, 777, 1024, 2048, 3001]))
    large = big and rng.random() < 0.125
    if large:
        n_cells = int(rng.integers(4_000, 70_000))          # both sides of the candidate-list limit (~45k) and of 65,536
    family = float(rng.choice([0.0, 0.2, 0.6]))
    wl = draw_whitelist(rng, n_cells, family)
    err = float(rng.choice([0.5, 1.0, 3.0, 6.0]))
    n_rows = int(rng.choice([1, 63, 64, 65, 500, 2048, 2500, 4097]))
    if large:
        n_rows = min(n_rows, 500)
    # rows: a barcode of the list (or a random 16-mer) with errors, flanks and umi around it
    ext = np.zeros((n_rows, 32), dtype=np.uint8)
    el = np.zeros(n_rows, dtype=np.uint8)
    bo = np.zeros(n_rows, dtype=np.uint8)
    bl = np.zeros(n_rows, dtype=np.uint8)
    p_edit = 0.02 * err
    for r in range(n_rows):
        u = rng.random()
        if u < 0.1:
            continue                                              # blank row
        src = wl[int(rng.integers(0, n_cells))] if rng.random() < 0.9 else ACGT[rng.integers(0, 4, 16)].tobytes()
        body = bytearray()
        for c in ACGT[rng.integers(0, 4, 2)].tobytes() + src + ACGT[rng.integers(0, 4, 6)].tobytes():
            v = rng.random()
            if v < p_edit:
                body.append(int(ACGT[rng.integers(0, 4)]))        # substitution
            elif v < 1.5 * p_edit:
                body.append(c); body.append(int(ACGT[rng.integers(0, 4)]))   # insertion
            elif v < 2.0 * p_edit:
                continue                                          # deletion
            else:
                body.append(c)
        body = bytes(body[:20]).ljust(20, b"A")
        ext[r, :20] = np.frombuffer(body, dtype=np.uint8)
        el[r], bo[r], bl[r] = 20, 2, 16
    irregular_rows = rng.random() < 0.35
    if irregular_rows:
        for r in np.flatnonzero(bl > 0):
            if rng.random() > 0.15:
                continue
            k = int(rng.integers(0, 11))
            if k == 0:
                ext[r, rng.integers(0, 20)] = ord("N")
            elif k == 1:
                ext[r, rng.integers(0, 20)] |= 0x20
            elif k == 2:
                L = int(rng.integers(21, 33)); ext[r, 20:L] = ACGT[rng.integers(0, 4, L - 20)]; el[r] = L
            elif k == 3:
                L = int(rng.integers(1, 16)); el[r] = L; bo[r] = 0; bl[r] = L
            elif k == 4:
                ext[r, :19] = ext[r, 1:20]; el[r] = 19; bo[r] = 1
            elif k == 5:
                ext[r, :16] = ext[r, 2:18]; el[r] = 16; bo[r] = 0
            elif k == 6:
                el[r] = int(rng.integers(18, 20))
            elif k == 7:
                ext[r, 16:18] = ext[r, 18:20]; el[r] = 18; bl[r] = 14
            elif k == 8 and rng.random() < 0.5:                   # a barcode verbatim one column to the right of where the row says it is:
                w = np.frombuffer(wl[int(rng.integers(0, n_cells))], dtype=np.uint8)   # distance 0 without the shortcut
                if w.size == 16:
                    ext[r, :] = 0; ext[r, 0:3] = ACGT[rng.integers(0, 4, 3)]; ext[r, 3:19] = w; ext[r, 19] = ACGT[rng.integers(0, 4)]
                    el[r], bo[r], bl[r] = 20, 2, 16
            elif k == 8:                                          # an exact copy of an entry, barcode at offset 1
                w = np.frombuffer(wl[int(rng.integers(0, n_cells))], dtype=np.uint8)
                ext[r, :] = 0; ext[r, 0] = ACGT[rng.integers(0, 4)]; ext[r, 1:17] = w; ext[r, 17:19] = ACGT[rng.integers(0, 4, 2)]
                el[r], bo[r], bl[r] = 19, 1, 16
            elif k == 9:                                          # the barcode further right than any host puts it (the ABI takes any offset)
                bo[r] = int(rng.integers(3, 5)); bl[r] = 16
            elif k == 10:                                         # a byte that is no letter at all
                ext[r, rng.integers(0, 20)] = int(rng.choice([0, 1, 0x7F, 0x80, 0xC1, 0xFF, ord("-"), ord("*")]))
    ext[np.arange(32)[None, :] >= el[:, None]] = 0
    mixed = rng.random() < 0.3
    if mixed:
        n_odd = int(rng.integers(1, 6)) if rng.random() > 0.15 else int(rng.integers(260, 700))     # a few, or hundreds
        for _ in range(n_odd):
            k = int(rng.integers(0, 10))
            base = wl[int(rng.integers(0, len(wl)))]
            if len(base) != 16:
                continue
            new = [base[:15], base + b"AC", base[:6] + b"N" + base[7:], base.lower(), base[:12], b"",      # b"": a blank line of the list
                   base + base[::-1] + b"ACGTAC", base * 5,              # 38 and 80 characters: past the match masks, past any row
                   base + base[:9], base + base[2:]][k]                  # 25 and 30: longer than a 20-column row, still on the fast path
            pos = int(rng.integers(0, len(wl) + 1))
            if rng.random() < 0.5 and len(wl) > 1:
                wl[wl.index(base)] = new                           # replaces its parent
            else:
                wl.insert(pos, new)
    if rng.random() < 0.04 and len(wl) <= 5000:              # a list without one regular entry: 15-mers, 17-mers or lower case
        kind = int(rng.integers(0, 3))
        wl = [(w[:15] if kind == 0 else w + b"A" if kind == 1 else w.lower()) if len(w) == 16 else w for w in wl]
    max_ed = int(rng.choice([0, 1, 2, 2, 2, 3, 3, 4, 5, 6, 7, 8]))
    params = dict(n_cells=len(wl), family=family, err=err, n_rows=n_rows, irregular_rows=bool(irregular_rows), mixed=bool(mixed),
                  max_ed=max_ed, tie=int(rng.integers(0, 2)), shortcut=bool(rng.random() < 0.8),
                  scan_mode=int(rng.choice([0, 0, 1, 2, 3, 4])), stride=int(rng.choice([20, 24, 32])))
    # (drawn last, so that every earlier seed keeps its case) rows of 33..64 bytes, the device's limit: what the file hosts
    # send for hand-made barcode cells of up to 60 nt; an entry of the list verbatim or a few edits away somewhere inside
    params["wide_rows"] = bool(rng.random() < 0.15)
    if params["wide_rows"]:
        ext = np.concatenate([ext, np.zeros((n_rows, 32), dtype=np.uint8)], axis=1)
        for r in np.flatnonzero(bl > 0):
            if rng.random() > 0.25:
                continue
            L = int(rng.integers(33, 65))
            row = ACGT[rng.integers(0, 4, L)].copy()
            w = np.frombuffer(wl[int(rng.integers(0, len(wl)))], dtype=np.uint8)
            if w.size and w.size <= L and rng.random() < 0.8:
                w = np.frombuffer(mutate(rng, w.tobytes(), int(rng.integers(0, 4))), dtype=np.uint8)[:L]
                at = int(rng.integers(0, L - w.size + 1))
                row[at:at + w.size] = w
            if rng.random() < 0.1:
                row[rng.integers(0, L)] = ord("N")
            off, umi = (int(rng.integers(0, 3)) if rng.random() < 0.7 else int(rng.integers(3, 12))), int(rng.integers(0, 3))
            ext[r, :] = 0
            ext[r, :L] = row
            el[r], bo[r], bl[r] = L, off, L - off - umi
    return params, wl, ext, el, bo, bl


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    big = "--big" in sys.argv
    seconds = float(argv[0]) if len(argv) > 0 else 120.0
    seed0 = int(argv[1]) if len(argv) > 1 else 1
    oracle.build()
    eng = Engine(0)
    os.makedirs("gpurun_out", exist_ok=True)
    t_end = time.time() + seconds
    n_cases = n_rows = n_fail = 0
    stats = {}
    seed = seed0
    last_print = time.time()
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        p, wl, ext, el, bo, bl = draw_case(rng, big)
        ext_s = np.ascontiguousarray(ext[:, :p["stride"]]) if p["stride"] >= int(el.max(initial=0)) else ext
        if ext_s is ext:
            p["stride"] = int(ext.shape[1])
        eng.set_option(_lib.BZ_OPT_SCAN_MODE, 0 if p["scan_mode"] == 4 else p["scan_mode"])
        eng.set_option(_lib.BZ_OPT_PAIR_FUSED, int(p["scan_mode"] == 4))
        eng.set_option(_lib.BZ_OPT_TIE_RULE, p["tie"])
        eng.set_option(_lib.BZ_OPT_SHORTCUT, int(p["shortcut"]))
        eng.set_whitelist(wl)
        hit, ed, end, flags = eng.assign(ext_s, el, bo, bl, p["max_ed"])
        o_hit, o_ed, o_end, o_flag = oracle.match_rows(ext, el, bo, bl, wl, p["max_ed"], p["tie"], p["shortcut"], threads=16)
        bad = np.flatnonzero((hit != o_hit) | (ed != o_ed) | (end != o_end) | ((flags & FLAG_MASK) != o_flag))
        n_cases += 1
        n_rows += ext.shape[0]
        key = f"ed{p['max_ed']}_m{p['scan_mode']}" + ("_big" if p["n_cells"] > 3_500 else "")
        stats[key] = stats.get(key, 0) + 1
        if bad.size:
            n_fail += 1
            r = int(bad[0])
            print(f"FAIL seed {seed}: {json.dumps(p)}: {bad.size} rows differ, first {r}: ext={bytes(ext[r, :el[r]])!r} off={bo[r]} bl={bl[r]} "
                  f"gpu=({hit[r]},{ed[r]},{end[r]},{flags[r]}) oracle=({o_hit[r]},{o_ed[r]},{o_end[r]},{o_flag[r]})", flush=True)
            np.savez_compressed(f"gpurun_out/fuzz_fail_{seed}.npz", ext=ext, el=el, bo=bo, bl=bl, wl=np.array(wl, dtype=object).astype("S"),
                                params=json.dumps(p))
            if n_fail >= 10:
                break
        seed += 1
        if time.time() - last_print > 30:
            last_print = time.time()
            print(f"... {n_cases} cases, {n_rows} rows, {n_fail} failures", flush=True)
    print(json.dumps({"cases": n_cases, "rows": n_rows, "failures": n_fail, "first_seed": seed0, "next_seed": seed, "seconds": seconds,
                      "cases_by_max_ed_and_mode": dict(sorted(stats.items()))}), flush=True)
    sys.exit(1 if n_fail else 0)


if __name__ == "__main__":
    main()
