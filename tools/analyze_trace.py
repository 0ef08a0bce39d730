#!/usr/bin/env python
"""Per-tile phase timeline of policy_step_kernel from an IC3_PS_TRACE build (see policy_step.hip):
python tools/analyze_trace.py trace.csv [--epi]  — s_memrealtime ticks are 10 ns.  --epi: a -DIC3_PS_TRACE_EPI build (three
stamps inside the cell epilogue instead of the heads / draws / env step stamps)."""
import sys

import numpy as np

# round-3 kernel: S2 + S4 share a phase (no barrier between them), the old cell state is requested behind S3
NAMES = ["start", "S0 loads", "S1 desc", "S2 enc + S4 h->LDS", "S3 enc->acc+c ld", "-", "S5 comm+Bload", "S6 C product",
         "S7 inp->LDS", "S8 gate loop", "settle", "barrier", "S9 epilogue", "S10 heads", "S11 draws",
         "S12 env step", "patch wait", "patches"]


NAMES_EPI = NAMES[:12] + ["S9a c_old there, 1st element", "S9b elements", "S9c head W->LDS, rest of fill", "S9d barrier",
                         "S10-12 heads, draws, env step, patch wait", "patches"]


def clk(path):
    """-DIC3_PS_TRACE_CLK build: shader clock of every gate loop = s_memtime cycles / s_memrealtime time"""
    d = np.loadtxt(path, delimiter=',', dtype=np.int64)
    us = (d[:, 10] - d[:, 9]).astype(np.float64) * 0.01          # slots 9 - 8 (column 0 is the tile index)
    cyc = (d[:, 15] - d[:, 14]).astype(np.float64)               # slots 14 - 13
    ok = (us > 1) & (cyc > 0)
    ghz = cyc[ok] / us[ok] / 1e3
    t0 = d[:, 1].min()
    start = (d[ok, 9] - t0) * 0.01
    print("gate loops: %d; shader clock while they ran: mean %.3f GHz, p10 %.3f, p50 %.3f, p90 %.3f" % (
        ok.sum(), ghz.mean(), np.percentile(ghz, 10), np.percentile(ghz, 50), np.percentile(ghz, 90)))
    print("loop duration: mean %.1f us = %.0f cycles (p10 %.0f, p90 %.0f cycles)" % (
        us[ok].mean(), cyc[ok].mean(), np.percentile(cyc[ok], 10), np.percentile(cyc[ok], 90)))
    last = start > np.percentile(start, 97)
    print("last 3 %% of the loops by start time (CUs draining): %.1f us = %.0f cycles at %.3f GHz" % (
        us[ok][last].mean(), cyc[ok][last].mean(), ghz[last].mean()))


def ws(path, G=256):
    """The wave-specialised kernel (policy_step_ws.hpp, IC3_PS_WS=1 on an IC3_PS_TRACE build): per tile the helper waves'
    front phases F (slots 0..8), the matrix waves' gate loop + epilogue G (9 = A tile there, 10 = loop end, 11 = epilogue end)
    and the helpers' back phases B (12 = h' there ... 17); tiles of one workgroup are b, b + G, ... behind its small tile."""
    d = np.loadtxt(path, delimiter=',', dtype=np.int64)
    t = d[:, 1:19].astype(np.float64) * 0.01
    ok = t[:, 0] > 0
    t0 = t[ok, 0].min()
    t = t - t0
    ntiles = len(d)
    print("tiles %d (stamped %d), launch span %.1f us" % (ntiles, ok.sum(), t[ok, 17].max()))
    F = t[:, 8] - t[:, 0]
    wait_f = t[:, 9] - t[:, 8]                    # matrix waves pick the tile up this long after the helpers finished it (< 0: they waited)
    loop = t[:, 10] - t[:, 9]
    epi = t[:, 11] - t[:, 10]
    wait_g = t[:, 12] - t[:, 11]                  # helpers start B this long after the epilogue ended
    B = t[:, 17] - t[:, 12]
    pw = t[:, 16] - t[:, 15]
    def row(name, x):
        x = x[ok]
        print("   %-46s mean %7.2f us  p10 %7.2f  p50 %7.2f  p90 %7.2f" % (name, x.mean(), np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90)))
    names = ["S0 loads", "S1 desc", "S2 enc + S4 h->LDS", "S3 enc->acc", "-", "S5 comm+Bload", "S6 C product", "S7 inp->LDS"]
    for k in range(8):
        row("F: " + names[k], t[:, k + 1] - t[:, k])
    row("F total (helpers, front phases of a tile)", F)
    row("A tile done -> matrix waves start (idle if > 0)", wait_f)
    row("G: gate loop (matrix waves)", loop)
    row("G: cell epilogue", epi)
    row("epilogue end -> helpers start B", wait_g)
    row("B: heads", t[:, 13] - t[:, 12])
    row("B: draws", t[:, 14] - t[:, 13])
    row("B: env step", t[:, 15] - t[:, 14])
    row("B: wait for the tile's zero stores", pw)
    row("B: patches + barrier", t[:, 17] - t[:, 16])
    row("B total", B)
    # per workgroup: the period of its matrix waves (loop start to next loop start) and their idle share
    per, idle = [], []
    for b in range(min(G, ntiles)):
        ids = [i for i in range(b, ntiles, G) if ok[i]]
        ids.sort(key=lambda i: t[i, 9])
        for x, y in zip(ids[:-1], ids[1:]):
            per.append(t[y, 9] - t[x, 9])
            idle.append(t[y, 9] - t[x, 11])
    if per:
        per, idle = np.array(per), np.array(idle)
        print("matrix-wave period per tile: mean %.2f us (p10 %.2f p90 %.2f); of it idle between epilogue end and the next loop: %.2f us"
              % (per.mean(), np.percentile(per, 10), np.percentile(per, 90), idle.mean()))
    fin = np.array([t[[i for i in range(b, ntiles, G) if ok[i]], 17].max() for b in range(min(G, ntiles))])
    print("workgroup finish: mean %.1f p10 %.1f p90 %.1f max %.1f us;  first gate loop starts at %.1f us (mean)" % (
        fin.mean(), np.percentile(fin, 10), np.percentile(fin, 90), fin.max(),
        np.mean([min(t[i, 9] for i in range(b, ntiles, G) if ok[i]) for b in range(min(G, ntiles))])))


def main(path, epi=False):
    global NAMES
    if epi:
        NAMES = NAMES_EPI
    d = np.loadtxt(path, delimiter=',', dtype=np.int64)
    t = d[:, 1:19].astype(np.float64) * 0.01          # us
    hw, xcc = d[:, 19], d[:, 20]
    t0 = t[:, 0].min()
    start, end = t[:, 0] - t0, t[:, 17] - t0
    print("tiles %d, launch span %.1f us" % (len(d), end.max()))
    cu = (xcc & 15) * 1024 + ((hw >> 13) & 7) * 64 + ((hw >> 12) & 1) * 32 + ((hw >> 8) & 15)   # xcc, se, sh, cu
    print("distinct CUs seen: %d" % len(np.unique(cu)))
    order = np.argsort(start)
    dur = np.diff(t, axis=1)
    life = end - start
    rounds = [("first 512 by start", order[:512]), ("next 512", order[512:1024]), ("rest", order[1024:])]
    for name, idx in rounds:
        if len(idx) == 0:
            continue
        print("\n%s: n=%d start %.1f..%.1f us, lifetime mean %.1f us (min %.1f max %.1f)" %
              (name, len(idx), start[idx].min(), start[idx].max(), life[idx].mean(), life[idx].min(), life[idx].max()))
        for k in range(17):
            print("   %-16s %6.2f us  (p10 %6.2f  p90 %6.2f)" % (NAMES[k + 1], dur[idx, k].mean(),
                                                               np.percentile(dur[idx, k], 10), np.percentile(dur[idx, k], 90)))
    # co-residency: for each tile of the last group, how many other tiles overlap it in time on the same CU
    print("\nper-CU tile counts: min %d max %d" % (np.bincount(np.unique(cu, return_inverse=True)[1]).min(),
                                                   np.bincount(np.unique(cu, return_inverse=True)[1]).max()))
    inv = np.unique(cu, return_inverse=True)[1]
    busy_end = np.zeros(inv.max() + 1)
    for c in range(inv.max() + 1):
        busy_end[c] = end[inv == c].max()
    print("CU finish time: mean %.1f p10 %.1f p90 %.1f max %.1f us" % (busy_end.mean(), np.percentile(busy_end, 10),
                                                                      np.percentile(busy_end, 90), busy_end.max()))
    # phase skew between co-resident workgroups: tiles on the same CU whose lifetimes overlap
    lags = []
    for c in range(inv.max() + 1):
        ids = np.where(inv == c)[0]
        ids = ids[np.argsort(start[ids])]
        for a, b in zip(ids[:-1], ids[1:]):
            if start[b] < end[a]:
                lags.append(t[b, 8] - t[a, 8])          # difference of gate-loop start times
    lags = np.array(lags)
    if len(lags):
        print("gate-loop start lag between overlapping tiles of a CU: mean %.1f us, p10 %.1f, p50 %.1f, p90 %.1f (n=%d)" %
              (lags.mean(), np.percentile(lags, 10), np.percentile(lags, 50), np.percentile(lags, 90), len(lags)))


if __name__ == '__main__' and '--ws' in sys.argv:
    ws([x for x in sys.argv[1:] if not x.startswith('--')][0])
    sys.exit(0)
if __name__ == '__main__':
    if '--clk' in sys.argv[2:]:
        clk(sys.argv[1])
    else:
        main(sys.argv[1], '--epi' in sys.argv[2:])
