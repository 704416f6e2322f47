#!/usr/bin/env python
"""Offline (no GPU): what a different work order could do for the per-SIMD balance of the tile compositing kernels.

    python tools/wg_balance_sim.py [gpurun_out/wg_trace.npz]

Input: the per-workgroup rows tools/wg_trace.py recorded on an MI355X (start / end clock, cycles, HW_ID | XCC_ID | list length).  The forward's
busy workgroups are all resident at once and the dispatcher places workgroup w on SIMD w mod 1024 (checked below on the rows), so the work a SIMD
gets is decided by the order alone: list position of workgroup w = xcd_list_pos(w) (csrc/gsr_composite_common.h), list = the scan's class order.
Printed: sum of list lengths per SIMD, max / mean, for the recorded order and for the alternatives quoted in profiles/r03_issue_probes.md
section 4 (finer length classes, exact longest-first, the same with every second round of 1024 workgroups reversed, a greedy fill under the fixed
number of slots per SIMD), and the concurrency / per-wave rate numbers of section 2."""
import heapq
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def decode(rows):
    rows = rows.astype(np.uint64)
    ok = rows[:, 1] > 0
    hw = rows[:, 3] & np.uint64(0xffffffff)
    xcc = (rows[:, 3] >> np.uint64(32)) & np.uint64(0xff)
    length = (rows[:, 3] >> np.uint64(40)).astype(np.int64)
    simd, cu, sh, se = (hw >> np.uint64(4)) & np.uint64(3), (hw >> np.uint64(8)) & np.uint64(15), (hw >> np.uint64(12)) & np.uint64(1), (hw >> np.uint64(13)) & np.uint64(7)
    key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd).astype(np.int64)
    t0 = rows[ok, 0].min()
    return ok, key, length, (rows[:, 0] - t0) / 100.0, (rows[:, 1] - t0) / 100.0, rows[:, 2].astype(np.float64)


def main(path):
    d = np.load(path)
    for name in ("fwd", "bwd"):
        ok, key, length, start, end, cyc = decode(d[name])
        span = end[ok].max()
        ts = np.linspace(0, span, 11)[1:-1]
        conc = [round(float(((start[ok] <= t) & (end[ok] > t)).sum() / 1024.0), 2) for t in ts]
        print("%s: %d busy workgroups, mean list %.0f, span %.1f us; waves resident per SIMD at 10..90 %% of the span: %s" % (name, ok.sum(), length[ok].mean(), span, conc))
        long_ = ok & (length >= 256)
        print("    shader cycles per list entry of a wave (lists >= 256, sharing its SIMD): %.0f" % (cyc[long_].sum() / length[long_].sum()))
    ok, key, length, start, end, cyc = decode(d["fwd"])
    NB, busy = len(ok), int(ok.sum())
    span = ((busy + 511) >> 9) << 9

    def xcd_list_pos(w):
        w = np.asarray(w)
        x, q = w & 7, w >> 3
        return np.where(w < span, (((q >> 6) << 3) + x) * 64 + (q & 63), w)

    w_all = np.arange(NB)
    lst = np.zeros(NB, np.int64)
    lst[xcd_list_pos(w_all)] = length  # list (position) -> length, as the scan ordered it
    same = ok[1024:2048] & ok[:1024]
    print("placement: workgroups w and w + 1024 on the same SIMD: %.3f of %d pairs" % (float(np.mean(key[1024:2048][same] == key[:1024][same])), same.sum()))

    def ratio(listL, mapping):
        w = np.arange(span)
        sums = np.bincount(w % 1024, weights=listL[mapping(w)], minlength=1024)
        return sums.max() / sums.mean()

    def snake(w):
        r, x = w >> 10, w & 1023
        return np.where(r % 2 == 1, (r << 10) | (1023 - x), w)

    print("forward, sum of list lengths per SIMD, max / mean:")
    print("  recorded order (3 length classes, patch order inside)   %.3f   with every second round of 1024 reversed %.3f" % (ratio(lst, xcd_list_pos), ratio(lst, lambda w: xcd_list_pos(snake(w)))))
    lb = lst[:busy]
    for ncls in (8, 16, 64):
        cls = np.minimum((lb * ncls) // (lb.max() + 1), ncls - 1)
        l2 = lst.copy()
        l2[:busy] = lb[np.argsort(-cls, kind="stable")]
        print("  %2d linear length classes                                 %.3f   reversed rounds %.3f" % (ncls, ratio(l2, xcd_list_pos), ratio(l2, lambda w: xcd_list_pos(snake(w)))))
    l2 = lst.copy()
    l2[:busy] = np.sort(lb)[::-1]
    print("  exact longest-first                                       %.3f   reversed rounds %.3f" % (ratio(l2, xcd_list_pos), ratio(l2, lambda w: xcd_list_pos(snake(w)))))
    slots = np.bincount(np.arange(busy) % 1024, minlength=1024)
    heap = [(0.0, s) for s in range(1024)]
    heapq.heapify(heap)
    used, sums = np.zeros(1024, int), np.zeros(1024)
    for item in np.sort(lb)[::-1]:
        while True:
            w_, s = heapq.heappop(heap)
            if used[s] < slots[s]:
                break
        used[s] += 1
        sums[s] += item
        heapq.heappush(heap, (sums[s], s))
    print("  greedy longest-first under the fixed slots per SIMD       %.3f" % (sums.max() / sums.mean()))
    print("  list length percentiles 0 / 5 / 25 / 50 / 75 / 95 / 100: %s" % np.percentile(lb, [0, 5, 25, 50, 75, 95, 100]).round(0).tolist())


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "wg_trace.npz"))
