#!/usr/bin/env python3
"""tools/timeline_report.py -- reads the records tools/timeline.hip wrote (gpurun_out/timeline.bin) and prints, per
scenario and launch: the kernel's span, where a wavefront's life goes (mean / p5 / p95 of every phase between two
stamps) and, on the common 100 MHz time base, how many wavefronts per CU are in an arithmetic phase at any time.

Stamps per wavefront: entry, loaded, (barrier arrive, barrier leave)*, last store issued, stores acknowledged.
Developer tool; nothing in the product imports it."""
import sys
import numpy as np

TICK_US = 0.01   # s_memrealtime: 100 MHz


def read_records(path):
    recs = []
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        nl = data.index(b"\n", pos)
        hdr = data[pos:nl].decode().split()
        if not hdr:
            pos = nl + 1
            continue
        assert hdr[0] == "REC", hdr
        scen, label, waves, ns, us, launches = hdr[1], hdr[2], int(hdr[3]), int(hdr[4]), float(hdr[5]), int(hdr[6])
        nbytes = waves * ns * 8
        arr = np.frombuffer(data[nl + 1: nl + 1 + nbytes], dtype=np.uint64).reshape(waves, ns)
        recs.append(dict(scen=scen, label=label, us=us, launches=launches, arr=arr))
        pos = nl + 1 + nbytes + 1
    return recs


def phase_names(nst):
    # nst stamps: entry, loaded, (a, l) * nb, issued, acked
    nb = (nst - 4) // 2
    names = ["load (issue%s)" % "", "round 1 + park" if nb != 1 else "rounds 1 + 2 (no workgroup barrier)"]
    seg = ["wait barrier %d" % (i + 1) for i in range(nb)]
    comp = {3: ["read", "round 2 + park", "read + round 3 + twiddle + stores"],
            1: ["read + round 3 + twiddle + stores"]}.get(
        nb, ["compute %d" % (i + 1) for i in range(nb)])
    out = [names[0], names[1]]
    for i in range(nb):
        out.append(seg[i])
        out.append(comp[i] if i < len(comp) else "compute %d" % (i + 1))
    out.append("store drain")
    return out


def is_compute(idx, nst):
    # segment idx (0-based between stamp idx and idx+1): 0 load, 1 compute, then alternating wait / compute, last = drain
    nseg = nst - 1
    if idx == 0 or idx == nseg - 1:
        return False
    return idx % 2 == 1


def report(rec, out):
    arr = rec["arr"]
    L = rec["launches"]
    per = arr.shape[0] // L
    out.write("=" * 110 + "\n")
    out.write("scenario %s / %s: %.2f us per transform in the timed region\n" % (rec["scen"], rec["label"], rec["us"]))
    spans = []
    for l in range(L):
        a = arr[l * per:(l + 1) * per]
        k = (a[:, 0] & 0xFF).astype(int)
        ok = k > 2
        if not ok.any():
            continue
        a = a[ok]
        nst = int(k[ok][0]) - 2
        st = a[:, 2:2 + nst].astype(np.int64)
        t0 = st[:, 0].min()
        span = (st[:, -1].max() - t0) * TICK_US
        spans.append((l, t0, st))
        hw = (a[:, 0] >> 8).astype(np.int64)
        cu = (hw >> 8) & 0xF
        sh = (hw >> 12) & 0x1
        se = (hw >> 13) & 0x7
        xcc = (a[:, 1] & 0xF).astype(np.int64)
        ncu = len(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))
        out.write("-- launch %d: %d wavefronts on %d distinct (xcc, se, sh, cu); span %.2f us (first entry -> last store acknowledged); "
                  "last store ISSUED at %.2f us\n" % (l, len(a), ncu, span, (st[:, -2].max() - t0) * TICK_US))
        ent = (st[:, 0] - t0) * TICK_US
        out.write("   entry skew: p50 %.2f  p95 %.2f  max %.2f us\n" % (np.percentile(ent, 50), np.percentile(ent, 95), ent.max()))
        names = phase_names(nst)
        d = np.diff(st, axis=1) * TICK_US
        tot = d.sum(axis=1).mean()
        for i, nm in enumerate(names):
            out.write("   %-38s mean %6.2f us (%4.1f %%)   p5 %6.2f  p95 %6.2f\n" %
                      (nm, d[:, i].mean(), 100 * d[:, i].mean() / tot, np.percentile(d[:, i], 5), np.percentile(d[:, i], 95)))
        comp = sum(d[:, i].mean() for i in range(len(names)) if is_compute(i, nst))
        out.write("   wavefront life %.2f us: arithmetic phases %.2f us (%.0f %%), load %.2f, barrier waits %.2f, drain %.2f\n" %
                  (tot, comp, 100 * comp / tot, d[:, 0].mean(),
                   sum(d[:, i].mean() for i in range(2, len(names) - 1) if not is_compute(i, nst)), d[:, -1].mean()))
    if not spans:
        return
    # common time base: wavefronts in an arithmetic phase per 0.5 us bin, all launches together
    T0 = min(s[1] for s in spans)
    T1 = max(s[2][:, -1].max() for s in spans)
    bin_t = 50   # ticks = 0.5 us
    nb = int((T1 - T0) // bin_t) + 1
    compw = np.zeros(nb)
    memw = np.zeros(nb)
    barw = np.zeros(nb)
    for (l, t0, st) in spans:
        nst = st.shape[1]
        for i in range(nst - 1):
            a0 = (st[:, i] - T0).astype(np.float64) / bin_t
            a1 = (st[:, i + 1] - T0).astype(np.float64) / bin_t
            tgt = compw if is_compute(i, nst) else (memw if (i == 0 or i == nst - 2) else barw)
            # add the overlap of [a0, a1) with each bin
            for w0, w1 in zip(a0, a1):
                b0, b1 = int(w0), int(min(w1, nb - 1e-9))
                if b1 == b0:
                    tgt[b0] += w1 - w0
                else:
                    tgt[b0] += (b0 + 1) - w0
                    tgt[b0 + 1:b1] += 1
                    tgt[b1] += w1 - b1
    out.write("-- wavefronts per CU (of 256 CUs) by state, 0.5 us bins from the first entry: arithmetic / load+drain / barrier\n")
    line = []
    for b in range(nb):
        line.append("%5.1f:%4.1f/%4.1f/%4.1f" % (b * 0.5, compw[b] / 256, memw[b] / 256, barw[b] / 256))
    for i in range(0, len(line), 6):
        out.write("   " + "  ".join(line[i:i + 6]) + "\n")
    out.write("   mean arithmetic wavefronts per CU over the window: %.2f (16 = four per SIMD)\n" % (compw.sum() / nb / 256))


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/timeline.bin"
    recs = read_records(path)
    for r in recs:
        report(r, sys.stdout)


if __name__ == "__main__":
    main()
