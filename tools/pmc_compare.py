#!/usr/bin/env python3
"""pmc_compare.py <outdir> <label>=<ENV=V,ENV=V...> ... -- run one bench.py command under rocprofv3 --pmc for several
environment variants and print the per-kernel counter averages side by side (developer tool; GPU box).
Counters come from WISHLIST filtered by `rocprofv3 -L`, in passes of at most 7 SQ counters."""
import glob, json, os, sqlite3, subprocess, sys

WISHLIST = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
            "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
            "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT",
            "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_INST_CYCLES_SMEM",
            "SQ_IFETCH", "SQ_IFETCH_LEVEL", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_SMEM", "SQ_WAVES_EQ_64",
            "SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE", "SQC_ICACHE_INPUT_VALID_READYB",
            "SQC_DCACHE_REQ", "SQC_DCACHE_HITS", "SQC_DCACHE_MISSES", "SQC_TC_REQ", "SQC_TC_INST_REQ", "SQC_TC_STALL",
            "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64",
            "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TA_TCP_STATE_READ_sum",
            "TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum",
            "TA_BUSY_avr", "TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_BUFFER_WAVEFRONTS_sum", "TA_FLAT_WAVEFRONTS_sum",
            "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "GRBM_GUI_ACTIVE", "GRBM_COUNT"]


def main():
    out = sys.argv[1]
    variants = []
    for a in sys.argv[2:]:
        if a.startswith("--cmd=") or a.startswith("--counters="):
            continue
        lab, _, envs = a.partition("=")
        variants.append((lab, dict(e.split("=", 1) for e in envs.split(",") if e)))
    cmd = [a[6:] for a in sys.argv[2:] if a.startswith("--cmd=")]
    root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
    cmd = cmd[0].split() if cmd else ["python", root + "/tools/mini_many.py", "3", "16"]
    os.makedirs(out, exist_ok=True)
    lst = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True).stdout
    open(os.path.join(out, "counters_list.txt"), "w").write(lst)
    wish = [a[11:].split(",") for a in sys.argv[2:] if a.startswith("--counters=")]
    wishlist = wish[0] if wish else WISHLIST
    have = [c for c in wishlist if ("\t" + c + "\n" in lst) or (" " + c + "\n" in lst) or (c + " " in lst) or (c + "\t" in lst) or (c + "\n" in lst)]
    missing = [c for c in wishlist if c not in have]
    print("missing counters:", missing)
    sq = [c for c in have if c.startswith("SQ")]
    other = [c for c in have if not c.startswith("SQ")]
    passes = [sq[i:i + 7] for i in range(0, len(sq), 7)] + [other[i:i + 3] for i in range(0, len(other), 3)]
    res = {}
    for lab, env in variants:
        for pi, cs in enumerate(passes):
            d = os.path.join(out, "%s_p%d" % (lab, pi))
            e = dict(os.environ); e.update(env); e["TMPDIR"] = "/tmp"
            r = subprocess.run(["rocprofv3", "--pmc"] + cs + ["-d", d, "--"] + cmd, capture_output=True, text=True, env=e, cwd="/tmp")
            f = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if not f:
                print("pass failed:", lab, cs, r.stderr[-300:])
                continue
            db = sqlite3.connect(f[0])
            q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
            try:
                for k, c, n, a in db.execute(q):
                    res.setdefault(k, {}).setdefault(c, {})[lab] = (n, a)
            except Exception as ex:
                print("query failed", lab, cs, ex)
            db.close()
            for g in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
                os.remove(g)
            dump(out, res, variants, wishlist)
            print("pass done:", lab, pi, cs, flush=True)
    dump(out, res, variants, wishlist)
    print(open(os.path.join(out, "pmc.txt")).read())


def dump(out, res, variants, wishlist):
    json.dump({k: {c: {l: v for l, v in d.items()} for c, d in cc.items()} for k, cc in res.items()}, open(os.path.join(out, "pmc.json"), "w"), indent=1)
    with open(os.path.join(out, "pmc.txt"), "w") as f:
        for k, cc in sorted(res.items()):
            if not any(n >= 8 for d in cc.values() for (n, a) in d.values()):
                continue
            f.write("== %s\n" % k[:150])
            for c in wishlist:
                if c in cc:
                    f.write("  %-40s " % c + "  ".join("%s: %14.1f (n=%d)" % (l, cc[c][l][1], cc[c][l][0]) for l, _ in variants if l in cc[c]) + "\n")


if __name__ == "__main__":
    main()
