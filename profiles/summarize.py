#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs of profiles/run_profile.sh into one text/JSON file.
usage: profiles/summarize.py gpurun_out/prof_<tag> profiles/<name>"""
import glob, json, os, sqlite3, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
res = {"kernels": [], "counters": {}}
lines = []
tdb = glob.glob(os.path.join(src, "trace", "*.db"))
if tdb:
    c = sqlite3.connect(tdb[0])
    lines.append("== rocprofv3 --kernel-trace --stats : kernel summary (durations in us) ==")
    lines.append("%-90s %6s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        res["kernels"].append(dict(name=name, calls=calls, total_us=total / 1.0, avg_us=avg / 1.0, pct=pct))
        lines.append("%-90s %6d %14.1f %12.1f %7.2f" % (name[:90], calls, total / 1.0, avg / 1.0, pct))
for pdb in sorted(glob.glob(os.path.join(src, "pmc*", "*.db"))):
    c = sqlite3.connect(pdb)
    acc = defaultdict(lambda: defaultdict(float)); nd = defaultdict(set)
    for kname, disp, cname, val in c.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
        if "render_kernel" not in kname and "trace_kernel" not in kname:
            continue
        k = kname.split("(")[0]
        acc[k][cname] += val; nd[k].add(disp)
    for k in acc:
        for cn, v in acc[k].items():
            res["counters"].setdefault(k, {})[cn] = v / len(nd[k])
for k, d in res["counters"].items():
    lines.append("")
    lines.append("== PMC counters, average per dispatch: %s ==" % k)
    for cn in sorted(d):
        lines.append("  %-28s %18.0f" % (cn, d[cn]))
    g = d.get
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        lines.append("  VALU lane utilisation  = THREAD_CYCLES_VALU / (64*ACTIVE_INST_VALU) = %.3f" % (g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU"))))
    if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
        lines.append("  VALU busy / wave cycles = %.3f" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
    if g("FETCH_SIZE") is not None:
        lines.append("  HBM read  (FETCH_SIZE KB x2 gfx950 correction) = %.1f MB" % (g("FETCH_SIZE") * 2 / 1024))
    if g("TCC_REQ_sum"):
        lines.append("  L2 (TCC) hit rate = %.3f, misses / requests = %.3f" % (g("TCC_HIT_sum") / g("TCC_REQ_sum"), g("TCC_MISS_sum") / g("TCC_REQ_sum")))
    if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs
        lines.append("  cycles per wave64 VALU instruction per SIMD = %.2f" % ((g("GRBM_GUI_ACTIVE") / 8) / (g("SQ_INSTS_VALU") / 1024)))
    if g("WRITE_SIZE") is not None:
        lines.append("  HBM write (WRITE_SIZE KB) = %.1f MB" % (g("WRITE_SIZE") / 1024))
open(dst + ".txt", "w").write("\n".join(lines) + "\n")
json.dump(res, open(dst + ".json", "w"), indent=1)
print("\n".join(lines))
