"""Turns the raw rocprofv3 output of tools/collect_profiles.sh into the summaries committed under profiles/:
  <tag>_bench_kernel_stats.txt         kernel stats of the profiled bench run + its JSON line
  <tag>_single_image_kernel_stats.txt  kernel stats and the per-launch timeline of one context extracting 5 images
  <tag>_kernel_counters.txt            every counter, per kernel and per image
  <tag>_kernel_counters.json           per pipeline stage: HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, gfx950 note of
                                       MI355X_MICROARCH.md 'HBM') and vector instructions per image -- what bench.py quotes
usage: summarize_profiles.py <tag> <raw dir> <out dir> <images per counter pass>"""
import collections
import csv
import glob
import json
import os
import sys

TAG, RAW, OUT, NIMG = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
STAGES = (("pyramid", ("k_blur_tile", "k_blur_duo", "k_blur_small", "k_blur_march", "k_pyr_tail")), ("detect", ("k_detect",)), ("refine", ("k_refine", "k_filter")),
          ("orientation", ("k_orientation",)), ("scan", ("k_scan_",)), ("descriptor", ("k_descriptor",)))


def short(n):
    return n.replace("popsift_hip::(anonymous namespace)::", "").replace("popsift_hip::", "").replace("void ", "").split("(")[0]


def stats(sub):
    f = sorted(glob.glob(os.path.join(RAW, sub, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    lines = ["%-44s %7s %12s %11s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for r in rows:
        lines.append("%-44s %7s %12.1f %11.2f %8s" % (short(r["Name"])[:44], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                     float(r["AverageNs"]) / 1e3, r["Percentage"][:6]))
    return "\n".join(lines)


def timeline(sub, nimg):
    f = sorted(glob.glob(os.path.join(RAW, sub, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-(len(rows) // nimg):]
    t0 = int(last[0]["Start_Timestamp"])
    out, tot = [], 0
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        tot += e - s
        out.append("%-36s @%9.1f us %9.1f us  grid %9s  wg %4s  vgpr %4s  lds %6s" % (
            short(r["Kernel_Name"])[:36], (s - t0) / 1e3, (e - s) / 1e3, r["Grid_Size_X"], r["Workgroup_Size_X"],
            r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?")))
    out.append("sum of kernel durations %.1f us; first-to-last span %.1f us" % (
        tot / 1e3, (int(last[-1]["End_Timestamp"]) - t0) / 1e3))
    return "\n".join(out)


def counters(sub):
    """kernel -> counter -> sum over all dispatches of the pass"""
    fs = glob.glob(os.path.join(RAW, sub, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.OrderedDict()
    if not fs:
        return agg
    for r in csv.DictReader(open(sorted(fs, key=os.path.getmtime)[-1])):
        k = agg.setdefault(short(r["Kernel_Name"]), collections.OrderedDict())
        k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return agg


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    bench_line = [l for l in open(os.path.join(RAW, "bench.log")).read().splitlines() if l.startswith('{"metric"')][-1]
    open(os.path.join(OUT, "%s_bench_kernel_stats.txt" % TAG), "w").write(
        "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline\n"
        "# (1 MI355X; includes warm-up and the single-image / profile-mode / sparse / host-to-host legs: for the timed loop alone see\n"
        "#  %s_bench_quick_kernel_stats.txt)\n" % TAG
        + stats("bench") + "\n\n# bench.py output of this profiled run:\n" + bench_line + "\n")
    open(os.path.join(OUT, "%s_single_image_kernel_stats.txt" % TAG), "w").write(
        "# rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py 5   (one context, 5 x config-2 image)\n" + stats("single") +
        "\n\n# per-launch timeline of the last image\n" + timeline("single", 5) + "\n")
    if os.path.isdir(os.path.join(RAW, "quick")):
        qline = [l for l in open(os.path.join(RAW, "quick.log")).read().splitlines() if l.startswith("{")][-1]
        open(os.path.join(OUT, "%s_bench_quick_kernel_stats.txt" % TAG), "w").write(
            "# rocprofv3 --kernel-trace --stats -- python3 bench.py --quick   (the TIMED LOOP alone: warm-up + timed steps, dense images only;\n"
            "# a launch extracts a batch of images, so durations are per launch, not per image)\n" + stats("quick") +
            "\n\n# bench.py --quick output of this profiled run:\n" + qline + "\n")
    allc = collections.OrderedDict()
    for sub in ("sq_inst", "sq_wait", "sq_lds", "tcc", "grbm", "fetch", "write"):
        for k, v in counters(sub).items():
            allc.setdefault(k, collections.OrderedDict()).update(v)
    names = []
    for v in allc.values():
        for c in v:
            if c not in names:
                names.append(c)
    lines = ["# rocprofv3 --pmc <group> --kernel-trace -- python3 tools/prof_run.py %d: one pass per counter group (tools/collect_profiles.sh)." % NIMG,
             "# Values are PER IMAGE (sum over the kernel's dispatches of the pass / %d images).  FETCH_SIZE / WRITE_SIZE in KiB as" % NIMG,
             "# reported; hbm_MB = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / 1e6 (gfx950 counts half of a wide read, MI355X_MICROARCH.md).",
             "# SQ_*_CYCLES count quad-cycles.", "%-34s" % "kernel" + " ".join("%14s" % n[-14:] for n in names) + "%12s" % "hbm_MB"]
    stage = collections.OrderedDict((s, {"hbm_bytes": 0.0, "valu_insts": 0.0, "kernels": []}) for s, _ in STAGES)
    for k, v in allc.items():
        hbm = (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 / NIMG
        lines.append("%-34s" % k[:34] + " ".join("%14.5g" % (v.get(n, 0.0) / NIMG) for n in names) + "%12.2f" % (hbm / 1e6))
        for s, pre in STAGES:
            if k.startswith(pre):
                stage[s]["hbm_bytes"] += hbm
                stage[s]["valu_insts"] += v.get("SQ_INSTS_VALU", 0.0) / NIMG
                stage[s]["kernels"].append(k)
    open(os.path.join(OUT, "%s_kernel_counters.txt" % TAG), "w").write("\n".join(lines) + "\n")
    stage["_source"] = "%s_kernel_counters.txt (rocprofv3 --pmc passes of tools/prof_run.py, per image)" % TAG
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from popsift_amd.srchash import kernel_source_hash
    stage["_kernel_source_hash"] = kernel_source_hash()   # bench.py: counters of other kernels than the tree's are stale
    json.dump(stage, open(os.path.join(OUT, "%s_kernel_counters.json" % TAG), "w"), indent=1)
    print(json.dumps(stage)[:1500])
