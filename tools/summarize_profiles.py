"""Turns gpurun_out/profiles_raw (tools/collect_profiles.sh) into the committed summaries in profiles/."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", "profiles_raw")
OUT = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"


def short(n):
    return n.replace("popsift_hip::(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def stats(sub):
    f = sorted(glob.glob(os.path.join(RAW, sub, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    lines = ["%-44s %7s %12s %11s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for r in rows:
        lines.append("%-44s %7s %12.1f %11.2f %8s" % (short(r["Name"])[:44], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                     float(r["AverageNs"]) / 1e3, r["Percentage"][:6]))
    return rows, "\n".join(lines)


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


def pmc(sub, counter, nimg):
    f = sorted(glob.glob(os.path.join(RAW, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = agg.setdefault(short(r["Kernel_Name"]), [0.0, 0])
        k[0] += float(r["Counter_Value"])
        k[1] += 1
    return agg


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    _, txt = stats("bench")
    bench_line = [l for l in open(os.path.join(RAW, "bench.log")).read().splitlines() if l.startswith('{"metric"')][-1]
    open(os.path.join(OUT, "%s_bench_kernel_stats.txt" % TAG), "w").write(
        "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline\n"
        "# (16 contexts in flight, 1 MI355X; includes warm-up, the latency and the profile-mode passes)\n"
        + txt + "\n\n# bench.py output of this profiled run:\n" + bench_line + "\n")
    _, txt = stats("roofline")
    rl = [l for l in open(os.path.join(RAW, "roofline.log")).read().splitlines() if l.startswith('{"metric"')][-1]
    open(os.path.join(OUT, "%s_roofline_pass_kernel_stats.txt" % TAG), "w").write(
        "# rocprofv3 --kernel-trace --stats -- python3 bench.py --only-roofline\n"
        "# (one context: 1 timed image, 5 latency images, 5 profile-mode images in which every blur launch\n"
        "#  is issued 4x back to back between one HIP event pair; compare roofline.avg_launch_us below with the\n"
        "#  avg_us of the k_blur_tile<*, 0, 64, *> rows)\n" + txt + "\n\n# bench.py --only-roofline output:\n" + rl + "\n")
    rows, txt = stats("single")
    open(os.path.join(OUT, "%s_single_image_kernel_stats.txt" % TAG), "w").write(
        "# rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py 5   (one context, 5 x config-2 image)\n" + txt +
        "\n\n# per-launch timeline of the last image\n" + timeline("single", 5) + "\n")
    fetch, write = pmc("pmc_fetch", "FETCH_SIZE", 3), pmc("pmc_write", "WRITE_SIZE", 3)
    lines = ["# HBM-side traffic per launch from rocprofv3 PMC passes (separate runs for FETCH_SIZE and WRITE_SIZE).",
             "# Units: KiB as reported; bytes = KiB * 1024; FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request,",
             "# MI355X_MICROARCH.md 'HBM').  3 images per run; values are averages per launch.",
             "%-36s %9s %14s %14s %14s" % ("kernel", "launches", "fetch_MB(x2)", "write_MB", "traffic_MB")]
    traffic = {}
    for k in fetch:
        n = fetch[k][1]
        fb = 2.0 * fetch[k][0] / n * 1024 / 1e6
        wb = write.get(k, [0, 1])[0] / max(write.get(k, [0, 1])[1], 1) * 1024 / 1e6
        traffic[k] = {"launches": n, "fetch_bytes": fb * 1e6, "write_bytes": wb * 1e6, "traffic_bytes": (fb + wb) * 1e6}
        lines.append("%-36s %9d %14.2f %14.2f %14.2f" % (k[:36], n, fb, wb, fb + wb))
    open(os.path.join(OUT, "%s_hbm_traffic.txt" % TAG), "w").write("\n".join(lines) + "\n")
    big = [v for k, v in traffic.items() if k.startswith("k_blur_tile<") and ", 0, 64" in k]
    if big:
        per_launch = sum(v["traffic_bytes"] * v["launches"] for v in big) / sum(v["launches"] for v in big)
        json.dump({"kernel": "k_blur_tile<HALO,0,64,*>", "config": "1920x1080 u8, default Config (octave 0 level launches)",
                   "traffic_bytes_per_launch": per_launch, "source": "%s_hbm_traffic.txt" % TAG},
                  open(os.path.join(OUT, "blur_traffic.json"), "w"), indent=1)
    print(open(os.path.join(OUT, "%s_hbm_traffic.txt" % TAG)).read())
