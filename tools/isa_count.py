#!/usr/bin/env python3
"""Instruction budget of a kernel's hot loop from the compiler's own ISA (no GPU needed).

    tools/isa_count.py [--kernel k_descriptor] [--marker ds_add_u64] [--samples 2] [extra hipcc flags ...]

Compiles popsift_amd/csrc/keypoint.hip for gfx950 with the Makefile's flags plus -gline-tables-only, takes the
innermost-but-one loop of <kernel> that contains <marker>, and counts its instructions
  - by kind: VALU (full rate), VALU quarter rate (transcendentals, 32-bit integer multiply), SALU, LDS, VMEM, waits;
  - by source group: the `.loc` line of every instruction is looked up in the `/* ISA: <group> */` markers of the
    source (a marker opens a group that lasts until the next marker or the end of the enclosing lambda / block given
    by `/* ISA: end */`).
--samples N: the loop body handles N samples (software-pipelined unroll), counts are printed per sample.
"""
import collections
import os
import re
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(R, "popsift_amd", "csrc", "keypoint.hip")
# Issue cost classes measured on MI355X (tools/ubench/valu_rate.hip, cycles per wave64 instruction with 8 waves on a SIMD):
#   full rate 1.34 -- f32 add / sub / mul / FMA (also with a literal), 32-bit add / sub, and / or / xor, arithmetic shift right, mov
#   half rate 2.36 -- compares, v_cndmask, conversions, floor / fract / trunc, min / max / max3, v_bfi, shifts left, v_lshl_add /
#                     v_add_lshl / v_lshl_or / v_add3, 24- and 32-bit multiplies and multiply-adds
#   transcendental 4.64 -- sqrt, rcp, rsq, exp, log, sin, cos
TRANS = ("v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32")
FULL = ("v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_ashrrev_i32", "v_lshrrev_b32", "v_mov_b32",
        "v_not_b32")
COST = {"valu": 1.0, "valu_half": 2.36 / 1.34, "valu_quarter": 4.64 / 1.34}


def source_groups():
    g, cur = {}, "other"
    for i, line in enumerate(open(SRC), 1):
        m = re.search(r"/\* ISA: ([a-z0-9_+ -]+) \*/", line)
        if m:
            cur = "other" if m.group(1) == "end" else m.group(1)
        g[i] = cur
    return g


def main():
    a = sys.argv[1:]
    kernel, marker, samples = "k_descriptor", "ds_add_u64", 2
    extra = []
    i = 0
    while i < len(a):
        if a[i] == "--kernel":
            kernel = a[i + 1]; i += 2
        elif a[i] == "--marker":
            marker = a[i + 1]; i += 2
        elif a[i] == "--samples":
            samples = int(a[i + 1]); i += 2
        else:
            extra.append(a[i]); i += 1
    out = "/tmp/isa_count.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-fno-slp-vectorize", "-Wno-unused-function", "-gline-tables-only", "-S", "--cuda-device-only",
                           SRC, "-o", out] + extra, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    # the function body
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN.*%s[A-Z].*:\s*(;.*)?$" % kernel, l) or
                 re.match(r"^_ZN\S*\d+%sE\S*:" % kernel, l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip() == "s_endpgm")
    body = lines[start:end]
    # basic blocks with their loop annotation
    hdr_of = {}
    cur = None
    blocks = collections.OrderedDict()
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)$", l)
        if m:
            cur = m.group(1)
            blocks[cur] = {"note": m.group(2), "ins": []}
            continue
        m2 = re.match(r"^; %bb\.(\d+):\s*;?(.*)$", l)
        if m2:
            cur = "bb." + m2.group(1)
            blocks[cur] = {"note": m2.group(2), "ins": []}
            continue
        if cur is None:
            continue
        if l.strip().startswith(";") and ("Loop" in l):
            blocks[cur]["note"] += " " + l
            continue
        blocks[cur]["ins"].append(l)
    # loop headers: label -> depth
    def header_of(name, b):
        m = re.search(r"Header=BB(\d+_\d+) Depth=(\d+)", b["note"])
        if m:
            return ".LBB" + m.group(1), int(m.group(2))
        m = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", b["note"])
        if m:
            return name, int(m.group(1))
        return None, 0
    marked = [n for n, b in blocks.items() if any(marker in x for x in b["ins"])]
    if not marked:
        sys.exit("marker %s not found in %s" % (marker, kernel))
    hdr, depth = header_of(marked[0], blocks[marked[0]])
    # all blocks of that loop (including deeper child loops whose parent chain mentions the header)
    member = []
    for n, b in blocks.items():
        h, dp = header_of(n, b)
        if h == hdr or ("Parent Loop BB%s " % hdr[4:]) in b["note"] or n == hdr:
            member.append(n)
    groups = source_groups()
    kp_files = set()
    for l in lines:
        m = re.match(r"\s*\.file\s+(\d+)\s+(.*)$", l)
        if m and "keypoint.hip" in m.group(2):
            kp_files.add(int(m.group(1)))
    kinds = collections.Counter()
    bygroup = collections.defaultdict(collections.Counter)
    loc = 0
    listing = []
    for n in member:
        for l in blocks[n]["ins"]:
            t = l.strip()
            m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
            if m:
                # instructions inlined from headers (fmaf, floorf, ...) and line-0 glue stay with the last source line
                if int(m.group(1)) in kp_files and int(m.group(2)) > 0:
                    loc = int(m.group(2))
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            op = t.split()[0]
            if op.startswith("v_"):
                base = op[:-4] if op.endswith(("_e32", "_e64")) else op
                base = base.replace("_sdwa", "").replace("_dpp", "")
                k = "valu_quarter" if base in TRANS else ("valu" if base in FULL else "valu_half")
            elif op.startswith("s_waitcnt") or op.startswith("s_nop"):
                k = "wait_nop"
            elif op.startswith("s_"):
                k = "salu"
            elif op.startswith("ds_"):
                k = "lds"
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                k = "vmem"
            else:
                k = "other"
            kinds[k] += 1
            bygroup[groups.get(loc, "other")][k] += 1
            listing.append((n, loc, groups.get(loc, "other"), t))
    vg = [l for l in lines[end:end + 400] if re.search(r"; (NumVgprs|Occupancy|ScratchSize|LDSByteSize)", l)][:4]
    print("# %s: loop %s (depth %d), %d blocks, body handles %d sample(s); flags: %s" %
          (kernel, hdr, depth, len(member), samples, " ".join(extra) or "(Makefile)"))
    print("# " + "  ".join(x.strip("; ").strip() for x in vg))
    tot_v = kinds["valu"] + kinds["valu_half"] + kinds["valu_quarter"]
    cost = sum(kinds[k] * c for k, c in COST.items())
    print("per sample: VALU %.1f = %.1f full-rate + %.1f half-rate + %.1f transcendental => %.1f full-rate issue slots;  SALU %.1f  LDS %.1f  VMEM %.1f  waits/nops %.1f"
          % (tot_v / samples, kinds["valu"] / samples, kinds["valu_half"] / samples, kinds["valu_quarter"] / samples, cost / samples,
             kinds["salu"] / samples, kinds["lds"] / samples, kinds["vmem"] / samples, kinds["wait_nop"] / samples))
    print("\n%-22s %8s %8s %8s %8s %8s %8s %8s" % ("group (per sample)", "VALU", "half", "transc", "slots", "SALU", "LDS", "VMEM"))
    tv = lambda c: c["valu"] + c["valu_half"] + c["valu_quarter"]
    for g in sorted(bygroup, key=lambda g: -tv(bygroup[g])):
        c = bygroup[g]
        print("%-22s %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f" % (g, tv(c) / samples, c["valu_half"] / samples, c["valu_quarter"] / samples,
                                                                 sum(c[k] * w for k, w in COST.items()) / samples,
                                                                 c["salu"] / samples, c["lds"] / samples, c["vmem"] / samples))
    if os.environ.get("ISA_LIST"):
        print()
        for n, lc, g, t in listing:
            print("%-10s %5d %-18s %s" % (n, lc, g, t))


if __name__ == "__main__":
    main()
