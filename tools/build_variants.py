#!/usr/bin/env python3
"""Here (no GPU): build experiment variants of libpopsift_hip into build_variants/vN.so (they travel to the GPU box and
are loaded through POPSIFT_HIP_LIB; the product library is never overwritten).

    tools/build_variants.py "<flags of v1>" "<flags of v2>" ...

Each argument is split on spaces (shell-quoted by the caller, so parentheses in -D values are fine) and added to the
Makefile's own flags for EVERY source file.  Only files whose text mentions a macro named in the flags are rebuilt; the
others are taken from the product build's objects.
"""
import os
import re
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "popsift_amd", "csrc")
OUT = os.path.join(R, "build_variants")
BASE = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
        "-Wno-unused-function"]


def main():
    os.makedirs(OUT, exist_ok=True)
    for f in os.listdir(OUT):
        if f.endswith(".so") or f == "flags.txt":
            os.remove(os.path.join(OUT, f))
    subprocess.check_call(["make", "-s"], cwd=C)
    mk = open(os.path.join(C, "Makefile")).read()
    srcs = re.search(r"^SRCS\s*=\s*(.*)$", mk, re.M).group(1).split()
    with open(os.path.join(OUT, "flags.txt"), "w") as fl:
        for i, v in enumerate(sys.argv[1:], 1):
            flags = v.split()
            macros = [re.match(r"-D([A-Za-z0-9_]+)", x).group(1) for x in flags if x.startswith("-D")]
            objs = []
            for s in srcs:
                b = s[:-4]
                text = open(os.path.join(C, s)).read()
                if macros and not any(m in text for m in macros):
                    objs.append(os.path.join(C, b + ".o"))
                    continue
                per = re.search(r"^FLAGS_%s\s*=\s*(.*)$" % b, mk, re.M)
                o = "/tmp/var%d_%s.o" % (i, b)
                subprocess.check_call(BASE + (per.group(1).split() if per else []) + flags + ["-c", s, "-o", o], cwd=C)
                objs.append(o)
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                                   os.path.join(OUT, "v%d.so" % i)] + objs)
            fl.write("v%d: %s\n" % (i, v))
            print("v%d: %s" % (i, v))


if __name__ == "__main__":
    main()
