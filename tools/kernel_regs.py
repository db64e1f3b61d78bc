#!/usr/bin/env python3
"""Register / spill / LDS report per kernel of one HIP source (device-only -S compile for gfx950)."""
import re, subprocess, sys, os
src = sys.argv[1]
out = os.path.join(os.path.dirname(os.path.abspath(src)), "_build", os.path.basename(src) + ".s")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DNDEBUG",
                "--cuda-device-only", "-S", "-x", "hip", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
for blk in s.split("- .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    print("%-62s vgpr %3s agpr %3s spill %3s lds %6s scratch %4s" % (name[-62:], g("vgpr_count"), blk.split("\n")[0].strip(),
          g("vgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
