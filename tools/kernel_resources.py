"""Per-kernel register / LDS / occupancy table of one .hip file, from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(build host; no GPU).  usage: python tools/kernel_resources.py spex_amd/csrc/spmm.hip [substring] [-DNAME ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(src, extra=()):
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off".split()
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + list(extra) + ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        os.path.abspath(src), "-o", "/dev/null"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(src)))
    out, cur = [], None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark: \s*Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            out.append(cur)
            continue
        m = re.search(r"remark: \s*([A-Za-z \[\]/]+): (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


if __name__ == "__main__":
    src = sys.argv[1]
    sub = [a for a in sys.argv[2:] if not a.startswith("-")]
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    for k in resources(src, extra):
        if sub and not any(s in k["name"] for s in sub):
            continue
        name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        print("%-60s VGPR %3d AGPR %3d SGPR %3d scratch %4d LDS %6d occ %d" % (
            name[:60], k.get("VGPRs", -1), k.get("AGPRs", -1), k.get("TotalSGPRs", -1), k.get("ScratchSize [bytes/lane]", -1),
            k.get("LDS Size [bytes/block]", -1), k.get("Occupancy [waves/SIMD]", -1)))
