"""Per-kernel register / spill / LDS table of the product build, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python tools/kernel_resources.py [--log BUILD_STDERR] [--check PATTERN ...]   (--check: exit 1 if a kernel matching PATTERN spills)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(log=None):
    if log is None:
        csrc = os.path.join(ROOT, "ctransformers_amd", "csrc")
        out = subprocess.run(["make", "-B", "OUT=/tmp/ctamd_res_build", "EXTRA=-Rpass-analysis=kernel-resource-usage"], cwd=csrc,
                             capture_output=True, text=True)
        if out.returncode != 0:
            sys.stderr.write(out.stderr[-4000:])
            raise SystemExit("build failed")
        text = out.stderr
    else:
        text = open(log).read()
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return kernels


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def main():
    pats = []
    if "--check" in sys.argv:
        pats = [a for a in sys.argv[sys.argv.index("--check") + 1:] if not a.startswith("--")]
    k = collect(sys.argv[sys.argv.index("--log") + 1] if "--log" in sys.argv else None)
    names = sorted(k)
    bad = []
    print("%-100s %5s %5s %6s %7s %4s" % ("kernel", "VGPR", "SGPR", "spill", "scratch", "occ"))
    for n, d in zip(names, demangle(names)):
        r = k[n]
        d = d.replace("(MatvecArgs)", "").replace("(PfArgs)", "")
        print("%-100s %5d %5d %6d %7d %4d" % (d[:100], r.get("VGPRs", -1), r.get("SGPRs", r.get("SGPRs ", -1)), r.get("VGPRs Spill", -1),
                                               r.get("ScratchSize", -1), r.get("Occupancy", -1)))
        if r.get("VGPRs Spill", 0) > 0 and any(re.search(p, d) for p in pats):
            bad.append(d)
    if bad:
        print("SPILLS in checked kernels:\n  " + "\n  ".join(bad))
        raise SystemExit(1)


if __name__ == "__main__":
    main()
