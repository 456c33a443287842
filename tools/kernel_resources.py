"""Per-kernel register / LDS / occupancy table of the product's device code (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py [-DSSF_EXPERIMENTS ...] > profiles/kernel_resources_rNN.txt

Also prints the number of 256-thread workgroups a CU admits by scalar registers (MI355X_MICROARCH.md, "Residency":
min(8, floor(800 / (ceil(sgpr / 16) * 16 + 16)))) -- the API's occupancy answer is one too high for 81-96 / 97-112."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "supersurfel_fusion_amd", "csrc")
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "
         "-fhip-fp32-correctly-rounded-divide-sqrt -Rpass-analysis=kernel-resource-usage").split()


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), text=True, stdout=subprocess.PIPE)
    return r.stdout.split("\n")


def main():
    extra = sys.argv[1:]
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("ssf_extract.hip", "ssf_track_fuse.hip"):
            r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-c", src, "-o", os.path.join(tmp, src + ".o")],
                               cwd=CSRC, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
            for blk in re.split(r"remark: Function Name: ", r.stderr)[1:]:
                name = blk.split("\n")[0].strip()

                def g(key):
                    m = re.search(re.escape(key) + r": (\d+)", blk)
                    return int(m.group(1)) if m else -1
                rows.append((src, name, g("SGPRs"), g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"),
                             g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
    names = demangle([r[1] for r in rows])
    print("%-78s %5s %5s %5s %7s %4s %7s %9s" % ("kernel", "sgpr", "vgpr", "agpr", "scratch", "occ", "lds", "wg/CU(sgpr)"))
    for r, n in zip(rows, names):
        n = re.sub(r"^void ssf::", "", n)
        n = re.sub(r"\(.*$", "", n)
        sg = r[2]
        by_sgpr = min(8, 800 // (((sg + 15) // 16) * 16 + 16)) if sg >= 0 else -1
        print("%-78s %5d %5d %5d %7d %4d %7d %9d" % (n[:78], r[2], r[3], r[4], r[5], r[6], r[7], by_sgpr))


if __name__ == "__main__":
    main()
