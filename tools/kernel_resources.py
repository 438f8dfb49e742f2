#!/usr/bin/env python3
"""Per-kernel register, spill and scratch figures of a HIP source, from the compiler's own
kernel-resource-usage remarks (compiles the file for gfx950 with the library's flags; no GPU needed).

    python tools/kernel_resources.py gpupathtracer_amd/csrc/pt_k_cornell.hip [out.tsv]

Extra compiler flags (kernel experiments) come from the environment: PT_EXTRA_FLAGS="-DPT_EXP_BLOCKS=4".
"""
import os
import re
import subprocess
import sys

FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wno-unused-function"]


def main():
    src = sys.argv[1]
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + os.environ.get("PT_EXTRA_FLAGS", "").split() + ["-c", "-o", "/dev/null", src, "-Rpass-analysis=kernel-resource-usage"],
                       stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-3000:])
        sys.exit(1)
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    head = ("kernel", "VGPRs", "AGPRs", "SGPRs", "scratch_B_per_lane", "waves_per_SIMD", "SGPR_spills", "VGPR_spills", "LDS_B_per_block")
    keys = ("VGPRs", "AGPRs", "SGPRs", r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]", "SGPRs Spill", "VGPRs Spill", r"LDS Size \[bytes/block\]")
    lines = ["\t".join(head)]
    for b, d in zip(blocks, dem):
        vals = []
        for k in keys:
            m = re.search(k + r": (\d+)", b)
            vals.append(m.group(1) if m else "-")
        d = re.sub(r"\(ptd::SceneView.*", "", d).replace("void ", "")
        lines.append("\t".join([d] + vals))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
