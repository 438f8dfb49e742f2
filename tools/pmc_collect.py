#!/usr/bin/env python3
"""Collects rocprofv3 hardware counters for one command in separate passes (one rocprofv3 run per
counter group, as the profiler does not split groups itself) and prints per-kernel sums and a few
derived ratios as JSON.  Development tool.

    python tools/pmc_collect.py OUT_DIR KERNEL_SUBSTRING -- python3 bench/configs.py --only c1 --quick

The profiled program must follow `--` directly (no env/bash wrappers).  PMC_PASSES="A B;C D" replaces the default
counter groups (one pass per `;`-separated group)."""
import csv
import glob
import json
import os
import subprocess
import sys

PASSES = [
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU"],
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VMEM_WR",
     "GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
]


def main():
    global PASSES
    if os.environ.get("PMC_PASSES"):          # other counter groups: "A B C;D E" = two passes
        PASSES = [g.split() for g in os.environ["PMC_PASSES"].split(";") if g.strip()]
    out_dir, needle = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    sums = {}
    launches = 0
    for i, group in enumerate(PASSES):
        d = os.path.join(out_dir, "pass%d" % i)
        os.makedirs(d, exist_ok=True)
        full = ["timeout", "-s", "KILL", "240", "rocprofv3", "--pmc"] + group + ["-d", d, "--output-format", "csv", "--"] + cmd
        r = subprocess.run(full, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print("pass %d failed (%d):\n%s" % (i, r.returncode, r.stdout[-2000:]), flush=True)
            sys.exit(1)            # start nothing more after a failed GPU step
        per = {}
        disp = set()
        spans = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if needle not in row["Kernel_Name"]:
                    continue
                per[row["Counter_Name"]] = per.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                disp.add(row["Dispatch_Id"])
                if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    spans[row["Dispatch_Id"]] = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-6
        launches = max(launches, len(disp))
        if spans and "FETCH_SIZE" in group:      # the kernel's time in the very pass that counted its bytes: bytes / time = GB/s as measured
            sums["_kernel_ms_sum_in_fetch_pass"] = sum(spans.values())
        sums.update(per)
        print("pass %d ok: %s" % (i, sorted(per)), flush=True)
    s = sums
    der = {"launches_summed": launches}
    if "SQ_INSTS_VALU" in s and "SQ_THREAD_CYCLES_VALU" in s:
        der["valu_lane_utilisation"] = s["SQ_THREAD_CYCLES_VALU"] / (64.0 * s["SQ_ACTIVE_INST_VALU"]) if s.get("SQ_ACTIVE_INST_VALU") else None
    if "SQ_WAVE_CYCLES" in s:
        der["wave_cycles_waiting_on_waitcnt_frac"] = s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"]
        der["wave_cycles_waiting_to_issue_frac"] = s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"]
        der["wave_cycles_issuing_frac"] = s["SQ_ACTIVE_INST_ANY"] / s["SQ_WAVE_CYCLES"]
        if "GRBM_GUI_ACTIVE" in s:
            simd_quads = s["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 / 4.0       # SQ counters tick once per 4 cycles
            der["mean_waves_per_simd"] = s["SQ_WAVE_CYCLES"] / simd_quads
            if "SQ_ACTIVE_INST_VALU" in s:
                der["valu_active_quads_per_simd_quad"] = s["SQ_ACTIVE_INST_VALU"] / simd_quads
            if "SQ_INSTS_VALU" in s:
                der["cycles_per_valu_inst_per_simd"] = simd_quads * 4.0 / s["SQ_INSTS_VALU"]
    if "FETCH_SIZE" in s:          # rocprofv3 reports KiB; on gfx950 the fetch counter reads half the bytes moved
        der["fetch_bytes_per_launch_corrected_x2"] = s["FETCH_SIZE"] * 1024.0 * 2.0 / max(1, launches)
    if "WRITE_SIZE" in s:
        der["write_kib_per_launch"] = s["WRITE_SIZE"] / max(1, launches)
    if s.get("_kernel_ms_sum_in_fetch_pass"):
        der["kernel_ms_avg"] = s["_kernel_ms_sum_in_fetch_pass"] / max(1, launches)
    s["_derived"] = der
    # which build was profiled: bench.py only quotes these numbers for the library they were measured on
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpupathtracer_amd", "libptcore.so")
    lib = os.environ.get("PT_LIBPTCORE_PATH") or lib
    if os.path.exists(lib):
        h = hashlib.sha256()
        with open(lib, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
        s["_libptcore_sha256"] = h.hexdigest()
    s["_command"] = " ".join(cmd)
    s["_kernel_substring"] = needle
    with open(os.path.join(out_dir, "summary.json"), "w") as f:
        json.dump(s, f, indent=1)
    print(json.dumps(s, indent=1))


if __name__ == "__main__":
    main()
