#!/usr/bin/env python
"""Measure the HBM-side traffic of the MSDA launches bench.py times and write profiles/<round>_pmc_traffic.json (SEMIDETR_ROUND, default r06).

Run on the GPU box:   python tools/measure_traffic.py
For each event group of the bench step (encoder / decoder, forward / backward, the batch sizes of the step) it runs
`tools/msda_probe.py` twice under rocprofv3 -- one `--pmc FETCH_SIZE` pass and one `--pmc WRITE_SIZE` pass,
`--kernel-trace` only (MI355X_MICROARCH.md, HBM section: separate passes; counter unit KiB; FETCH_SIZE reports half
the bytes of 16-byte coalesced reads on gfx950 -> x2; WRITE_SIZE uncorrected) -- and stores per-launch averages per
kernel.  The kernel names are matched against what the LIBRARY says it launched for that shape
(semidetr_msda_last_kernels, printed by the probe): if a kernel the library names does not show up in the trace (or
the other way round) the script FAILS, so a renamed / re-dispatched kernel can never leave a stale traffic figure
behind.  bench.py only uses an entry whose kernel list equals the one it observes in its own run.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "traffic")
RAWM = ["--io", "raw", "--masked"]      # round 6: the headline step = fused prologue + padding mask on bench.py's own inputs (bench.Workload)
GROUPS = [  # bench group name, probe arguments
    # the encoder launches as the adaptive policy runs them at the bench's sample spread: region-window forward, window gather + region scatter
    ("msda_fwd_enc_bs4_Lq22223", ["--shape", "enc", "--bs", "4", "--dir", "fwd", "--policy", "window"] + RAWM),
    ("msda_bwd_enc_bs4_Lq22223", ["--shape", "enc", "--bs", "4", "--dir", "bwd", "--policy", "window"] + RAWM),
    ("msda_fwd_enc_bs1_Lq22223", ["--shape", "enc", "--bs", "1", "--dir", "fwd", "--policy", "window"] + RAWM),
    ("msda_bwd_enc_bs1_Lq22223", ["--shape", "enc", "--bs", "1", "--dir", "bwd", "--policy", "window"] + RAWM),
    # ... and through the patch kernels (what a call site whose offsets have grown runs)
    ("msda_fwd_enc_bs4_Lq22223_patch", ["--shape", "enc", "--bs", "4", "--dir", "fwd", "--policy", "patch"] + RAWM),
    ("msda_bwd_enc_bs4_Lq22223_patch", ["--shape", "enc", "--bs", "4", "--dir", "bwd", "--policy", "patch"] + RAWM),
    # the reference op contract (no prologue, no mask) on the probe's own inputs, as rounds 1-5 measured it
    ("msda_fwd_enc_bs4_Lq22223_locattn", ["--shape", "enc", "--bs", "4", "--dir", "fwd", "--policy", "window"]),
    ("msda_bwd_enc_bs4_Lq22223_locattn", ["--shape", "enc", "--bs", "4", "--dir", "bwd", "--policy", "window"]),
    ("msda_fwd_dec_bs4_Lq1100", ["--shape", "dec", "--bs", "4", "--lq", "1100", "--dir", "fwd"] + RAWM),
    ("msda_bwd_dec_bs4_Lq1100", ["--shape", "dec", "--bs", "4", "--lq", "1100", "--dir", "bwd"] + RAWM),
    ("msda_bwd_dec_bs1_Lq1100", ["--shape", "dec", "--bs", "1", "--lq", "1100", "--dir", "bwd"] + RAWM),
    ("msda_fwd_micro_bs2_Lq300", ["--shape", "micro", "--bs", "2", "--dir", "fwd"]),
    ("msda_bwd_micro_bs2_Lq300", ["--shape", "micro", "--bs", "2", "--dir", "bwd"]),
    # the same shape with 8 input sets rotated (> 256 MB: nothing is Infinity-Cache resident), as bench.py's cold micro-benchmark
    ("msda_fwd_micro_bs2_Lq300_cold", ["--shape", "micro", "--bs", "2", "--dir", "fwd", "--cold", "8", "--iters", "16"]),
    ("msda_bwd_micro_bs2_Lq300_cold", ["--shape", "micro", "--bs", "2", "--dir", "bwd", "--cold", "8", "--iters", "16"]),
]


def short(name):
    """'void (anonymous namespace)::msda_fwd_d32<1, 4, 408, (anonymous namespace)::LocAttnIO>(...)' -> 'msda_fwd_d32<1, 4, 408, LocAttnIO>'"""
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    depth = 0
    for i, ch in enumerate(n):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return n[:i]
    return n


def run_pass(counter, args, tag):
    d = os.path.join(OUT, tag + "_" + counter)
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "tools", "msda_probe.py"), "--iters", "3", "--print-kernels"] + args      # a later --iters wins
    env = dict(os.environ, TMPDIR="/tmp", SEMIDETR_EXPERIMENTS="0")      # the PRODUCT library is what is measured
    p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        raise SystemExit("rocprofv3 failed for %s %s:\n%s" % (tag, counter, p.stderr[-2000:]))
    reported = [ln.split("=", 1)[1].strip() for ln in p.stdout.splitlines() if ln.startswith("KERNELS=")]
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return reported[-1].split("+") if reported else [], {k: sum(v) / len(v) for k, v in vals.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    res = {"_source": "tools/measure_traffic.py on MI355X: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, "
                      "--kernel-trace only) of tools/msda_probe.py per event group; per-launch averages; counter unit KiB; "
                      "hbm_bytes_corrected = 2 * fetch + write (gfx950: FETCH_SIZE counts 16-byte coalesced reads at half "
                      "their size; the factor is calibrated for streaming reads and only assumed for the corner gathers)."}
    for group, args in GROUPS:
        rep_f, fetch = run_pass("FETCH_SIZE", args, group)
        rep_w, write = run_pass("WRITE_SIZE", args, group)
        assert rep_f == rep_w and rep_f, (group, rep_f, rep_w)
        kernels, total = {}, 0.0
        for want in rep_f:       # every kernel the library names must be in the trace exactly once (as a prefix)
            hits = [k for k in fetch if want in k]
            if len(hits) != 1:
                raise SystemExit("%s: library reports kernel %r, trace has %r -- fix semidetr_msda_last_kernels or this "
                                 "script before trusting any traffic number" % (group, want, sorted(fetch)))
            k = hits[0]
            kernels[k] = {"fetch_kib_raw": fetch[k], "write_kib": write.get(k, 0.0)}
            total += (2 * fetch[k] + write.get(k, 0.0)) * 1024
        # (msda_mask_extents_kernel: the once-per-mask summary launch of the fused + masked calls, ~3 us, cached by the front end: not part of the op)
        stray = [k for k in fetch if "msda_" in k and k not in kernels and "msda_mask_extents_kernel" not in k]
        if stray:
            raise SystemExit("%s: msda kernels in the trace the library did not report: %r" % (group, stray))
        res[group] = {"kernels": rep_f, "per_kernel": kernels, "hbm_bytes_corrected": int(total)}
        print(group, rep_f, "%.1f MB" % (total / 1e6), flush=True)
    with open(os.path.join(ROOT, "gpurun_out", os.environ.get("SEMIDETR_ROUND", "r06") + "_pmc_traffic.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
