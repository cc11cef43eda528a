#!/usr/bin/env python3
"""Rewrites the committed summaries from the raw outputs of profiles/collect.sh <tag> and profiles/collect_mfma.sh <tag2>
(gpurun_out/prof_<tag>, gpurun_out/prof_<tag2>).
    python profiles/refresh.py final mfma_final [tag of profiles/collect_pmc_short.sh]
AZ_PROFILE_ROUND names the files written (default round3: profiles/round3_kernel_stats.txt, ...)."""
import csv
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..")
tag, mtag = sys.argv[1], sys.argv[2]
P = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
PC = os.path.join(ROOT, "gpurun_out", "prof_" + (sys.argv[3] if len(sys.argv) > 3 else tag))   # counter passes (collect_pmc_short.sh)
PM = os.path.join(ROOT, "gpurun_out", "prof_" + mtag)


def run(*a):
    return subprocess.check_output([sys.executable] + list(a), text=True)


def val(txt, pat):
    for line in txt.splitlines():
        if pat in line:
            return float(line.split()[-1])
    raise KeyError(pat)


# kernel trace
ROUND = os.environ.get("AZ_PROFILE_ROUND", "round3")
hdr = ("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tower-parity --no-exchange   (MI355X, 3993 games per GPU, "
       "41 rounds per step; the 3x3 256->256 convolutions run the two-workgroups-per-CU kernel k_conv3x3_c256_bf16_duo, one launch per layer and "
       "group of 363 boards; collected by "
       "profiles/collect.sh)\n")
body = run(os.path.join(HERE, "rocpd_summary.py"), os.path.join(P, "trace", "trace_results.db"))
bench = open(os.path.join(P, "trace_bench.json")).read().strip().splitlines()[-1]
open(os.path.join(HERE, ROUND + "_kernel_stats.txt"), "w").write(hdr + body + "\n# bench line of the same run:\n" + bench + "\n")

# HBM traffic.  The tower takes the batch through its layers in groups of boards, so a launch covers one group; the number of
# boards per launch comes from the bench line of the same run.
if not os.path.exists(os.path.join(PC, "pmc_write", "write_counter_collection.csv")):
    print("no counter passes under", PC, "- traffic summary left as it is")
    sys.exit(0)
bl = json.loads(bench)
boards = bl["roofline"]["boards_per_launch"]
f = run(os.path.join(HERE, "pmc_summary.py"), os.path.join(PC, "pmc_fetch", "fetch_counter_collection.csv"))
w = run(os.path.join(HERE, "pmc_summary.py"), os.path.join(PC, "pmc_write", "write_counter_collection.csv"))
fp, fr, ws = val(f, "duo<false"), val(f, "duo<true"), val(w, "duo<true")
rd, wr = 2 * (fp + fr) / 2 * 1024 / 1e6, ws * 1024 / 1e6
a_in, a_out = boards * 441 * 512 / 1e6, boards * 361 * 512 / 1e6
hdr2 = ("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 0 --rounds-per-step 2 --no-cpu-baseline --no-tower-parity --no-exchange   "
        "(profiles/collect_pmc_short.sh: the counter passes abort inside the profiler on longer runs of this many dispatches; "
        "3993 games per GPU, the tower in groups of %.0f boards)\n# counter unit: KiB per dispatch (mean).  On gfx950 FETCH_SIZE counts 128-B "
        "read requests as 64 B: double it for bytes.\n# The counters sit between L2 and the fabric: they count what leaves the XCDs' L2s, "
        "whether the 256 MB memory-side cache or HBM itself serves it.\n" % boards)
tail = """
# conv 3x3 256->256, %.0f boards per launch (mean of the plain and the residual variant, 20 launches each per group and tower pass):
#   read  = 2 * (%.1f + %.1f)/2 KiB = %.1f MB ; write = %.1f KiB = %.1f MB ; total %.1f MB
#   algorithmic: activations in %.1f MB (padded NHWC bf16) + weights 1.2 MB + out %.1f MB (+%.1f MB residual read on every second launch) = %.1f MB mean
#   read amplification %.2fx (plain) / %.2fx (residual): halo rows of neighbouring tiles, and the second channel half of a tile
#   (its own workgroup, same XCD) re-reading the tile's activation slab when it has left L2
""" % (boards, fp, fr, rd, ws, wr, rd + wr, a_in, a_out, a_out, a_in + 1.2 + 1.5 * a_out, 2 * fp * 1024 / 1e6 / (a_in + 1.2),
       2 * fr * 1024 / 1e6 / (a_in + 1.2 + a_out))
open(os.path.join(HERE, ROUND + "_pmc_traffic.txt"), "w").write(hdr2 + f + "\n" + w + tail)
jp = os.path.join(HERE, ROUND + "_traffic.json")
j = {"kernel": "k_conv3x3_c256_bf16_duo", "boards": boards, "board": 19, "channels": 256, "fetch_correction": 2.0,
     "unit_note": "counter values are KiB per dispatch",
     "source": "profiles/%s_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE, separate passes; FETCH_SIZE doubled for gfx950)" % ROUND}
j.update(fetch_size_kib_plain=fp, fetch_size_kib_resid=fr, write_size_kib=ws, traffic_bytes_per_launch=int(round((rd + wr) * 1e6)))
json.dump(j, open(jp, "w"), indent=1)
print("%.0f boards per launch: read %.1f MB write %.1f MB" % (boards, rd, wr))
if mtag == "-":
    sys.exit(0)

# MFMA counters of the harness
m = run(os.path.join(HERE, "pmc_summary.py"), os.path.join(PM, "pmc", "mfma_counter_collection.csv"))
rows = [line for line in m.splitlines() if "fillBuffer" not in line]
g, busy = val(m, "GRBM_GUI_ACTIVE"), val(m, "SQ_VALU_MFMA_BUSY_CYCLES")
wavec, wi, wa = val(m, "SQ_WAVE_CYCLES"), val(m, "SQ_WAIT_INST_ANY"), val(m, "SQ_WAIT_ANY")
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(os.path.join(PM, "trace", "trace_kernel_trace.csv")))
     if "slab" in r["Kernel_Name"]]
ms = sum(d) / len(d) / 1e6
cyc, per_simd = g / 8, busy / 1024
util = per_simd / cyc
runline = [line for line in open(os.path.join(PM, "run.log")) if line.startswith("variant")][-1].strip()
hdr3 = ("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE "
        "-- tools/conv_bench 4096 10 5 1 0\n# (standalone harness: the residual variant of the slab conv kernel, 4096 boards of 19x19, uniform random "
        "data; collected by profiles/collect_mfma.sh)\n")
tail3 = """
# per launch: 1744.3 GFLOP = 106.46 M v_mfma_f32_16x16x32_bf16 x 16 cycles = 1.7034e9 matrix-pipe cycles (the counter reads exactly that)
# GRBM_GUI_ACTIVE %.3f M is the sum over 8 XCDs -> %.3f M shader cycles per launch; 1024 SIMDs -> %.4f M busy cycles per SIMD
# MFMA pipe utilisation = %.4f / %.3f = %.1f %% of the kernel's cycles
# kernel-trace pass of the same harness: %.3f ms per launch -> effective clock %.2f GHz -> %.1f %% of the 2.5 PFLOP/s nominal peak
# harness line of the trace pass: %s
# (random data; the self-play bench's post-ReLU activations clock higher: 1240-1290 TFLOP/s)
# SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.0f %% (issue stalls: MFMA dependency / pipe), SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.0f %% (waitcnt + barrier)
""" % (g / 1e6, cyc / 1e6, per_simd / 1e6, per_simd / 1e6, cyc / 1e6, 100 * util, ms, cyc / ms / 1e6, 100 * util * (cyc / ms / 1e6) / 2.4,
       runline, 100 * wi / wavec, 100 * wa / wavec)
open(os.path.join(HERE, ROUND + "_pmc_mfma.txt"), "w").write(hdr3 + "\n".join(rows) + "\n" + tail3)
print("read %.1f MB write %.1f MB per launch; MFMA pipe %.1f %%, %.2f GHz" % (rd, wr, 100 * util, cyc / ms / 1e6))
