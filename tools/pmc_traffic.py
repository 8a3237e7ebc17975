#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately as
MI355X_MICROARCH.md prescribes) -> profiles/<tag>_pmc_traffic.json, which bench.py reads for roofline.traffic.

usage: pmc_traffic.py <fetch.db> <write.db> <out.json> <steps in the profiled run, warm-up included> [note]
       pmc_traffic.py --regroup <in.json> <out.json> <steps>      (re-aggregate the per-kernel table of an earlier run)

Families are bench.py's profiling families (one launch = one library call, which may be two kernels: split-K
convolution + its finalize, n-split weight gradient + its reduce), so the family figure is HBM bytes per STEP and
bench.py divides it by the launches per step it counted itself.

Corrections applied (same guide, section HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes
of 16 B/lane streaming reads, which is what every kernel here issues (global_load_dwordx4 / 16-byte LDS-DMA), so it
is doubled.  Calibration points inside these very runs: final_bwd_data_kernel writes 192 planes = 491 520 KiB and
WRITE_SIZE says 491 520.0; final_fwd_kernel reads the same 192 planes and FETCH_SIZE says 245 778 (x2 = 491 556)."""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_id import source_id          # noqa: E402 -- ties the file to the kernel sources it was measured on (bench.py checks it)

FAMILIES = [
    # (the bf16-storage family's kernels, bench.py --config 2: bf16_dgrad_block_kernel and bf16_conv_kernel<3, 3, 2, ...> serve only the
    # dense layers' data gradients; bf16_wgrad_kernel<3> also the five transition-up and the first convolution -- 6 of 50 launches)
    ("dgrad_dense", re.compile(r"dgrad_block8?_kernel|dgrad_wino8_kernel|dgrad_wino3_kernel|dgrad_dense_kernel|conv_dma_kernel<3, \d+, 1, 0, 2,|"
                               r"bf16_dgrad_block_kernel|bf16_conv_kernel<3, 3, 2,")),
    ("conv3x3_dense_fwd", re.compile(r"wino_fwd_kernel|conv_dma_kernel<3, \d+, 1, 1, 0,|finalize_partial_kernel|bf16_conv_kernel<3, 1, 0,")),
    ("wgrad_dense", re.compile(r"wgrad_nsplit_kernel|wgrad_nsplit_reduce_kernel|wgrad_taps_kernel<12, 1>|wgrad_mfma_kernel<3, 1, 1, 0>|"
                               r"bf16_wgrad_kernel<3>|bf16_wgrad_reduce_kernel")),
]


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                       "group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def families_from_kernels(kernels, steps):
    fams = {}
    for fam, rx in FAMILIES:
        sel = [v for k, v in kernels.items() if rx.search(k)]
        if sel:
            fetch = sum(v["fetch_bytes_per_launch"] * v["launches"] for v in sel)
            write = sum(v["write_bytes_per_launch"] * v["launches"] for v in sel)
            fams[fam] = {"kernel_dispatches_per_step": sum(v["launches"] for v in sel) / steps,
                         "fetch_bytes_per_step": fetch / steps, "write_bytes_per_step": write / steps,
                         "traffic_bytes_per_step": (fetch + write) / steps}
    return fams


def main():
    if sys.argv[1] == "--regroup":
        src = json.load(open(sys.argv[2]))
        steps = int(sys.argv[4])
        src["families"] = families_from_kernels(src["kernels"], steps)
        src["steps"] = steps
        json.dump(src, open(sys.argv[3], "w"), indent=1, sort_keys=True)
        print(json.dumps(src["families"], indent=1))
        return
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    steps = int(sys.argv[4])
    out = {"note": sys.argv[5] if len(sys.argv) > 5 else "", "steps": steps, "source": source_id(),
           "unit": "bytes (2*FETCH_SIZE + WRITE_SIZE, KiB -> B); families per step, kernels per dispatch",
           "families": {}, "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        if "endo::" not in name:
            continue
        nf, f = fetch.get(name, (0, 0.0))
        nw, w = write.get(name, (0, 0.0))
        out["kernels"][name[:110]] = {"launches": max(nf, nw), "fetch_bytes_per_launch": 2048.0 * f / max(nf, 1),
                                      "write_bytes_per_launch": 1024.0 * w / max(nw, 1)}
    out["families"] = families_from_kernels(out["kernels"], steps)
    with open(sys.argv[3], "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out["families"], indent=1))


if __name__ == "__main__":
    main()
