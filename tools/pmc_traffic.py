#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately as
MI355X_MICROARCH.md prescribes) -> profiles/<tag>_pmc_traffic.json, which bench.py reads for roofline.traffic.

usage: pmc_traffic.py <fetch.db> <write.db> <out.json> [note]

Corrections applied (same guide, section HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes
of 16 B/lane streaming reads, which is what every kernel here issues (global_load_dwordx4 / 16-byte LDS-DMA), so it
is doubled.  Calibration points inside these very runs: final_bwd_data_kernel writes 192 planes = 491 520 KiB and
WRITE_SIZE says 491 520.0; final_fwd_kernel reads the same 192 planes and FETCH_SIZE says 245 778 (x2 = 491 556)."""
import json
import re
import sqlite3
import sys

FAMILIES = [
    ("dgrad_dense", re.compile(r"dgrad_block_kernel|dgrad_dense_kernel|conv_dma_kernel<3, \d+, 1, 0, 2,")),
    ("conv3x3_dense_fwd", re.compile(r"conv_dma_kernel<3, \d+, 1, 1, 0,")),
    ("wgrad_dense", re.compile(r"wgrad_taps_kernel<12, 1>|wgrad_mfma_kernel<3, 1, 1, 0>")),
]


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                       "group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"note": sys.argv[4] if len(sys.argv) > 4 else "", "unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, KiB -> B)",
           "families": {}, "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        if "endo::" not in name:
            continue
        nf, f = fetch.get(name, (0, 0.0))
        nw, w = write.get(name, (0, 0.0))
        out["kernels"][name[:110]] = {"launches": max(nf, nw), "fetch_bytes_per_launch": 2048.0 * f / max(nf, 1),
                                      "write_bytes_per_launch": 1024.0 * w / max(nw, 1)}
    for fam, rx in FAMILIES:
        nf = sum(v[0] for k, v in fetch.items() if rx.search(k))
        f = sum(v[1] for k, v in fetch.items() if rx.search(k))
        nw = sum(v[0] for k, v in write.items() if rx.search(k))
        w = sum(v[1] for k, v in write.items() if rx.search(k))
        if nf and nw:
            out["families"][fam] = {"launches": nf, "fetch_bytes_per_launch": 2048.0 * f / nf,
                                    "write_bytes_per_launch": 1024.0 * w / nw,
                                    "traffic_bytes_per_launch": 2048.0 * f / nf + 1024.0 * w / nw}
    with open(sys.argv[3], "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out["families"], indent=1))


if __name__ == "__main__":
    main()
