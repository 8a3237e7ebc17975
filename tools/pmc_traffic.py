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

# A kernel belongs to a family by its BASE name plus, where one base name serves several layer kinds, the template arguments that
# say which (round 3 matched on a textual prefix of the template list: when `wgrad_taps_kernel<12, 1>` grew a third argument the family
# silently lost 112 launches).  classify() must return a family for every convolution-like kernel -- "other" is an explicit answer --
# and main() fails when one of them is unknown.
def split_name(name):
    """'void endo::b16::conv_kernel<3, 1, 0>(endo::P)' -> ('conv_kernel', ['3', '1', '0'])"""
    name = name.split("(")[0].strip()
    if name.startswith("void "):
        name = name[5:]
    args = []
    if "<" in name:
        name, rest = name.split("<", 1)
        args = [a.strip() for a in rest.rstrip(">").split(",")]
    return name.split("::")[-1], args


CONV_LIKE = re.compile(r"wgrad|dgrad|conv|wino")


def classify(name):
    """family of bench.py's profiling scopes, 'other' for convolution-like kernels outside the three dense-layer families, None for
    kernels that are not convolution-like; raises on a convolution-like kernel this table does not know."""
    base, a = split_name(name)
    if not CONV_LIKE.search(base):
        return None
    if base in ("wgrad_nsplit_kernel", "wgrad_nsplit_reduce_kernel", "wgrad_wino_kernel", "wgrad_wino_reduce_kernel", "wgrad_wino_finish_kernel",
                "wgrad_f34_kernel", "wgrad_f34_reduce_kernel", "wgrad_f34_reduce_batch_kernel", "wgrad_x3_kernel"):
        return "wgrad_dense"
    if base == "wgrad_taps_kernel":                       # <COUT, IN_MODE, BF>: IN_MODE 1 = BN + ReLU input (dense layer), 0 = raw (first conv), 2 = x2 gather (transition up)
        return "wgrad_dense" if a[1] == "1" else "other"
    if base == "wgrad_mfma_kernel":                       # <KS, IN_MODE, ...>: the register-staged fallback
        return "wgrad_dense" if a[0] == "3" and a[1] == "1" else "other"
    if base in ("dgrad_block_kernel", "dgrad_newmap_kernel", "dgrad_block8_kernel", "dgrad_wino8_kernel", "dgrad_wino3_kernel", "dgrad_wino3p_kernel", "dgrad_dense_kernel"):
        return "dgrad_dense"
    if base in ("wino_fwd_kernel", "wino4_fwd_kernel", "finalize_partial_kernel"):
        return "conv3x3_dense_fwd"
    if base in ("conv_dma_kernel", "conv_mfma_kernel"):   # <KS, KC, Q, IN_MODE, EPI, ...>
        if a[0] == "3" and a[2] == "1" and a[3] == "1" and a[4] == "0":
            return "conv3x3_dense_fwd"
        if a[0] == "3" and a[2] == "1" and a[3] == "0" and a[4] == "2":
            return "dgrad_dense"
        return "other"
    # the 16-bit-storage family (bench.py --config 2 / 4)
    if base == "bf16_dgrad_block_kernel":
        return "dgrad_dense"
    if base == "bf16_conv_kernel":                        # <KS, NT, EPI, ...>
        if a[0] == "3" and a[1] == "3" and a[2] == "2":
            return "dgrad_dense"
        if a[0] == "3" and a[1] == "1" and a[2] == "0":
            return "conv3x3_dense_fwd"
        return "other"
    if base == "bf16_wgrad_kernel":                       # <KS, T>: KS 3 also serves the five transition-up layers and the first convolution (6 of 50 launches)
        return "wgrad_dense" if a[0] == "3" else "other"
    if base == "bf16_wgrad_reduce_kernel":                # one per bf16_wgrad_kernel launch, 3x3 and 1x1 alike (5 of 55 are the 1x1's: ~1 % of the family's bytes)
        return "wgrad_dense"
    if base in ("wgrad1x1_dma_kernel", "wgrad1x1_mfma_kernel", "tu_wgrad_subpix_kernel", "tu_wgrad_subpix_reduce_kernel", "wino_fwd_weights_kernel", "wino4_fwd_weights_kernel",
                "dgrad_wino_weights_kernel", "tu_subpix_dgrad_weights_kernel", "wgrad_wino_weights_kernel", "td_bwd_prep_kernel",
                "td_dgrad_gemm_kernel", "td_wgrad_gemm_kernel", "td_wgrad_reduce_kernel", "td_dgrad_kernel", "td_dgrad_small_kernel", "td_fwd_kernel"):
        return "other"
    raise SystemExit("pmc_traffic: convolution-like kernel %r belongs to no family -- extend classify()" % name)


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                       "group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def families_from_kernels(kernels, steps):
    fams = {}
    for name, v in kernels.items():
        fam = classify(name)
        if fam is None or fam == "other":
            continue
        f = fams.setdefault(fam, {"kernel_dispatches_per_step": 0.0, "fetch_bytes_per_step": 0.0, "write_bytes_per_step": 0.0})
        f["kernel_dispatches_per_step"] += v["launches"] / steps
        f["fetch_bytes_per_step"] += v["fetch_bytes_per_launch"] * v["launches"] / steps
        f["write_bytes_per_step"] += v["write_bytes_per_launch"] * v["launches"] / steps
    for f in fams.values():
        f["traffic_bytes_per_step"] = f["fetch_bytes_per_step"] + f["write_bytes_per_step"]
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
        out["kernels"][name[:160]] = {"launches": max(nf, nw), "fetch_bytes_per_launch": 2048.0 * f / max(nf, 1),
                                      "write_bytes_per_launch": 1024.0 * w / max(nw, 1)}
    out["families"] = families_from_kernels(out["kernels"], steps)
    with open(sys.argv[3], "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out["families"], indent=1))


if __name__ == "__main__":
    main()
