#!/usr/bin/env python3
"""Per (kernel, grid) stall picture from rocprofv3 --pmc SQ passes.  usage: pmc_sq_report.py <out.txt> <db> [<db> ...]
Counters from several passes are merged by (kernel name, grid); values are averages per dispatch.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
summed over SIMDs (MI355X_MICROARCH.md)."""
import collections
import sqlite3
import sys


def main():
    out = sys.argv[1]
    data = collections.defaultdict(dict)
    dur = {}
    for db in sys.argv[2:]:
        cur = sqlite3.connect(db).cursor()
        for name, grid, wg, cname, n, v, d in cur.execute(
                "select kernel_name, grid_size, workgroup_size, counter_name, count(*), avg(value), avg(end-start) "
                "from counters_collection group by kernel_name, grid_size, counter_name"):
            key = (name, grid, wg)
            data[key][cname] = v
            data[key]["_n"] = n
            dur[key] = d
    keys = sorted(data, key=lambda k: -dur[k] * data[k]["_n"])
    with open(out, "w") as f:
        for k in keys[:40]:
            c = data[k]
            f.write("%s grid=%d wg=%d calls=%d avg_us(serialised)=%.1f\n" % (k[0][:120], k[1], k[2], c["_n"], dur[k] / 1e3))
            wc = c.get("SQ_WAVE_CYCLES")
            parts = []
            if wc:
                for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                           "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA"):
                    if nm in c:
                        parts.append("%s/WAVE=%.3f" % (nm[3:], c[nm] / wc))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
                parts.append("MFMA_BUSY/SQ_BUSY=%.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]))
            f.write("    " + "  ".join(parts) + "\n")
            f.write("    " + "  ".join("%s=%.4g" % (n, v) for n, v in sorted(c.items()) if n != "_n") + "\n")


if __name__ == "__main__":
    main()
