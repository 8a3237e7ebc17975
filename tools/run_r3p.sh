#!/bin/bash
# round 3, GPU run p: HBM-side traffic of the bf16 conv variants (separate --pmc passes)
mkdir -p gpurun_out/r3p; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c -d $R/gpurun_out/r3p/$tag -o out --output-format csv -- $R/tools/bin/bf16_conv_variants 180 > $R/gpurun_out/r3p/$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r3p/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:70], r['Counter_Name'])
        acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        print(f"{c:14s} {v / n:14.1f} per dispatch  x{n:3d}  {k}")
PY
