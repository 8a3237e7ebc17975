cd $GRAFT_REPO_ROOT
for a in "48 16 256 320" "84 16 256 320" "180 16 256 320" "144 16 128 160" "228 16 128 160"; do timeout 300 tools/bin/x3_bench $a 2>&1 | grep "dense-layer\|fp32 MFMA (v\|F(3x3\|error\|failed\|without x" | cut -c1-230; done
