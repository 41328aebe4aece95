# per-kernel rocprofv3 stats for builds under ab/ (VARIANTS), workload $1
L=bot-sort-onnx-tensorrt_amd/libbotsort_hip.so
cp $L /tmp/keep.so
for v in ${VARIANTS:-head chunk2}; do cp ab/$v.so $L; echo "== $v"; WL="$1" bash scripts/gpu_stats.sh | grep -v "^==" | awk -F, '{printf "%-60s %s %s\n", substr($1,1,60), $2, $4}' | head -14; done
cp /tmp/keep.so $L
