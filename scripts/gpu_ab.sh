# A/B of builds of the library kept under ab/ (alternating, same box): VARIANTS="old new" bash scripts/gpu_ab.sh c2 c3
L=bot-sort-onnx-tensorrt_amd/libbotsort_hip.so
cp $L /tmp/keep.so
for rep in 1 2 3; do for v in ${VARIANTS:-old new}; do cp ab/$v.so $L; echo "== $v"; bash scripts/${QUICK:-gpu_lapq.sh} ${@:-c2 c3 target}; done; done
cp /tmp/keep.so $L
