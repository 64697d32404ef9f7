set -e
python tools/dev/real_layer_bench.py capture 2>&1 | tail -1
for r in 1 2; do
for d in 0 4 8 16 32 12 28 60; do
  if [ $d == 0 ]; then lib=""; else lib=cvpce_amd/libcvpce_hip_conv3x3_halo3_dbg$d.so; fi
  echo "== dbg $d"; CVPCE_LIB=$lib python tools/dev/real_layer_bench.py time conv2_1,conv2_2 2>&1 | grep conv2
done
done
