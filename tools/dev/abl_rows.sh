for d in 0 4 8 32; do
  lib=cvpce_amd/libcvpce_hip_conv3x3_halo2_dbg$d.so; [ $d = 0 ] && lib=cvpce_amd/libcvpce_hip.so
  for r in 0 1; do
    echo "== dbg $d rows $r: $(CVPCE_LIB=$PWD/$lib CVPCE_SKIP_ROWS=$r timeout -k 10 200 python tools/dev/embed_layers.py 2>&1 | grep -E 'c4_1|c4_2|c5_1' | head -3 | awk '{printf "%s %s ms | ", $1, $2}')"
  done
done
