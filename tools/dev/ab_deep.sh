# dev A/B: work-list halo2 with the weight fragments fetched one step ahead (shallow) vs two (the build default)
for i in 1 2 3; do
  for v in shallow deep; do
    lib=cvpce_amd/libcvpce_hip.so; [ $v = shallow ] && lib=cvpce_amd/libcvpce_hip_shallow.so
    echo "$v: $(CVPCE_LIB=$PWD/$lib CVPCE_SKIP_ROWS=1 timeout -k 10 200 python tools/dev/embed_layers.py 2>&1 | grep -E 'skip=True|c4_2|c5_1|c3_2' | head -4 | awk '{printf "%s %s | ", $1, $2}')"
  done
done
