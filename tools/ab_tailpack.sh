set -e
cd $GRAFT_REPO_ROOT/stylegan3-editing_amd/csrc
echo "== with tail packing"
(cd $GRAFT_REPO_ROOT && timeout -k 10 120 python tools/bench_layer.py conv L6 L8 L11 L12 L13 2>&1 | grep "^conv")
cp ../lib/libsg3hip.so /tmp/lib_new.so
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fvisibility=hidden -DSG3_TAILPACK=0 -c sg3_modconv.hip -o /tmp/modconv_old.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libsg3hip.so sg3_bias_act.o sg3_upfirdn2d.o sg3_filtered_lrelu.o /tmp/modconv_old.o sg3_conv2d.o sg3_wgrad.o
echo "== compiled out"
(cd $GRAFT_REPO_ROOT && timeout -k 10 120 python tools/bench_layer.py conv L6 L8 L11 L12 L13 2>&1 | grep "^conv")
cp /tmp/lib_new.so ../lib/libsg3hip.so
