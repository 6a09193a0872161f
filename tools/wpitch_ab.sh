# GPU tool: padded slot pitch of the patch kernel (bank-conflict-free fragment reads) on / off, per layer
python -m pytest tests/test_gpu_nhwc.py -x -q -k "patch_kernel_tiles and direct" 2>&1 | tail -2
for wp in 0 1; do
  echo "== WPITCH=$wp"
  export BODAHIP_EXTRA_DEFS="-DWPITCH=$wp" BODAHIP_CACHE_DIR=/tmp/kc_wp$wp
  BATCH=256 SEL=0,1,2,3,4 TILES=auto python tools/nhwc_sweep.py alexnet 8 2>&1 | tail -5
  SEL=0,3,7,12,17 TILES=auto python tools/nhwc_sweep.py resnet-50 12 2>&1 | tail -5
  SEL=2,6,11,12,17,30,39,45,51 TILES=auto python tools/nhwc_sweep.py googlenet_conv 12 2>&1 | tail -9
done
