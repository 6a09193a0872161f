export TILES="auto 64x256x0x1x4x2 64x128x0x1x4x2 128x128x0x2x2x2 32x128x0x1x4x2 64x64x0x2x2x2 32x256x0x1x4x2 128x256x0x2x4x1 128x64x0x2x2x2 64x128x0x2x2x2 64x256x0x2x4x1 128x128x0x2x4x1"
for cg in 0 2 8; do
  echo "== CG=$cg (0 = planner)"
  [ $cg != 0 ] && export BODAHIP_NHWC_PATCH_CG=$cg BODAHIP_CACHE_DIR=/tmp/kc_cg$cg
  SEL=3,7,12,17 python tools/nhwc_sweep.py resnet-50 12 2>&1 | tail -4
  SEL=2,6,11,12,17,30,39,45,51 python tools/nhwc_sweep.py googlenet_conv 12 2>&1 | tail -9
done
