export TILES="auto 128x256x0x2x2x2 128x256x0x2x2x1 128x128x0x1x2x2 128x128x0x2x1x2 64x256x0x1x2x2 128x256x0x2x4x1 256x128x0x2x2x1"
for cg in 0 2; do
  echo "== CG=$cg (0 = planner)"
  [ $cg != 0 ] && export BODAHIP_NHWC_PATCH_CG=$cg BODAHIP_CACHE_DIR=/tmp/kc_cg$cg
  SEL=3,7,12,17 python tools/nhwc_sweep.py resnet-50 12 2>&1 | tail -4
  SEL=2,11,39 python tools/nhwc_sweep.py googlenet_conv 12 2>&1 | tail -3
done
