# GPU tool: planner choices of the ADIRECT patch kernel against forced tiles and channel groups per step.
export BODAHIP_CACHE_DIR=/tmp/kc_ad
export TILES="auto 128x128x0x4x1x2 64x256x0x2x2x2 128x256x0x4x2x1 64x128x0x1x4x2 64x128x0x2x2x2 128x64x0x4x1x2 64x64x0x2x2x2 32x128x0x1x4x2 64x128x0x2x1x2 128x128x0x4x1x1"
for cg in 0 8 2; do
  echo "== direct, CG=$cg (0 = planner)"
  [ $cg != 0 ] && export BODAHIP_NHWC_PATCH_CG=$cg
  SEL=0,3,7,12,17 python tools/nhwc_sweep.py resnet-50 12 2>&1 | tail -5
  SEL=2,6,11,12,17,30,39,45,51 python tools/nhwc_sweep.py googlenet_conv 12 2>&1 | tail -9
done
