# GPU tool: AlexNet conv1-5 at 256 images on the patch kernel: tiles x channel groups per step
export TILES="auto 128x128x0x4x1x2 64x256x0x2x2x2 256x128x0x4x1x1 128x256x0x4x2x1 128x128x0x4x1x1 256x256x0x4x2x1"
for cg in 0 1 2 3 8; do
  echo "== CG=$cg (0 = planner)"
  [ $cg != 0 ] && export BODAHIP_NHWC_PATCH_CG=$cg BODAHIP_CACHE_DIR=/tmp/kc_cg$cg
  BATCH=256 SEL=0,1,2,3,4 python tools/nhwc_sweep.py alexnet 8 2>&1 | tail -5
done
