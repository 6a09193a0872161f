# GPU tool: the patch kernel's ADIRECT variant (filter fragments straight from global memory) against the LDS-staged form, per layer and tile.
export BODAHIP_CACHE_DIR=/tmp/kc_ad
echo "== parity under ADIRECT (planner tiles)"
BODAHIP_EXTRA_DEFS="-DADIRECT=1 -DPF=8" python -m pytest tests/test_gpu_nhwc.py -x -q -k "patch" 2>&1 | tail -3
echo "== staged (today)"
TILES="auto" SEL=0,3,7,12,17 python tools/nhwc_sweep.py resnet-50 12 2>&1 | tail -5
TILES="auto" SEL=0,2,6,11,12,17,30,39,45,51 python tools/nhwc_sweep.py googlenet_conv 12 2>&1 | tail -10
export TILES="auto 128x128x0x4x1x2 64x256x0x2x2x2 64x128x0x2x1x2 128x256x0x4x2x1 256x128x0x4x1x1 64x256x0x1x4x2 32x256x0x1x2x2 32x512x0x1x4x2 64x512x0x2x4x1"
for pf in 8 4 12; do
  echo "== ADIRECT PF=$pf"
  export BODAHIP_EXTRA_DEFS="-DADIRECT=1 -DPF=$pf"
  SEL=0,3,7,12,17 python tools/nhwc_sweep.py resnet-50 12 2>&1 | tail -5
  [ $pf = 8 ] && SEL=0,2,6,11,12,17,30,39,45,51 python tools/nhwc_sweep.py googlenet_conv 12 2>&1 | tail -10
done
