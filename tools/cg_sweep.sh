# GPU tool: channel groups per K step (BODAHIP_NHWC_PATCH_CG) x tile of the channels-last input-patch kernel on chosen GoogLeNet layers (SEL), us per launch
for CG in ${CGS:-1 2 4}; do
  export BODAHIP_NHWC_PATCH_CG=$CG
  echo "=== CG=$CG"
  SEL=${SEL:-2,6,11,12,17,26,30,33,39} SLICES=1 MAXPEL=99999 PATCH_TILES="${PATCH_TILES:-64x256x0x2x2x2 64x128x0x2x2x2 128x128x0x4x1x2}" timeout 600 python tools/ksl_sweep.py run 2>&1 | cut -c1-120
done
