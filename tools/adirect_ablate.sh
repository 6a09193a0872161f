# GPU tool: ablations of the ADIRECT patch kernel on two layers (1 = no global loads, 2 = no MFMAs, 4 = no K loop)
export TILES="128x128x0x4x1x2 64x128x0x2x2x2 64x256x0x2x2x2"
for ab in 0 1 2 4; do
  echo "== ABLATE=$ab"
  export BODAHIP_EXTRA_DEFS="-DABLATE=$ab" BODAHIP_CACHE_DIR=/tmp/kc_ab$ab
  SEL=7,12,17 python tools/nhwc_sweep.py resnet-50 12 2>&1 | tail -3
done
