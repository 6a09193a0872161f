# GPU tool: ablations of the patch kernel at 256 images (AlexNet conv3 / conv4): 1 = no global loads, 2 = no MFMAs, 4 = no K loop
export TILES="auto 128x128x0x4x1x3"
for ab in 0 1 2 4; do
  echo "== ABLATE=$ab"
  export BODAHIP_EXTRA_DEFS="-DABLATE=$ab" BODAHIP_CACHE_DIR=/tmp/kc_ab$ab
  BATCH=256 SEL=2,3 python tools/nhwc_sweep.py alexnet 8 2>&1 | tail -2
done
