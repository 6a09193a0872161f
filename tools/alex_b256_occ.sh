# GPU tool: occupancy variants of the patch kernel at 256 images (waves per SIMD x fragments in flight)
export TILES="128x128x0x4x1x2 128x128x0x4x1x3 128x128x0x4x1x4 64x256x0x2x2x3 64x128x0x2x1x4"
for pf in 8 6 4 2; do
  echo "== PF=$pf"
  export BODAHIP_EXTRA_DEFS="-DPF=$pf" BODAHIP_CACHE_DIR=/tmp/kc_pf$pf
  BATCH=256 SEL=1,2,3 python tools/nhwc_sweep.py alexnet 8 2>&1 | tail -3
done
