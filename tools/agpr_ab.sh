# GPU tool: accumulators in AGPRs (inline-asm MFMA) vs VGPRs in the patch kernel
python -m pytest tests/test_gpu_nhwc.py -x -q -k "patch_kernel_tiles and direct" 2>&1 | tail -2
for ag in 0 1; do
  echo "== AGPR_ACC=$ag"
  export BODAHIP_EXTRA_DEFS="-DAGPR_ACC=$ag" BODAHIP_CACHE_DIR=/tmp/kc_ag$ag
  [ $ag = 1 ] && python -m pytest tests/test_gpu_nhwc.py -x -q -k "patch_kernel_tiles and direct" 2>&1 | tail -2
  BATCH=256 SEL=1,2,3,4 TILES=auto python tools/nhwc_sweep.py alexnet 8 2>&1 | tail -4
  SEL=3,7,12,17 TILES=auto python tools/nhwc_sweep.py resnet-50 12 2>&1 | tail -4
done
