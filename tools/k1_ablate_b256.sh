# GPU tool: ablations of the implicit-GEMM kernel on NiN's 1x1 layers at 256 images (full | 1 no operand loads | 2 no reads / MFMAs | 4 no K loop | 6 no epilogue)
for ab in 0 1 2 4 6; do
  export BODAHIP_EXTRA_DEFS="-DABLATE=$ab" BODAHIP_CACHE_DIR=/tmp/kc_k1ab$ab
  [ $ab = 0 ] && export BODAHIP_EXTRA_DEFS=""
  echo "== ABLATE=$ab"; BATCH=256 SEL=1,3,5,7 TILES=auto python tools/nhwc_sweep.py nin 8 2>&1 | tail -4
done
