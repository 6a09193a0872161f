export TILES="auto 64x256x0x1x4x2 64x128x0x1x4x2 32x128x0x1x4x2"
for v in "0 0" "1 2" "1 4" "1 1"; do set -- $v
  echo "== DBUF=$1 CG=$2"
  export BODAHIP_EXTRA_DEFS="-DDBUF=$1" BODAHIP_CACHE_DIR=/tmp/kc_db$1$2; [ $2 != 0 ] && export BODAHIP_NHWC_PATCH_CG=$2
  SEL=3,7,12,17 python tools/nhwc_sweep.py resnet-50 12 2>&1 | tail -4
  SEL=2,11,39,45 python tools/nhwc_sweep.py googlenet_conv 12 2>&1 | tail -4
done
