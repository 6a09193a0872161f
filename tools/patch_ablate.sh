for ab in "" "-DABLATE=1" "-DABLATE=2" "-DABLATE=3" "-DABLATE=4"; do
  if [ -n "$ab" ]; then export BODAHIP_EXTRA_DEFS="$ab" BODAHIP_CACHE_DIR=/tmp/kc_pab$(echo $ab | tr -d '=-'); else unset BODAHIP_EXTRA_DEFS BODAHIP_CACHE_DIR; fi
  echo "== ${ab:-full}"
  TILES=auto SEL=3,7,12,17 python tools/nhwc_sweep.py resnet-50 12 2>&1 | tail -4 | cut -c1-110
  TILES=auto SEL=2,11,39 python tools/nhwc_sweep.py googlenet_conv 12 2>&1 | tail -3 | cut -c1-110
done
