# launch floor by events: ABLATE=5 (kernel returns at once) under event flag / timing variants
for fl in 0 0x20000000 0x40000000; do
  echo "== BODAHIP_EVENT_FLAGS=$fl"
  BODAHIP_EVENT_FLAGS=$fl BODAHIP_EXTRA_DEFS="-DABLATE=5" BODAHIP_CACHE_DIR=/tmp/kc5 NET=googlenet_conv SEL=2,9,39 python tools/nhwc_ablate.py child
  BODAHIP_EVENT_FLAGS=$fl NET=googlenet_conv SEL=2,9,39 python tools/nhwc_ablate.py child
done
